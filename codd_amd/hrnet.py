"""RAFT3D context network: HRNetV2-W18-small-v2 + ResizeConcatConv on the HIP conv family.

reference: raft3d.py:109-137 (ResizeConcatConv), :152-160 (cnet), config
configs/models/codd.py:44-74.  The backbone is mmseg's ``HRNet`` (un-vendored dependency); the
module tree below reproduces its attribute names so that mmseg-format state dicts load by key.
BatchNorm runs in eval mode (``norm_eval=True``, frozen) and is folded into the preceding
convolution when the weights are packed.
"""
import torch
import torch.nn as nn

from . import ops
from .ops import Slice
from .registry import register

def packed_cbn(conv, bn):
    """PackedConv of conv followed by eval-mode BatchNorm (folded); cached ON the conv module per parameter version
    (freed with the model; the entry keeps its BatchNorm partner alive, so the id in the key cannot be recycled)."""
    ver = (conv.weight.data_ptr(), conv.weight._version, bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version)
    cache = conv.__dict__.setdefault("_codd_packed_cat", {})
    key = ("bn", id(bn))
    ent = cache.get(key)
    if ent is None or ent[0] != ver:
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        w = conv.weight.detach() * s.view(-1, 1, 1, 1)
        b = bn.bias.detach() - bn.running_mean.detach() * s
        if conv.bias is not None:
            b = b + conv.bias.detach() * s
        ent = cache[key] = (ver, ops.PackedConv(w, b), bn)
    return ent[1]


def cbn(conv, bn, x, act="none", **kw):
    return ops.conv2d(x, packed_cbn(conv, bn), stride=tuple(conv.stride), pad=tuple(conv.padding), act=act, **kw)


def _conv(i, o, k, s=1, p=0):
    return nn.Conv2d(i, o, k, s, p, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def run(self, x):
        y = cbn(self.conv1, self.bn1, x, "relu")
        y = cbn(self.conv2, self.bn2, y, "relu")
        idn = x if self.downsample is None else cbn(self.downsample[0], self.downsample[1], x)
        return cbn(self.conv3, self.bn3, y, "relu", res1=idn)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, c):
        super().__init__()
        self.conv1 = _conv(c, c, 3, 1, 1)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = _conv(c, c, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(c)

    def run(self, x):
        y = cbn(self.conv1, self.bn1, x, "relu")
        return cbn(self.conv2, self.bn2, y, "relu", res1=x)


class HRModule(nn.Module):
    def __init__(self, channels, num_blocks):
        super().__init__()
        nb = len(channels)
        self.branches = nn.ModuleList([nn.Sequential(*[BasicBlock(c) for _ in range(num_blocks)]) for c in channels])
        fuse = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(_conv(channels[j], channels[i], 1), nn.BatchNorm2d(channels[i])))
                elif j == i:
                    row.append(None)
                else:
                    chain = []
                    for k in range(i - j):
                        co = channels[i] if k == i - j - 1 else channels[j]
                        mods = [_conv(channels[j], co, 3, 2, 1), nn.BatchNorm2d(co)]
                        if k != i - j - 1:
                            mods.append(nn.ReLU(inplace=False))
                        chain.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*chain))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)

    def run(self, xs):
        nb = len(xs)
        xs = list(xs)
        nblk = len(self.branches[0])
        if nb > 1 and ops.MULTI_CONV and ops.CONV_PRECISION == "fp32":
            # step s of every branch (conv1 / conv2 of block s // 2) is independent of the other branches: one
            # multi-job launch per step (ops.conv2d_multi) instead of one ~10 us launch per branch and step
            ys, blockin = list(xs), list(xs)
            for st in range(2 * nblk):
                jobs = []
                for i in range(nb):
                    blk = self.branches[i][st // 2]
                    conv, bn = (blk.conv1, blk.bn1) if st % 2 == 0 else (blk.conv2, blk.bn2)
                    jobs.append(dict(x=ys[i], pc=packed_cbn(conv, bn), pad=1, act="relu", res1=blockin[i] if st % 2 else None))
                ys = ops.conv2d_multi(jobs)
                if st % 2:
                    blockin = list(ys)
        else:
            ys = []
            for i in range(nb):
                x = xs[i]
                for blk in self.branches[i]:
                    x = blk.run(x)
                ys.append(x)
        xs = ys

        # Fuse layers (mmseg HRModule.forward: out_i = relu(sum_j t_ij) in j order).  Every convolution that reads branch
        # outputs only -- the 1x1 convolutions of the up paths (j > i) and all but the last convolution of the down chains
        # (j < i) -- is independent of the others: they are issued level by level inside ops.deferred_convs() and leave as
        # multi-job launches with the parameters of their single launches (33 convolutions of a 4-branch network in ~10
        # launches, bit-identical to the single launches).
        pre = {}
        level = [(i, j, 0, xs[j]) for i in range(nb) for j in range(nb) if j > i or i - j >= 2]
        while level:
            nxt = []
            with ops.deferred_convs():
                for i, j, k, t in level:
                    if j > i:
                        f = self.fuse_layers[i][j]
                        pre[(i, j)] = cbn(f[0], f[1], t)
                    else:
                        f = self.fuse_layers[i][j][k]
                        y = cbn(f[0], f[1], t, "relu")
                        if k + 2 < i - j:
                            nxt.append((i, j, k + 1, y))
                        else:
                            pre[(i, j)] = y  # input of the chain's last convolution
            level = nxt
        # All output branches in lockstep over j: term j of every branch at once.  The last convolutions of the down chains
        # j -> i (i > j) accumulate into DIFFERENT output branches, so they leave as one deferred multi-job launch per j;
        # every branch sees its terms in j order.  The "+ x_i" term has no launch of its own: it rides on the next
        # up-sampling term as its ``extra`` addend, or -- for the last branch, where it is the last term -- on the last
        # chain convolution as res2: (acc + x_i) + term / ((conv + acc) + x_i), the roundings of a separate add launch.
        accs = [torch.empty_like(x) for x in xs]
        folds = [x.is_contiguous() for x in xs]
        carry = [None] * nb
        for j in range(nb):
            first, last = j == 0, j == nb - 1
            with ops.deferred_convs():
                for i in range(j + 1, nb):  # down chains j -> i: the chain's last convolution
                    chain = self.fuse_layers[i][j]
                    t = pre[(i, j)] if (i, j) in pre else xs[j]
                    assert (i, j) in pre or len(chain) == 1
                    f = chain[-1]
                    if folds[i] and i == nb - 1 and j == i - 1:  # x_i (the branch's last term) rides along as res2
                        cbn(f[0], f[1], t, "relu", res1=None if first else accs[i], res2=xs[i], out=accs[i])
                    else:
                        cbn(f[0], f[1], t, "relu" if last else "none", res1=None if first else accs[i], out=accs[i])
            # the branch's own term
            if folds[j] and not last:
                carry[j] = xs[j]
            elif not (folds[j] and last and nb > 1):
                ops.add_relu(xs[j], None if first else accs[j], relu=last, out=accs[j])
            for i in range(j):  # up paths j -> i
                f = self.fuse_layers[i][j]
                t = pre[(i, j)]
                if carry[i] is not None:
                    ops.resize_bilinear(t, xs[i].shape[2:], False, out=accs[i], accumulate=j - 1 > 0, relu=last, extra=carry[i])
                    carry[i] = None
                else:
                    ops.resize_bilinear(t, xs[i].shape[2:], False, out=accs[i], accumulate=not first, relu=last)
        return accs


def _transition_new(cin, cout):
    return nn.Sequential(nn.Sequential(_conv(cin, cout, 3, 2, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)))


@register
class HRNet(ops.RuntimeState, nn.Module):
    """mmseg.models.backbones.HRNet restricted to what configs/models/codd.py:44-74 uses."""

    def __init__(self, extra=None, norm_cfg=None, norm_eval=True, init_cfg=None, **kwargs):
        super().__init__()
        extra = extra or {}
        s2 = tuple(extra.get("stage2", {}).get("num_channels", (18, 36)))
        s3 = tuple(extra.get("stage3", {}).get("num_channels", (18, 36, 72)))
        s4 = tuple(extra.get("stage4", {}).get("num_channels", (18, 36, 72, 144)))
        nm = [extra.get(f"stage{i}", {}).get("num_modules", d) for i, d in ((2, 1), (3, 3), (4, 2))]
        nblk = [extra.get(f"stage{i}", {}).get("num_blocks", (2,))[0] for i in (2, 3, 4)]
        self.conv1 = _conv(3, 64, 3, 2, 1)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = _conv(64, 64, 3, 2, 1)
        self.bn2 = nn.BatchNorm2d(64)
        down = nn.Sequential(_conv(64, 256, 1), nn.BatchNorm2d(256))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, down), Bottleneck(256, 64))
        self.transition1 = nn.ModuleList([
            nn.Sequential(_conv(256, s2[0], 3, 1, 1), nn.BatchNorm2d(s2[0]), nn.ReLU(inplace=True)),
            _transition_new(256, s2[1])])
        self.stage2 = nn.Sequential(*[HRModule(s2, nblk[0]) for _ in range(nm[0])])
        self.transition2 = nn.ModuleList([None, None, _transition_new(s2[-1], s3[2])])
        self.stage3 = nn.Sequential(*[HRModule(s3, nblk[1]) for _ in range(nm[1])])
        self.transition3 = nn.ModuleList([None, None, None, _transition_new(s3[-1], s4[3])])
        self.stage4 = nn.Sequential(*[HRModule(s4, nblk[2]) for _ in range(nm[2])])
        self.out_channels = s4

    def forward(self, x):
        x = cbn(self.conv1, self.bn1, x, "relu")
        x = cbn(self.conv2, self.bn2, x, "relu")
        for blk in self.layer1:
            x = blk.run(x)
        t0, t1 = self.transition1[0], self.transition1[1][0]
        ys = [cbn(t0[0], t0[1], x, "relu"), cbn(t1[0], t1[1], x, "relu")]
        for m in self.stage2:
            ys = m.run(ys)
        t = self.transition2[2][0]
        ys = ys + [cbn(t[0], t[1], ys[-1], "relu")]
        for m in self.stage3:
            ys = m.run(ys)
        t = self.transition3[3][0]
        ys = ys + [cbn(t[0], t[1], ys[-1], "relu")]
        for m in self.stage4:
            ys = m.run(ys)
        return ys


class ResizeConcatConv(nn.Module):
    """reference raft3d.py:109-137."""

    def __init__(self, in_channels, out_channels=32):
        super().__init__()
        self.in_channels = tuple(in_channels)
        self.out_channels = out_channels
        self.convs = nn.Sequential(nn.Conv2d(sum(in_channels), out_channels, 1, bias=False), nn.ReLU(inplace=True))

    def forward(self, inputs):
        from .stereo import packed
        B = inputs[0].shape[0]
        size = tuple(inputs[1].shape[2:])
        cat = torch.empty(B, sum(self.in_channels), *size, device=inputs[0].device, dtype=torch.float32)
        off = 0
        for x, c in zip(inputs, self.in_channels):
            ops.resize_bilinear(x, size, True, out=Slice(cat, off, c))
            off += c
        return ops.conv2d(cat, packed(self.convs[0]), act="relu")
