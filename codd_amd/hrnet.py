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


def chain_blocks(owner, blocks):
    """ops.PackedChain of consecutive BasicBlocks (conv-bn-relu, conv-bn + x, relu each; eval-mode BatchNorm folded):
    one launch, the activations between the 2 * len(blocks) convolutions stay in LDS (csrc/chain.hip).  Cached on
    ``owner`` per parameter version; dropped by stereo.invalidate_packed."""
    pcs = [packed_cbn(c, b) for blk in blocks for c, b in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2))]
    ver = tuple(id(pc) for pc in pcs)  # packed_cbn re-creates the PackedConv whenever conv / bn parameters change
    cache = owner.__dict__.setdefault("_codd_packed_cat", {})
    ent = cache.get("chain")
    if ent is None or ent[0] != ver:
        layers, n = [], len(blocks)
        for i in range(n):  # block input in buffer 0 (staged for the first block): conv1 0 -> 1, conv2 1 -> 0 (+ 0)
            c1, c2 = pcs[2 * i], pcs[2 * i + 1]
            layers.append(dict(w=c1._w, b=c1.bias, src=-1 if i == 0 else 0, dst=1, act="relu"))
            layers.append(dict(w=c2._w, b=c2.bias, src=1, dst=-1 if i == n - 1 else 0, res=0, act="relu"))
        ent = cache["chain"] = (ver, ops.PackedChain(layers, stage=0), pcs)
    return ent[1]


CHAIN_MAX_CHANNELS = 48  # codd_conv_chain: cout <= 48 (three 16-channel MFMA blocks per wave)
import os as _os
# the 2-4 resolution branches of an HRModule (and then its fused outputs) on parallel streams.  OFF: eager it works
# (context network alone 2.1 -> 1.6 ms), but the frame runs as a captured hipGraph and ROCm 7.2's hipGraphInstantiate
# segfaults on these nested forks -- in round 2 with the branch streams forked from the context network's side stream,
# in round 3 also with ops.Fork.prefork (streams brought into the capture through the origin stream first).
FORK_BRANCHES = _os.environ.get("CODD_HRNET_FORK", "0") == "1"


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


# HRModule fuse layers as level-wise multi-job convolutions + one summation launch per branch (A/B switch).  OFF: 45
# launches per frame less but the same frame rate (the context network is hidden beside the stereo network), and the
# other fp32 summation order inside the multi-job convolutions is enough to move the one near-camera pixel cluster of
# BASELINE.json configs[4] across a splat pixel boundary (mean |disparity delta| over all pixels 1.9e-5 -> 2.5e-3 px on
# frame 1) -- the context features enter all 16 updates
FUSE_SUM = _os.environ.get("CODD_HR_FUSE_SUM", "0") == "1"
# round 4: only the SUMMATION of a fuse layer as one launch per output branch, convolutions untouched: 26 launches per
# frame less and +1.0 % frame rate (every launch of the frame graph costs ~3.8 us of wall clock, DESIGN finding 43) --
# but the sum kernel rounds one ulp differently from resize-accumulate (fp contraction), and that alone flips
# configs[4]'s near-camera cluster (2.5e-3 px on frame 1, finding 30): OFF.
FUSE_TERMS = _os.environ.get("CODD_HR_FUSE_TERMS", "0") == "1"
FOLD_SELF = _os.environ.get("CODD_HR_FOLD_SELF", "1") == "1"  # (A/B: the "+ x_i" term of a fuse layer without its own launch)
DEFER_FUSE = _os.environ.get("CODD_HR_DEFER_FUSE", "1") == "1"
LOCKSTEP_FUSE = _os.environ.get("CODD_HR_LOCKSTEP_FUSE", "1") == "1"  # (A/B: chain-ending convolutions of all branches per j)  # (A/B: fuse-layer convolutions as deferred multi-job launches)


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

    def run(self, xs, fk=None):
        """``fk``: ops.Fork with >= len(xs) - 1 streams -- the branches, and then the fused outputs, are independent of
        each other and run side by side (branch / output 0 on the caller's stream)."""
        nb = len(xs)
        xs = list(xs)

        def on(i, fn):
            return fn() if fk is None or i == 0 else fk.run(i - 1, fn)

        def branch(i):
            x = xs[i]
            blocks = list(self.branches[i])
            if ops.use_chain(*x.shape[2:]) and blocks[0].conv1.out_channels <= CHAIN_MAX_CHANNELS:
                return ops.conv_chain(x, chain_blocks(self.branches[i], blocks))  # the branch in one launch
            for blk in blocks:
                x = blk.run(x)
            return x

        nblk = len(self.branches[0])
        if fk is None and nb > 1 and ops.MULTI_CONV and ops.CONV_PRECISION == "fp32" and not ops.use_chain(*xs[0].shape[2:]):
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
            ys = [on(i, lambda i=i: branch(i)) for i in range(nb)]
            if fk is not None:
                fk.join()
        xs = ys

        if fk is None and FUSE_SUM:
            # fuse layers: every path's convolutions level by level as multi-job launches (ops.conv2d_multi; the
            # paths are independent of each other), then ONE summation launch per output branch (ops.hr_fuse_sum)
            # instead of a resize / add launch per term: 9 instead of ~40 launches for a 4-branch module
            term = {}
            level = []  # (i, j, k): conv k of path j -> i still to run; t = its input
            for i in range(nb):
                for j in range(nb):
                    if j != i:
                        level.append((i, j, 0, xs[j]))
            while level:
                jobs, nxt = [], []
                for i, j, k, t in level:
                    if j > i:
                        f = self.fuse_layers[i][j]
                        jobs.append(dict(x=t, pc=packed_cbn(f[0], f[1]), stride=1, pad=0, act="none"))
                    else:
                        f = self.fuse_layers[i][j][k]
                        jobs.append(dict(x=t, pc=packed_cbn(f[0], f[1]), stride=2, pad=1,
                                         act="relu" if k != i - j - 1 else "none"))
                ys = ops.conv2d_multi(jobs)
                for (i, j, k, _), y in zip(level, ys):
                    if j > i or k == i - j - 1:
                        term[(i, j)] = y
                    else:
                        nxt.append((i, j, k + 1, y))
                level = nxt
            return [ops.hr_fuse_sum([xs[i] if j == i else term[(i, j)] for j in range(nb)], xs[i].shape[2:], relu=True)
                    for i in range(nb)]

        def fuse_terms(i):
            """Output branch i with the convolutions exactly as below (same launches, same bits) but ONE summation
            launch (ops.hr_fuse_sum: same bilinear expression, same j order) instead of a resize / add launch per term."""
            terms = []
            for j in range(nb):
                if j == i:
                    terms.append(xs[j])
                elif j > i:
                    f = self.fuse_layers[i][j]
                    terms.append(cbn(f[0], f[1], xs[j]))
                else:
                    t = xs[j]
                    chain = self.fuse_layers[i][j]
                    for k, f in enumerate(chain):
                        t = cbn(f[0], f[1], t, "relu" if k != len(chain) - 1 else "none")
                    terms.append(t)
            return ops.hr_fuse_sum(terms, xs[i].shape[2:], relu=True)

        if FUSE_TERMS and fk is None:
            return [fuse_terms(i) for i in range(nb)]

        # Every convolution of the fuse layers that reads branch outputs only -- the 1x1 convolutions of the up paths
        # (j > i) and all but the last convolution of the down chains (j < i) -- is independent of the others: they are
        # issued level by level inside ops.deferred_convs() and leave as multi-job launches with the parameters of
        # their single launches (bit-identical; 33 convolutions of a 4-branch network in ~10 launches).  The
        # accumulation into the output branch (resize / add / last chain convolution with its res1 operand) keeps its
        # launches and its j order.
        pre = {}
        if fk is None and DEFER_FUSE:
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

        def fuse(i):
            acc = torch.empty_like(xs[i])
            # the "+ x_i" term has no launch of its own (FOLD_SELF): it rides on the next up-sampling term as its
            # ``extra`` addend, or -- for the last branch, where it is the last term -- on the last chain convolution
            # as res2; (acc + x_i) + term / ((conv + acc) + x_i): the same roundings in the same order
            fold = FOLD_SELF and xs[i].is_contiguous()
            carry = None
            for j in range(nb):
                first, last = j == 0, j == nb - 1
                if j == i:
                    if fold and not last:
                        carry = xs[j]  # added by the next term (j + 1 > i: an up-sampling term)
                    elif not (fold and last and nb > 1):  # (folded into the previous chain convolution below)
                        ops.add_relu(xs[j], None if first else acc, relu=last, out=acc)
                elif j > i:
                    f = self.fuse_layers[i][j]
                    t = pre[(i, j)] if (i, j) in pre else cbn(f[0], f[1], xs[j])
                    if carry is not None:
                        ops.resize_bilinear(t, xs[i].shape[2:], False, out=acc, accumulate=j - 1 > 0, relu=last, extra=carry)
                        carry = None
                    else:
                        ops.resize_bilinear(t, xs[i].shape[2:], False, out=acc, accumulate=not first, relu=last)
                else:
                    chain = self.fuse_layers[i][j]
                    if (i, j) in pre:
                        t, k0 = pre[(i, j)], len(chain) - 1
                    else:
                        t, k0 = xs[j], 0
                    self_next = fold and i == nb - 1 and j == i - 1  # x_i (the last term) rides on this chain's last conv
                    for k in range(k0, len(chain)):
                        f = chain[k]
                        if k != len(chain) - 1:
                            t = cbn(f[0], f[1], t, "relu")
                        elif self_next:
                            cbn(f[0], f[1], t, "relu", res1=None if first else acc, res2=xs[i], out=acc)
                        else:
                            cbn(f[0], f[1], t, "relu" if last else "none", res1=None if first else acc, out=acc)
            return acc

        if fk is None and DEFER_FUSE and LOCKSTEP_FUSE:
            # all output branches in lockstep over j: term j of every branch at once.  The last convolutions of the down
            # chains j -> i (i > j) accumulate into DIFFERENT output branches, so they do not depend on one another and
            # leave as one deferred multi-job launch per j (13 instead of 22 launches per frame); every branch still
            # sees its terms in j order with the same operands, i.e. the same bits as fuse(i) above.
            accs = [torch.empty_like(x) for x in xs]
            folds = [FOLD_SELF and x.is_contiguous() for x in xs]
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
                    t = pre[(i, j)] if (i, j) in pre else cbn(f[0], f[1], xs[j])
                    if carry[i] is not None:
                        ops.resize_bilinear(t, xs[i].shape[2:], False, out=accs[i], accumulate=j - 1 > 0, relu=last, extra=carry[i])
                        carry[i] = None
                    else:
                        ops.resize_bilinear(t, xs[i].shape[2:], False, out=accs[i], accumulate=not first, relu=last)
            return accs

        outs = [on(i, lambda i=i: fuse(i)) for i in range(nb)]
        if fk is not None:
            fk.join()
        return outs


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
        fk = self.fork(x.device) if FORK_BRANCHES and getattr(self, "fork_branches", True) else None
        for m in self.stage2:
            ys = m.run(ys, fk)
        t = self.transition2[2][0]
        ys = ys + [cbn(t[0], t[1], ys[-1], "relu")]
        for m in self.stage3:
            ys = m.run(ys, fk)
        t = self.transition3[3][0]
        ys = ys + [cbn(t[0], t[1], ys[-1], "relu")]
        for m in self.stage4:
            ys = m.run(ys, fk)
        return ys

    def fork(self, device):
        """The branch streams (callers that run this network on a side stream pre-fork them from their origin stream:
        ops.Fork.prefork)."""
        fk = self.__dict__.get("_fk")
        if fk is None or fk.dev != device:
            fk = self.__dict__["_fk"] = ops.Fork(device, len(self.out_channels) - 1)
        return fk


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
