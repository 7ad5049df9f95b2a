"""Motion + RAFT3D on MI355X.

reference: model/motion/motion.py:48-209, model/motion/raft3d/raft3d.py:44-280,
blocks/{extractor,gru,corr}.py, se3_field.py, projective_ops.py.  The lietorch / lietorch_extras /
pytorch3d CUDA ops of the reference are replaced by the HIP kernels in csrc/motion.hip; an SE3
field is a plain float tensor [B, H, W, 7] = (t, q_xyzw).

Per-iteration schedule (16x per frame): geometry -> pyramid lookup -> flow/corr encoders ->
ConvGRU (z|r as one 256-channel conv pair, q) -> one 1024-channel head conv + four 1x1 heads ->
Gauss-Newton step.  Attribute names equal the reference's, so checkpoints load by key.
"""

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .hrnet import ResizeConcatConv
from .ops import Slice
from .registry import MODELS, build_backbone, build_loss, register
from .stereo import cv, packed

BF_DEFAULT = 1050 * 0.2  # reference motion.py:45
MERGE_ENC_HEADS = True  # (A/B switch; see BasicUpdateBlock.run)
FUSE_NORM_RECORDS = True  # (A/B switch; see ResidualBlock.run)
# ConvGRU gates as convolution epilogues + each conv*1 / conv*2 pair as ONE dual-tap-set launch (BasicUpdateBlock.run)
FUSE_GATES = True
# the feature encoder runs on a side stream beside the stereo network: small-footprint launch configurations (A/B)
FNET_CORESIDENT = False
# the state-only launches in front of the first update (RAFT3D._preloop) on the fnet side stream behind the pyramid
PRELOOP_SIDE = True
# the flow encoder's 7x7 convolution beside the correlation encoder's first 3x3: 1 = small-footprint configurations for
# the former, 2 = for both (A/B)
ENC_CORESIDENT = 1
# side streams inside the update block (A/B switches): the flow encoder / mask head beside the correlation encoder, and
# the next update's z|r convolution beside the Gauss-Newton step
LOOP_FORK_ENC = True
LOOP_FORK_ZR = True

def packed_cat(mods):
    """One PackedConv whose output channels are the concatenation of several same-shape convs; cached on the first
    module (keyed by the partners' identities, which the cache entry keeps alive)."""
    ver = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in mods)
    cache = mods[0].__dict__.setdefault("_codd_packed_cat", {})
    key = tuple(id(m) for m in mods)
    ent = cache.get(key)
    if ent is None or ent[0] != ver:
        w = torch.cat([m.weight.detach() for m in mods], 0)
        b = torch.cat([m.bias.detach() for m in mods], 0)
        ent = cache[key] = (ver, ops.PackedConv(w, b), tuple(mods))
    return ent[1]


def packed_cat_in(mods):
    """One PackedConv computing the SUM of several same-geometry convs of different inputs: weights concatenated along
    the input channels (the inputs are concatenated in the same order), biases added.  conv_a(x_a) + conv_b(x_b) =
    conv_[a|b]([x_a | x_b]).  Cached like packed_cat."""
    ver = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in mods)
    cache = mods[0].__dict__.setdefault("_codd_packed_cat", {})
    key = ("in",) + tuple(id(m) for m in mods)
    ent = cache.get(key)
    if ent is None or ent[0] != ver:
        w = torch.cat([m.weight.detach() for m in mods], 1)
        b = sum(m.bias.detach() for m in mods)
        ent = cache[key] = (ver, ops.PackedConv(w, b), tuple(mods))
    return ent[1]


def packed_dual(pairs):
    """One PackedConv holding TWO tap sets over the same input (ops.conv_gate, dil2): ``pairs`` = [(conv_a1, conv_a2),
    (conv_b1, conv_b2), ...]; output channels are the concatenation a | b | ..., the weight rows of every conv*1
    (small dilation) come first, then those of conv*2: [cout, cin, 2k, k]; biases added.  conv1(x) + conv2(x) in one
    launch.  Cached like packed_cat."""
    mods = [m for pr in pairs for m in pr]
    ver = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in mods)
    cache = mods[0].__dict__.setdefault("_codd_packed_cat", {})
    key = ("dual",) + tuple(id(m) for m in mods)
    ent = cache.get(key)
    if ent is None or ent[0] != ver:
        w = torch.cat([torch.cat([a.weight.detach(), b.weight.detach()], 2) for a, b in pairs], 0)
        bias = torch.cat([a.bias.detach() + b.bias.detach() for a, b in pairs], 0)
        ent = cache[key] = (ver, ops.PackedConv(w, bias), tuple(mods))
    return ent[1]


# ------------------------------------------------------------------------------------- encoder
class ResidualBlock(nn.Module):
    """reference blocks/extractor.py:9-58 (norm_fn='instance': no affine parameters)."""

    def __init__(self, in_planes, planes, norm_fn="instance", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.downsample = None if stride == 1 else nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride))

    def run(self, x, xs=None):
        """-> (block output fp32, the same as split-bf16 records | None).  In the split / bf16 precision modes the
        InstanceNorm apply passes write the following convolution's input records themselves (ops.instnorm xs_out):
        norm -> ReLU -> records for conv2, and norm -> ReLU -> + x -> ReLU -> fp32 + records for the block output --
        no re-layout launches and no separate add (9 -> 6 launches per block)."""
        B, _, H, W = x.shape
        sy = self.conv1.stride[0]
        Ho, Wo, C = (H + 2 - 3) // sy + 1, (W + 2 - 3) // sy + 1, self.conv1.out_channels
        y1 = ops.split_buffer((id(self), "y1"), B, C, Ho, Wo, 1, x.device) if FUSE_NORM_RECORDS else None
        if y1 is None:
            y = ops.instnorm(cv(self.conv1, x), relu=True)
            y = ops.instnorm(cv(self.conv2, y), relu=True)
            if self.downsample is not None:
                x = ops.instnorm(cv(self.downsample[0], x), relu=False)
            return ops.add_relu(x, y, relu=True), None
        ops.instnorm(cv(self.conv1, x, xs=xs), relu=True, xs_out=y1, want_fp32=False)
        t = cv(self.conv2, None, xs=y1)
        if self.downsample is not None:
            x = ops.instnorm(cv(self.downsample[0], x), relu=False)
        out_xs = ops.split_buffer((id(self), "out"), B, C, Ho, Wo, 1, x.device)
        return ops.instnorm(t, relu=True, res=x, res_relu=True, xs_out=out_xs), out_xs


class BasicEncoder(nn.Module):
    """reference blocks/extractor.py:119-199."""

    def __init__(self, output_dim=128, norm_fn="instance", dropout=0.0):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64), ResidualBlock(64, 64))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, stride=2), ResidualBlock(96, 96))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, stride=2), ResidualBlock(128, 128))
        self.conv2 = nn.Conv2d(128, output_dim, 1)

    def forward(self, x):
        with ops.coresident(FNET_CORESIDENT):
            return self._forward(x)

    def _forward(self, x):
        t = cv(self.conv1, x)
        B, C, H, W = t.shape
        xs = ops.split_buffer((id(self), "stem"), B, C, H, W, 1, x.device) if FUSE_NORM_RECORDS else None
        x = ops.instnorm(t, relu=True, xs_out=xs)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x, xs = blk.run(x, xs)
        return cv(self.conv2, x, xs=xs)


# ------------------------------------------------------------------------------------- update block
class ConvGRU(nn.Module):
    """reference blocks/gru.py:9-35."""

    def __init__(self, hidden_dim=128, dilation=4):
        super().__init__()
        for g in "zrq":
            setattr(self, f"conv{g}1", nn.Conv2d(hidden_dim, hidden_dim, 3, padding=1))
            setattr(self, f"conv{g}2", nn.Conv2d(hidden_dim, hidden_dim, 3, dilation=dilation, padding=dilation))

    def convs_zr(self, h):
        """The two z|r gate convolutions (each 128 -> 256: convz* and convr* stacked)."""
        return (ops.conv2d(h, packed_cat((self.convz1, self.convr1)), pad=1),
                ops.conv2d(h, packed_cat((self.convz2, self.convr2)), pad=4, dil=4))


def _head(cout):
    return nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, cout, 1))


class BasicUpdateBlock(ops.RuntimeState, nn.Module):
    """reference raft3d.py:44-106."""

    def __init__(self, hidden_dim=128, input_dim=128):
        super().__init__()
        self.gru = ConvGRU(hidden_dim)
        self.corr_enc = nn.Sequential(nn.Conv2d(196, 256, 3, padding=1), nn.ReLU(inplace=True),
                                      nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True),
                                      nn.Conv2d(256, 384, 1))
        self.flow_enc = nn.Sequential(nn.Conv2d(9, 128, 7, padding=3), nn.ReLU(inplace=True), nn.Conv2d(128, 384, 1))
        self.ae = _head(32)
        self.delta = _head(3)
        self.weight = _head(3)
        self.mask = _head(576)

    def zr_convs(self, net):
        """Fork the two z|r gate convolutions of the NEXT update: they depend on the hidden state
        only, so they run beside this update's head convolution and the (VALU-bound) Gauss-Newton
        step instead of competing with the next correlation-encoder chain."""
        g = self.gru
        fk = self._forks(net.device)[1]
        ns = self._split(net)
        if self._gates_fused(net) and getattr(self, "_fused_now", False):
            # ONE launch: the 3x3 and the dilated 3x3 z|r convolutions as two tap sets over one input tile, result in
            # the channel-quad layout the gate epilogue of the merged encoder-head convolution reads
            B, _, h, w = net.shape
            t12 = ops.c4_buffer((id(self), "t12"), B, 256, h, w, net.device)
            pc = packed_dual(((g.convz1, g.convz2), (g.convr1, g.convr2)))
            fk.run(0, lambda: ops.conv_gate(pc, ns, 1, pad=4, dil=4, dil2=1, out=t12))
            return t12, None
        t2 = fk.run(0, lambda: ops.conv2d(net, packed_cat((g.convz2, g.convr2)), pad=4, dil=4, xs=ns))
        t1 = fk.run(1, lambda: ops.conv2d(net, packed_cat((g.convz1, g.convr1)), pad=1, xs=ns))
        return t1, t2

    def _gates_fused(self, net):
        return FUSE_GATES and MERGE_ENC_HEADS and ops.CONV_PRECISION in ("split", "split16", "bf16", "fp16")

    def _h4(self, net):
        """The hidden state in the channel-quad layout (ops.C4Tensor) of the gate epilogues: written by the q-gate
        epilogue together with the records; only the initial state (context network output) needs a copy."""
        c = getattr(self, "_h4c", None)
        if c is None or c[0] is not net:
            B, _, h, w = net.shape
            c = self._h4c = (net, ops.to_c4(net, ops.c4_buffer((id(self), "h4"), B, 128, h, w, net.device)))
        return c[1]

    def _ctx4(self, inp):
        """The context stream in the channel-quad layout of the gate epilogues: one copy per frame (the same tensor
        object enters all updates of a frame)."""
        if isinstance(inp, ops.C4Tensor):
            return inp
        c = getattr(self, "_ctx4c", None)
        if c is None or c[0] is not inp:
            B, C, h, w = inp.shape
            c = self._ctx4c = (inp, ops.to_c4(inp, ops.c4_buffer((id(self), "ctx4"), B, C, h, w, inp.device)))
        return c[1]

    def _split(self, t):
        """The split-bf16 form (border 4: serves the 3x3 and the dilated 3x3 convolutions) of the hidden state, shared
        by all of its consumers -- the forked z|r convolutions of the next update and this update's head convolution.
        gru_gate_q_xs writes it together with the fp32 state (``_xs`` = (tensor, its split form)); only the initial
        state (context network output) needs a re-layout pass, into the same persistent tensor."""
        c = getattr(self, "_xs", None)
        if c is None or c[0] is not t:
            hb = ops.split_buffer((id(self), "net"), t.shape[0], 128, t.shape[2], t.shape[3], 4, t.device)
            c = self._xs = (t, ops.split_input(t, border=4, out=hb) if hb is not None else None)
        return c[1]

    def input_buffers(self, like):
        """(corr, minfo) split-bf16 input tensors of the two encoder chains (persistent, zero-bordered; borders 1 and
        3 for the 3x3 / 7x7 first convolutions) that ops.raft_geometry_lookup writes directly; (None, None) outside the
        split / bf16 precision modes."""
        B, _, h, w = like.shape
        return (ops.split_buffer((id(self), "corr_in"), B, 196, h, w, 1, like.device),
                ops.split_buffer((id(self), "minfo_in"), B, 9, h, w, 3, like.device))

    def _forks(self, dev):
        if getattr(self, "_fk", None) is None or self._fk[0].dev != dev:
            self._fk = (ops.Fork(dev, 3), ops.Fork(dev, 2))
            self._fk[0].inline, self._fk[1].inline = not LOOP_FORK_ENC, not LOOP_FORK_ZR
        return self._fk

    def run(self, net, inp, corr, minfo, need_mask, zr=None, prefetch_next=False, fuse_heads=False):
        """One update (reference raft3d.py:92-106).  Launch schedule: the correlation encoder chain,
        the flow encoder chain and the two z|r gate convolutions only depend on data available at
        the start of the update, so they are forked onto side streams; likewise the two q
        convolutions and the three small 1x1 heads.  A 72x120 map gives 576-block grids (2.25
        workgroups per CU): one launch at a time leaves a quarter of the CUs idle in its last
        round, concurrent launches fill them.  ``zr`` = z|r convolutions already forked by the
        previous update (``prefetch_next``); returns (net, mask, ae, delta, weight, zr_next)."""
        g = self.gru
        fk, fkz = self._forks(net.device)
        # gates as convolution epilogues: only the record-input form of the call (the hot path, RAFT3D.forward); a
        # call with explicit corr / minfo tensors returns the fp32 NCHW state and keeps the separate gate kernels
        self._fused_now = corr is None and self._gates_fused(net)
        if zr is not None and (zr[1] is None) != self._fused_now:
            zr = None  # (forked by a call of the other form)

        # conv -> conv links: the producer writes its result as split-bf16 records straight into the consumer's input
        # tensor (persistent, zero-bordered: ops.split_buffer), so neither an fp32 tensor nor a re-layout pass exists
        # between them (None outside the split / bf16 precision modes: plain fp32 tensors then)
        def sb(name, C, border):
            return ops.split_buffer((id(self), name), net.shape[0], C, net.shape[2], net.shape[3], border, net.device)

        cin, min_ = self.input_buffers(net) if corr is None else (None, None)

        def corr_chain():
            s0, s2 = sb("corr_enc0", 256, 1), sb("corr_enc2", 256, 0)
            if s0 is None:
                c = cv(self.corr_enc[0], corr, act="relu")
                c = cv(self.corr_enc[2], c, act="relu")
                return cv(self.corr_enc[4], c)
            cv(self.corr_enc[0], corr, act="relu", xs=cin, xs_out=s0)
            cv(self.corr_enc[2], None, act="relu", xs=s0, xs_out=s2)
            return cv(self.corr_enc[4], None, xs=s2)

        def flow_chain():
            s0 = sb("flow_enc0", 128, 0)
            if s0 is None:
                return cv(self.flow_enc[2], cv(self.flow_enc[0], minfo, act="relu"))
            cv(self.flow_enc[0], minfo, act="relu", xs=min_, xs_out=s0)
            return cv(self.flow_enc[2], None, xs=s0)

        if zr is None:
            zr = self.zr_convs(net)
        t1, t2 = zr
        enc = sb("enc_cat", 384, 0)
        if enc is not None and MERGE_ENC_HEADS:
            # the two encoder chains end in 1x1 convolutions to the same 384 gate-input channels, which the gates only
            # ever use summed (with the context stream ``inp``): their inputs are written side by side into ONE split
            # tensor [corr_enc[2] out 256 | flow_enc[0] out 128] and ONE 1x1 convolution with the input-concatenated
            # weights (+ res1 = inp) produces inp + cor + mot -- one launch and two 13 MB tensors less per update,
            # and the gate kernels read one stream instead of three
            def flow7():
                with ops.coresident(ENC_CORESIDENT in (1, 2)):
                    cv(self.flow_enc[0], minfo, act="relu", xs=min_, xs_out=enc, xs_out_coff=256)
            fk.run(2, flow7)
            with ops.coresident(ENC_CORESIDENT == 2):
                cv(self.corr_enc[0], corr, act="relu", xs=cin, xs_out=sb("corr_enc0", 256, 1))
            cv(self.corr_enc[2], None, act="relu", xs=sb("corr_enc0", 256, 1), xs_out=enc, xs_out_coff=0)
            fk.join()
            if self._fused_now:
                # gates as epilogues: the merged 1x1 convolution adds the context stream (``inp`` = its channel-quad
                # copy, RAFT3D.forward) and the z|r convolutions' result and writes z, r * h (records) and q's input
                # stream; the dual-tap-set q convolution ends in the state update -- 4 launches per update less
                B, _, h, w = net.shape
                h4 = self._h4(net)
                zq = ops.c4_buffer((id(self), "zq"), B, 256, h, w, net.device)
                rs, hb = sb("rh", 128, 4), sb("net", 128, 4)
                fkz.join()
                ops.conv_gate(packed_cat_in((self.corr_enc[4], self.flow_enc[2])), enc, 2, out=zq, res1=self._ctx4(inp), res2=t1,
                              post=h4, xs_out=rs)
                ops.conv_gate(packed_dual(((g.convq1, g.convq2),)), rs, 3, pad=4, dil=4, dil2=1, out=h4, res1=zq, post=h4,
                              xs_out=hb)
                net = self._h4c[0]  # (the fp32 NCHW state is not kept: h4 / hb ARE the state; same token object)
                self._xs = (net, hb)
                return self._heads(net, need_mask, prefetch_next, fuse_heads, sb, fk)
            inp = ops.conv2d(None, packed_cat_in((self.corr_enc[4], self.flow_enc[2])), xs=enc, res1=inp)
            cor = mot = None
        else:
            mot = fk.run(2, flow_chain)
            cor = corr_chain()
            fk.join()
        fkz.join()
        rs, hb = sb("rh", 128, 4), sb("net", 128, 4)
        if rs is None:
            zr_g, rh = ops.gru_gate_zr(t1, t2, inp, cor, mot, net)
            q2 = fk.run(0, lambda: cv(g.convq2, rh))
            q1 = cv(g.convq1, rh)
            fk.join()
            net = ops.gru_gate_q(q1, q2, inp, cor, mot, zr_g, net)
        else:  # the gates write r*h / the new state directly in the following convolutions' input form
            z = ops.gru_gate_zr_xs(t1, t2, inp, cor, mot, net, rs)
            q2 = fk.run(0, lambda: cv(g.convq2, None, xs=rs))
            q1 = cv(g.convq1, None, xs=rs)
            fk.join()
            net = ops.gru_gate_q_xs(q1, q2, inp, cor, mot, z, net, hb)
            self._xs = (net, hb)
        return self._heads(net, need_mask, prefetch_next, fuse_heads, sb, fk)

    def _heads(self, net, need_mask, prefetch_next, fuse_heads, sb, fk):
        zr_next = None
        # the four 3x3 head convs share their input: one 768/1024-channel conv; the mask head (576
        # up-sampling weights) is only consumed after the last iteration (raft3d.py:267-273)
        heads = (self.ae[0], self.delta[0], self.weight[0]) + ((self.mask[0],) if need_mask else ())
        hs = sb("hid", 256 * len(heads), 0)
        if hs is None:
            hid = ops.conv2d(net, packed_cat(heads), pad=1, act="relu")
            sl = lambda i: Slice(hid, 256 * i, 256)
        else:  # the 768 / 1024 hidden channels only ever exist as the four 1x1 heads' split-form input
            # (fused gates: ``net`` is only the state's identity token, its fp32 values are the INITIAL state)
            ops.conv2d(None if self._fused_now else net, packed_cat(heads), pad=1, act="relu",
                       xs=self._split(net), xs_out=hs)
            sl = lambda i: None
        if prefetch_next:
            # forked AFTER the head convolution is enqueued: the side streams wait for it, so the next update's z|r
            # convolutions (MFMA + LDS) run beside the Gauss-Newton builder (VALU only, 27 KB LDS: both fit a CU)
            # instead of sharing the matrix pipes with the head convolution
            zr_next = self.zr_convs(net)
        if hs is not None and fuse_heads:
            # the ae / delta / weight 1x1 heads run inside the Gauss-Newton record packing (ops.se3_gn_step_heads)
            # the mask head's 1x1 convolution is only read after the loop (cvx_upsample): beside the Gauss-Newton step,
            # joined by the caller (RAFT3D.forward)
            mask = fk.run(1, lambda: cv(self.mask[2], None, xs=hs, xs_coff=768)) if need_mask else None
            return net, mask, None, None, None, zr_next, hs
        delta = fk.run(0, lambda: cv(self.delta[2], sl(1), xs=hs, xs_coff=256))
        weight = fk.run(1, lambda: cv(self.weight[2], sl(2), act="sigmoid", xs=hs, xs_coff=512))
        mask = fk.run(2, lambda: cv(self.mask[2], sl(3), xs=hs, xs_coff=768)) if need_mask else None
        ae = cv(self.ae[2], sl(0), xs=hs, xs_coff=0)
        fk.join()
        return net, mask, ae, delta, weight, zr_next, None

    def head_matrix(self):
        """The three 1x1 heads (ae 32, delta 3, weight 3 rows x 256) packed as the MFMA A operands of
        codd_se3_gn_step_heads (include/codd_hip.h) + their 38 biases; cached on the module per parameter version."""
        mods = (self.ae[2], self.delta[2], self.weight[2])
        ver = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in mods)
        ver = ver + (ops.CONV_PRECISION in ("fp16", "split16"),)  # (fp16 records need fp16 A operands)
        c = self.__dict__.get("_codd_head_matrix")
        if c is None or c[0] != ver:
            Wm = torch.cat([m.weight.detach().reshape(m.weight.shape[0], 256) for m in mods], 0).float()
            bm = torch.cat([m.bias.detach() for m in mods], 0).float().contiguous()
            c = self.__dict__["_codd_head_matrix"] = (ver, pack_head_matrix(Wm, f16=ver[-1]), bm)
        return c[1], c[2]


def pack_head_matrix(Wm, f16=False):
    """[38, 256] fp32 (ae rows 0..31, delta 32..34, weight 35..37) -> [32, 2, 64, 8] bf16 A-operand blocks (hi | lo
    planes); ``f16``: the hi plane holds IEEE fp16 instead (hidden records of terms = 16; the lo plane is unused)."""
    dev = Wm.device
    full = torch.zeros(48, 768, device=dev)  # rows: 32 ae | delta 3 | weight 3 | 10 zero; columns: hidden channel
    full[:32, :256] = Wm[:32]
    full[32:35, 256:512] = Wm[32:35]
    full[35:38, 512:768] = Wm[35:38]
    lane = torch.arange(64, device=dev)
    row, kg = lane % 16, lane // 16
    k8 = torch.arange(8, device=dev)
    blocks = []
    for t in range(2):
        for s_ in range(8):
            cols = (32 * s_ + 8 * kg)[:, None] + k8[None]
            blocks.append(full[(16 * t + row)[:, None], cols])
    for s_ in range(16):
        cols = (256 + 32 * s_ + 8 * kg)[:, None] + k8[None]
        blocks.append(full[(32 + row)[:, None], cols])
    v = torch.stack(blocks, 0)  # [32, 64, 8] fp32
    if f16:
        hi = v.half()
        lo = (v - hi.float()).half()  # (read only with split-fp16 records, terms = 48)
        return torch.stack([hi, lo], 1).contiguous().view(torch.bfloat16)  # (16-bit payload; dtype is a label)
    hi = v.bfloat16()
    lo = (v - hi.float()).bfloat16()
    return torch.stack([hi, lo], 1).contiguous()  # [32, 2, 64, 8]


@register
class RAFT3D(ops.RuntimeState, nn.Module):
    """reference raft3d.py:140-280 (inference branch)."""

    def __init__(self, cnet_cfg=None):
        super().__init__()
        self.hidden_dim = self.context_dim = 128
        self.corr_levels, self.corr_radius = 4, 3
        self.fnet = BasicEncoder(output_dim=128, norm_fn="instance")
        if cnet_cfg is None:
            raise NotImplementedError("the reference's FPN fallback is undefined upstream; pass cnet_cfg")
        cfg = dict(cnet_cfg)
        self.cnet = nn.Sequential(build_backbone(cfg),
                                  ResizeConcatConv(cfg["extra"]["stage4"]["num_channels"], 128 * 4))
        self.update_block = BasicUpdateBlock(hidden_dim=128)

    def context(self, image):
        # exact-fp32 convolutions (ops.stage policy): the context features enter every one of the 16 updates, and their
        # split-bf16 rounding (1e-5 relative) was what moved near-camera points across pixel boundaries in the splat --
        # 63 flipped pixels at 640x512 with the context network on split-bf16, none with it on fp32
        with ops.stage("context"):
            return self.cnet[1](self.cnet[0](image))

    # -- side-stream prefetch ------------------------------------------------------------------
    # fnet(image) (needed before the correlation pyramid) and cnet(image) (only needed by the NEXT
    # frame, raft3d.py:278) depend on nothing but the current left image, so they are issued on two
    # side HIP streams before the stereo network starts and joined where they are consumed.  Their
    # ~200 small launches fill the CUs that HITNet's coarse levels and the GRU loop's 576-block
    # convolutions leave idle; under stream capture this becomes two parallel branches of the frame
    # graph.
    def prefetch(self, image, state=None):
        """``state``: the recurrent state of the sequence; when it holds the previous frame's feature map the all-pairs
        correlation pyramid (reference blocks/corr.py:28-45: a function of the two feature maps only) is built on the
        fnet side stream as well, i.e. beside the stereo network instead of in front of the update loop.
        (Issue orders that were measured and dropped -- stereo network first, context network after the feature encoder or in
        portions inside the update loop, stream priorities: DESIGN.md findings 46-48, 50.)"""
        if ops.Fork.serial:
            self._pending = None
            return
        dev = image.device
        if getattr(self, "_side", None) is None or self._side[0].device != dev:
            self._side = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        cur = torch.cuda.current_stream(dev)
        out = {}
        for key, stream, fn in (("fmap", self._side[0], self.fnet), ("netinp", self._side[1], self.context)):
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                out[key] = fn(image)
                if key == "fmap" and state is not None and "memory" in state and state.get("raft_feat") is not None:
                    out["pyr"] = (state["raft_feat"], ops.allpairs_corr(state["raft_feat"], out["fmap"]))
                    if PRELOOP_SIDE and state.get("raft_netinp") is not None:
                        out["pre"] = self._preloop(state["raft_netinp"])
        self._pending = out

    def _preloop(self, net_inp):
        """The launches in front of the first update that only read the PREVIOUS frame's state (the context network's
        output: reference raft3d.py:205-207,278): hidden-state / context split, their layouts for the gate epilogues,
        the identity field and the first update's z|r convolutions.  Issued behind the correlation pyramid on the fnet
        side stream they leave the frame's critical path (stereo -> update loop -> fusion); same launches, same bits."""
        B, _, h, w = net_inp.shape
        ub = self.update_block
        net, inp = ops.context_split(net_inp)
        pre = dict(src=net_inp, net=net, inp=inp, T=ops.se3_identity(B, h, w, net_inp.device), zr=None)
        if ub._gates_fused(net) and ub.input_buffers(net)[0] is not None:
            ub._fused_now = True
            ub._h4(net), ub._ctx4(inp)
            fkz = ub._forks(net_inp.device)[1]
            prev, fkz.inline = fkz.inline, True  # (one launch: on THIS side stream, not on a fork of it)
            try:
                pre["zr"] = ub.zr_convs(net)
            finally:
                fkz.inline = prev
        return pre

    def _join(self, key, dev):
        pend = getattr(self, "_pending", None)
        if not pend or key not in pend:
            return None
        if not getattr(self, "_nowait", False):  # pipelined runner: the tensors are already complete
            torch.cuda.current_stream(dev).wait_stream(self._side[0] if key == "fmap" else self._side[1])
        return pend.pop(key)

    def forward(self, image_curr, depth_prev, depth_curr, intrinsics, state, outputs, iters=12, train_mode=False):
        dev = image_curr.device
        if "memory" not in state:
            fm, ni = self._join("fmap", dev), self._join("netinp", dev)
            state["raft_feat"] = fm if fm is not None else self.fnet(image_curr)
            state["raft_netinp"] = ni if ni is not None else self.context(image_curr)
            return
        B, _, H, W = image_curr.shape
        h, w = H // 8, W // 8
        K = [np.float32(v) for v in intrinsics]
        K8 = [float(v / np.float32(8.0)) for v in K]
        fmap_prev, net_inp = state["raft_feat"], state["raft_netinp"]
        pend = getattr(self, "_pending", None) or {}
        pl = pend.pop("pre", None)
        if pl is not None and pl["src"] is not net_inp:
            pl = None
        T = pl["T"] if pl is not None else ops.se3_identity(B, h, w, image_curr.device)
        pre = pend.pop("pyr", None)  # (fmap_prev it was built from, pyramid): made on the fnet side stream by prefetch
        fmap_curr = self._join("fmap", dev)
        if fmap_curr is None:
            fmap_curr = self.fnet(image_curr)
        pyr = pre[1] if pre is not None and pre[0] is fmap_prev else ops.allpairs_corr(fmap_prev, fmap_curr)
        net, inp = (pl["net"], pl["inp"]) if pl is not None else ops.context_split(net_inp)
        if pl is not None and pl.get("depth_prev") is depth_prev:
            d1 = pl["d1"]
        else:
            d1 = ops.subsample(depth_prev.contiguous(), 3, 3, 8)  # depth[:, 3::8, 3::8] (raft3d.py:213-216)
        d2 = ops.subsample(depth_curr.contiguous(), 3, 3, 8)
        mask = weight = None
        zr = pl["zr"] if pl is not None else None
        cxs, mxs = self.update_block.input_buffers(net)
        for it in range(iters):
            # projection + pyramid lookup, one launch; in the split-bf16 modes its results are written straight into
            # the encoder convolutions' input tensors (corr = minfo = None then)
            xyz, minfo, corr = ops.raft_geometry_lookup(T, d1, d2, K8, pyr, minfo_xs=mxs, corr_xs=cxs)
            net, mask, ae, delta, weight, zr, hid = self.update_block.run(
                net, inp, corr, minfo, need_mask=it == iters - 1, zr=zr, prefetch_next=it < iters - 1, fuse_heads=True)
            if hid is not None:  # split-bf16 path: heads + record packing in one launch
                weight = ops.se3_gn_step_heads(T, hid, *self.update_block.head_matrix(), xyz, d1, K8, radius=32)
            else:
                ops.se3_gn_step(T, ae, xyz, delta, weight, d1, K8, radius=32)
        self.update_block._forks(dev)[0].join()  # the mask head's 1x1 convolution (forked beside the last Gauss-Newton step)
        T_up, outputs["weight"] = ops.cvx_upsample_se3_weight(T, weight.contiguous(), mask)  # one pass over the mask
        outputs["Ts"] = T_up
        # reference raft3d.py:268-270: the induced 2-D flow + inverse-depth change of the up-sampled field
        outputs["flow2d_est_induced"] = ops.induced_flow(T_up, depth_prev, [float(v) for v in K])
        state["raft_feat"] = fmap_curr
        ni = self._join("netinp", dev)
        state["raft_netinp"] = ni if ni is not None else self.context(image_curr)


@register
class Motion(ops.RuntimeState, nn.Module):
    """reference motion.py:48-209."""

    def __init__(self, raft3d=None, ds_scale=4, iters=16, loss=None):
        super().__init__()
        self.ds_scale = ds_scale
        self.iters = iters
        self.raft3d = MODELS.build(raft3d)
        self.loss = build_loss(loss) if loss is not None else None

    def prefetch(self, left_img, state=None, img_metas=None):
        """Issue the image-only parts of the motion stage (fnet [+ the correlation pyramid], cnet) on side streams, and
        behind them the state-only ones: the previous frame's depth map and its 1/8 sub-sampling (motion.py:154-159,
        raft3d.py:213-216) -- same launches as in ``forward``, off the frame's critical path."""
        self.raft3d.prefetch(left_img, state)
        pend = getattr(self.raft3d, "_pending", None)
        if PRELOOP_SIDE and pend and "pre" in pend and img_metas is not None and len(state.get("memory", ())) == 3:
            disp_prev = state["memory"][2]
            bf = self._bf(img_metas)
            with torch.cuda.stream(self.raft3d._side[0]):
                depth_prev = ops.disp_to_depth(disp_prev.contiguous(), bf)
                pend["pre"].update(disp_src=disp_prev, bf=bf, depth_prev=depth_prev,
                                   d1=ops.subsample(depth_prev.contiguous(), 3, 3, 8))

    @staticmethod
    def _bf(img_metas):
        fx = np.float32(img_metas[0]["intrinsics"][0])
        return float(np.float32(np.float32(BF_DEFAULT) / fx) * fx)  # depth_scale * fx in fp32 (motion.py:154-159)

    def forward(self, state, outputs, img_metas, train_mode=False, **kwargs):
        img_curr = outputs["left_img"]
        if "memory" not in state:
            self.raft3d(img_curr, None, None, None, state, outputs, train_mode=train_mode)
            return
        intr = [np.float32(v) for v in img_metas[0]["intrinsics"]]
        bf = self._bf(img_metas)
        K = [float(v) for v in intr]
        img_prev, feat_prev, disp_prev = state["memory"]
        disp_curr = outputs["pred_disp"]
        pl = (getattr(self.raft3d, "_pending", None) or {}).get("pre")
        if pl is not None and pl.get("disp_src") is disp_prev and pl.get("bf") == bf:
            depth_prev = pl["depth_prev"]  # (made on the feature encoder's side stream: joined in RAFT3D.forward)
        else:
            depth_prev = ops.disp_to_depth(disp_prev.contiguous(), bf)  # [B,H,W]
        depth_curr = ops.disp_to_depth(disp_curr, bf).squeeze(1)
        self.raft3d(img_curr, depth_prev, depth_curr, intr, state, outputs, iters=self.iters, train_mode=train_mode)
        T_up = outputs["Ts"]
        B, H, W = depth_prev.shape
        # 1/ds-resolution feature warp with T, depth sampled at [o::ds, o::ds] and K / ds -- independent of the
        # full-resolution warp below: forked onto a side stream (two ~95-us launch chains side by side)
        ds, o = self.ds_scale, self.ds_scale // 2 - 1
        Kd = [float(v / np.float32(ds)) for v in intr]
        fk = self.__dict__.get("_fk")
        if fk is None or fk.dev != T_up.device:
            fk = self.__dict__["_fk"] = ops.Fork(T_up.device, 1)
        feat_warp, _ = fk.run(0, lambda: ops.splat(T_up, depth_prev, feat_prev, None, False, H // ds, W // ds, o, o, ds,
                                                   Kd, 4.0))
        # full-resolution warp of [img_prev | induced flow | confidence]; depth -> disparity fused
        warped, disp_warp = ops.splat(T_up, depth_prev, img_prev, outputs["weight"], True, H, W, 0, 0, 1, K, 2.0, bf=bf)
        fk.join()
        state["memory"] = [warped[:, :3], feat_warp, warped[:, 6:], disp_warp, warped[:, 3:6]]
