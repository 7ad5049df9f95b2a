"""HITNetMF on MI355X: parameter containers with the reference's attribute names (so published
state dicts load by key) whose forward is a schedule of C-ABI kernel launches.

reference: model/stereo/hitnet/{hitnet,backbone,initialization,propagation}.py.
Registry names, constructor kwargs and the ``stereo_matching`` contract follow
configs/models/codd.py:20-39 and model/codd.py:94-96.

Differences from the reference's op schedule (results identical up to fp32 rounding):
  * left and right images go through the U-Net as one batch of 2;
  * torch.cat is never executed: producers write into channel slices of the consumer's input
    buffer, or the conv kernel reads two sources;
  * the tile cost volume is never materialised (fused arg-min), TileWarping's three offsets,
    the PixelUnshuffle and the ||fea_l||_1 feature are one kernel, for both hypothesis sets.
"""

import torch
import torch.nn as nn

from . import ops
from .ops import Slice
from .registry import MODELS, build_backbone, build_loss, register


def _lrelu():
    return nn.LeakyReLU(0.2, inplace=True)


def packed(m, deconv=False, cout_keep=None):
    """PackedConv of an nn.Conv2d / nn.ConvTranspose2d, rebuilt when its parameters change.  The pack lives ON the
    module (``m._codd_packed``), so it is freed with the model and can never be served to another module."""
    ver = (m.weight.data_ptr(), m.weight._version, None if m.bias is None else m.bias._version)
    cache = m.__dict__.setdefault("_codd_packed", {})
    ent = cache.get(cout_keep)
    if ent is None or ent[0] != ver:
        ent = cache[cout_keep] = (ver, ops.PackedConv(m.weight, m.bias, deconv, cout_keep))
    return ent[1]


def invalidate_packed(module):
    """Drop every cached pack below ``module`` (apis.load_checkpoint calls this after loading a state dict)."""
    for m in module.modules():
        m.__dict__.pop("_codd_packed", None)
        m.__dict__.pop("_codd_packed_cat", None)


def roll(owner, key, specs, residual=False):
    """ops.PackedRoll (one rolling-window launch: csrc/conv_roll.hip) of the nn.Conv2d modules in ``specs`` =
    [(conv, act), ...] (one 3x3, two 3x3 or 1x1 -> 3x3); cached on ``owner`` per parameter version like ``packed``."""
    mods = [m for m, _ in specs]
    ver = tuple((m.weight.data_ptr(), m.weight._version, None if m.bias is None else m.bias._version) for m in mods)
    cache = owner.__dict__.setdefault("_codd_packed_cat", {})
    ent = cache.get(("roll", key))
    if ent is None or ent[0] != ver:
        ent = cache[("roll", key)] = (ver, ops.PackedRoll([dict(w=m.weight, b=m.bias, act=a) for m, a in specs],
                                                          residual=residual), tuple(mods))
    return ent[1]


def cv(m, x, x2=None, act="none", **kw):
    """Run nn.Conv2d ``m`` through the HIP conv kernel using the module's own geometry."""
    return ops.conv2d(x, packed(m), x2=x2, stride=tuple(m.stride), pad=tuple(m.padding), dil=tuple(m.dilation),
                      act=act, **kw)


# ------------------------------------------------------------------------------------- backbone
def _down(i, o):
    return nn.Sequential(nn.Conv2d(i, o, 4, 2, 1), _lrelu(), nn.Conv2d(o, o, 3, 1, 1), _lrelu())


def _up(i, o):
    return nn.Sequential(nn.ConvTranspose2d(i, o, 2, 2, 0), _lrelu())


def _merge(i, o):
    return nn.Sequential(nn.Conv2d(i, o, 1), _lrelu(), nn.Conv2d(o, o, 3, 1, 1), _lrelu(),
                         nn.Conv2d(o, o, 3, 1, 1), _lrelu())


@register
class HITUNet(nn.Module):
    """reference backbone.py:42-88."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(3, 16, 3, 1, 1), _lrelu())
        self.down1 = _down(16, 16)
        self.down2 = _down(16, 24)
        self.down3 = _down(24, 24)
        self.down4 = nn.Sequential(_down(24, 32), nn.Conv2d(32, 32, 3, 1, 1), _lrelu(),
                                   nn.Conv2d(32, 32, 3, 1, 1), _lrelu())
        self.up4 = _up(32, 24)
        self.up3 = _up(24, 24)
        self.up2 = _up(24, 16)
        self.up1 = _up(16, 16)
        self.merge4 = _merge(48, 24)
        self.merge3 = _merge(48, 24)
        self.merge2 = _merge(32, 16)
        self.merge1 = _merge(32, 16)

    def stages(self, x):
        """Generator over the feature pyramid in production order (1/16, 1/8, 1/4, 1/2, 1/1): a consumer may use the
        coarse scales while the decoder still runs (HITNetMF.stereo_matching, pipelined schedule)."""
        def seq(s, t):
            for m in s:
                if isinstance(m, nn.Conv2d):
                    if (m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1)
                            and m.in_channels == m.out_channels and ops.use_roll(m.out_channels, *t.shape)):
                        t = ops.conv_roll(t, roll(m, "single", [(m, "lrelu")]))
                    else:
                        t = cv(m, t, act="lrelu")
            return t

        def up_merge(up, merge, skip, t):
            u = ops.conv2d(t, packed(up[0], deconv=True), act="lrelu")
            if ops.use_roll(merge[0].out_channels, *skip.shape):
                # 1x1 (skip | up) -> 3x3 as one rolling-window launch, the last 3x3 as another
                t = ops.conv_roll(skip, roll(merge, "head", [(merge[0], "lrelu"), (merge[2], "lrelu")]), x2=u)
                return ops.conv_roll(t, roll(merge, "tail", [(merge[4], "lrelu")]))
            t = cv(merge[0], skip, x2=u, act="lrelu")
            t = cv(merge[2], t, act="lrelu")
            return cv(merge[4], t, act="lrelu")

        yield from self._stages(x, seq, up_merge)

    def forward(self, x):
        return list(self.stages(x))

    def _stages(self, x, seq, up_merge):
        x0 = seq(self.conv1, x)
        x1 = seq(self.down1, x0)
        x2 = seq(self.down2, x1)
        x3 = seq(self.down3, x2)
        x4 = seq(self.down4[0], x3)
        x4 = cv(self.down4[1], x4, act="lrelu")
        x4 = cv(self.down4[3], x4, act="lrelu")
        yield x4
        u4 = up_merge(self.up4, self.merge4, x3, x4)
        yield u4
        u3 = up_merge(self.up3, self.merge3, x2, u4)
        yield u3
        u2 = up_merge(self.up2, self.merge2, x1, u3)
        yield u2
        yield up_merge(self.up1, self.merge1, x0, u2)


# ------------------------------------------------------------------------------------- tile init
_LEVELS = ("16x", "8x", "4x", "2x", "1x")


FORK_INIT_LEVELS = False  # (A/B switch; TileInitialization.forward)
STEREO_PIPE = True
# initialisation of scales 1/2 (>= 1) and 1/4 (>= 2) on the decoder's side stream, right behind the decoder stage that
# produces their features, instead of on the caller's stream in front of the propagation step that consumes them: the
# caller's stream (the long pole: ~60 dependent small launches) sheds 12 of them.  Same launches, same bits.  Round 4,
# three alternating runs each: stereo-only 441 -> 460 -> 472 frames/s (0 / 1 / 2), full frame 97.1 -> 97.6 -> 97.7.
PIPE_INIT_SIDE = 2  # (A/B: HITNetMF._stereo_matching_pipelined)
FORK_INIT_FINE = 3  # (A/B: this many of the finest scales on ONE side stream)


@register
class TileInitialization(ops.RuntimeState, nn.Module):
    """reference initialization.py:48-230."""

    def __init__(self, max_disp, fea_c=[16, 16, 24, 24, 32]):
        super().__init__()
        self.maxdisp = max_disp
        c1, c2, c4, c8, c16 = fea_c
        for name, c in zip(_LEVELS, (c16, c8, c4, c2, c1)):
            setattr(self, f"tile_conv{name}", nn.Sequential(nn.Conv2d(c, 16, 4, 4, 0), _lrelu(),
                                                            nn.Conv2d(16, 16, 1, 1, 0), _lrelu()))
        for name, c in zip(_LEVELS, (17, 17, 33, 25, 25)):
            setattr(self, f"tile_fea_dscrpt{name}", nn.Sequential(nn.Conv2d(c, 13, 1), _lrelu()))

    def init_level(self, lvl, fl, fr, feat):
        """One scale: tile features of both views, cost-volume arg-min, tile descriptor -> the 16-channel hypothesis
        Slice at the head of the aug-hypothesis buffer (``feat``: fea_l[lvl - 2] for lvl >= 2)."""
        name = _LEVELS[lvl]
        tc = getattr(self, f"tile_conv{name}")
        pc0, pc1 = packed(tc[0]), packed(tc[2])
        # left / right tile features are independent chains: convolutions of the same depth leave as ONE multi-job launch
        # where both are of the multi-job class (ops.deferred_convs: same parameters, same bits)
        with ops.deferred_convs():
            tl = ops.conv2d(fl, pc0, stride=4, act="lrelu")
            # right features: same weights, stride (4,1) on the image zero-padded 3 px on the right
            tr = ops.conv2d(fr, pc0, stride=(4, 1), pad_tl=(0, 0, 0, 3), act="lrelu")
        with ops.deferred_convs():
            tl = ops.conv2d(tl, pc1, act="lrelu")
            tr = ops.conv2d(tr, pc1, act="lrelu")
        B, _, Ht, Wt = tl.shape
        aug = torch.empty(B, 32 if lvl == 0 else 64, Ht, Wt, device=fl.device, dtype=torch.float32)
        cost = torch.empty(B, 1, Ht, Wt, device=fl.device, dtype=torch.float32)
        ops.tile_costvol_argmin(tl, tr, self.maxdisp // (16 >> lvl), cost, Slice(aug, 0, 3))
        feat = tl if feat is None else feat
        cv(getattr(self, f"tile_fea_dscrpt{name}")[0], cost, x2=feat, act="lrelu", out=Slice(aug, 3, 13))
        return Slice(aug, 0, 16)

    def forward(self, fea_l, fea_r):
        """-> [None, hyps]: the cost volumes are not materialised at inference (the reference only
        consumes them in the training loss, hitnet.py:84-85).  hyps[l] is a 16-channel Slice at
        the head of the aug-hypothesis buffer TilePropagation consumes (32 ch at 1/16, else 64)."""
        def level(lvl):
            return self.init_level(lvl, fea_l[lvl], fea_r[lvl], None if lvl < 2 else fea_l[lvl - 2])

        # the five scales are independent 6-launch chains (the four coarse ones ~10 us launches that leave the chip
        # idle): the coarse scales on side streams beside the finest one
        n = len(_LEVELS)
        if FORK_INIT_FINE and not ops.Fork.serial and getattr(self, "fork_streams", True):
            # ONE branch: the FORK_INIT_FINE finest scales on a side stream beside the coarse scales' initialisation
            # (latency-bound ~10 us launches).  Joined BEFORE returning: the result is a plain list that may be
            # iterated, sliced or indexed in any order (the overlap with the coarse PROPAGATION steps is what
            # HITNetMF._stereo_matching_pipelined does with explicit events).
            fk = self.__dict__.get("_fk")
            if fk is None or fk.dev != fea_l[0].device:
                fk = self.__dict__["_fk"] = ops.Fork(fea_l[0].device, 1)
            fine = fk.run(0, lambda: [level(lvl) for lvl in range(n - FORK_INIT_FINE, n)])
            coarse = [level(lvl) for lvl in range(n - FORK_INIT_FINE)]
            fk.join()
            return [None, coarse + fine]
        if FORK_INIT_LEVELS and not ops.Fork.serial and getattr(self, "fork_streams", True):
            fk = self.__dict__.get("_fk")
            if fk is None or fk.dev != fea_l[0].device:
                fk = self.__dict__["_fk"] = ops.Fork(fea_l[0].device, n - 1)
            hyps = [fk.run(lvl, level, lvl) for lvl in range(n - 1)] + [level(n - 1)]
            fk.join()
        else:
            hyps = [level(lvl) for lvl in range(n)]
        return [None, hyps]


# ------------------------------------------------------------------------------------- propagation
def _convbn(i, o, k, s, p, d):
    return nn.Sequential(nn.Conv2d(i, o, k, s, d if d > 1 else p, d))


class BasicBlock(nn.Module):
    """reference propagation.py:103-121 (no BN)."""

    def __init__(self, c1, c2, s, downsample, p, d):
        super().__init__()
        self.conv1 = nn.Sequential(_convbn(c1, c2, 3, s, p, d), _lrelu())
        self.conv2 = _convbn(c2, c2, 3, 1, p, d)

    def run(self, x):
        """lrelu(conv2(lrelu(conv1(x))) + x): the trailing LeakyReLU of the enclosing Sequential is
        fused into the second conv's epilogue."""
        c1, c2 = self.conv1[0][0], self.conv2[0]
        same3 = all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) for c in (c1, c2))
        if same3 and ops.use_roll(c1.out_channels, *x.shape) and c1.in_channels == c1.out_channels:
            # both convolutions + residual + LeakyReLU in one rolling-window launch (intermediate in LDS)
            return ops.conv_roll(x, roll(self, "block", [(c1, "lrelu"), (c2, "lrelu")], residual=True))
        t = cv(c1, x, act="lrelu")
        return cv(c2, t, res1=x, act="lrelu")


def _resblock(c, d=1):
    return nn.Sequential(BasicBlock(c, c, s=1, p=1, downsample=None, d=d), _lrelu())


class TileUpdate0(nn.Module):
    """reference propagation.py:124-172."""

    def __init__(self, in_c, out_c, hid_c):
        super().__init__()
        self.decrease = nn.Sequential(nn.Conv2d(64, 16, 1), _lrelu())
        self.conv0 = nn.Sequential(nn.Conv2d(in_c, hid_c, 1), _lrelu())
        self.resblock0 = _resblock(32)
        self.resblock1 = _resblock(32)
        self.lastconv = nn.Conv2d(hid_c, out_c, 3, 1, 1)

    def forward(self, fl, fr, hyp):
        aug = hyp.buf  # [hyp 16 | local cv 16]
        w, _ = ops.tile_warp_cost(fl, fr, hyp)
        cv(self.decrease[0], w, act="lrelu", out=Slice(aug, 16, 16))
        t = cv(self.conv0[0], aug, act="lrelu")
        t = self.resblock0[0].run(t)
        t = self.resblock1[0].run(t)
        return [cv(self.lastconv, t, res1=hyp, act="relu_ch0")]


class TileUpdate(nn.Module):
    """reference propagation.py:175-248."""

    def __init__(self):
        super().__init__()
        self.decrease = nn.Sequential(nn.Conv2d(64, 16, 1), _lrelu())
        self.conv0 = nn.Sequential(nn.Conv2d(64, 32, 1), _lrelu())
        self.resblock0 = _resblock(32)
        self.resblock1 = _resblock(32)
        self.lastconv = nn.Conv2d(32, 34, 3, 1, 1)

    def forward(self, fl, fr, hyp, prev):
        aug = hyp.buf  # [cur 16 | cv_cur 16 | up_prev 16 | cv_prev 16]
        up = Slice(aug, 32, 16)
        ops.hyp_upsample(prev, 2.0, up)
        w0, w1 = ops.tile_warp_cost(fl, fr, hyp, up)
        with ops.deferred_convs():  # (two independent 1x1 convolutions: one multi-job launch, same bits)
            cv(self.decrease[0], w0, act="lrelu", out=Slice(aug, 16, 16))
            cv(self.decrease[0], w1, act="lrelu", out=Slice(aug, 48, 16))
        t = cv(self.conv0[0], aug, act="lrelu")
        t = self.resblock0[0].run(t)
        t = self.resblock1[0].run(t)
        upd = cv(self.lastconv, t)
        B, _, h, w = upd.shape
        out = torch.empty(B, 16, h, w, device=upd.device, dtype=torch.float32)
        return [ops.hyp_select(upd, hyp, up, out)]


class PostTileUpdate(nn.Module):
    """reference propagation.py:251-290."""

    def __init__(self, in_c, out_c, hid_c, resblk_num, final=False):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_c, hid_c, 1), _lrelu(), nn.Conv2d(hid_c, hid_c, 3, 1, 1), _lrelu())
        self.resblocks = nn.Sequential(*[_resblock(hid_c, 3 if (i == 1 and not final) else 1)
                                         for i in range(resblk_num)])
        self.lastconv = nn.Conv2d(hid_c, out_c, 3, padding=1)
        self._final = final

    def forward(self, fl, prev):
        if ops.use_roll(self.conv1[0].out_channels, *fl.shape):
            t = ops.conv_roll(fl, roll(self, "head", [(self.conv1[0], "lrelu"), (self.conv1[2], "lrelu")]), x2=prev)
        else:
            t = cv(self.conv1[0], fl, x2=prev, act="lrelu")
            t = cv(self.conv1[2], t, act="lrelu")
        for blk in self.resblocks:
            t = blk[0].run(t)
        if self._final:
            # only channel 0 of the 3-channel output is consumed (propagation.py:372): compute that
            # one, relu(prev_d + out0).
            return ops.conv2d(t, packed(self.lastconv, cout_keep=1), pad=1, res1=Slice(_buf(prev), _off(prev), 1),
                              act="relu")
        return cv(self.lastconv, t, res1=prev, act="relu_ch0")




class FinalTileUpdate(PostTileUpdate):
    """reference propagation.py:293-333."""

    def __init__(self, in_c, out_c, hid_c, resblk_num):
        super().__init__(in_c, out_c, hid_c, resblk_num, final=True)


def _buf(x):
    return x.buf if isinstance(x, Slice) else x


def _off(x):
    return x.coff if isinstance(x, Slice) else 0


def _up1(h):
    B, _, hh, ww = h.shape
    out = torch.empty(B, 16, 2 * hh, 2 * ww, device=h.device, dtype=torch.float32)
    return ops.hyp_upsample(h, 1.0, out)


@register
class TilePropagation(nn.Module):
    """reference propagation.py:336-454 (inference branch)."""

    def __init__(self):
        super().__init__()
        self.tile_update0 = TileUpdate0(32, 16, 32)
        self.tile_update1 = TileUpdate()
        self.tile_update2 = TileUpdate()
        self.tile_update3 = TileUpdate()
        self.tile_update4 = TileUpdate()
        self.tile_update4_1 = PostTileUpdate(40, 16, 32, 4)
        self.tile_update5 = PostTileUpdate(32, 16, 32, 4)
        self.tile_update6 = FinalTileUpdate(32, 3, 16, 2)

    def forward(self, fea_l, fea_r, init):
        h = self.tile_update0(fea_l[0], fea_r[0], init[0])[0]
        for i, upd in enumerate((self.tile_update1, self.tile_update2, self.tile_update3, self.tile_update4), 1):
            h = upd(fea_l[i], fea_r[i], init[i], h)[0]
        r1 = self.tile_update4_1(fea_l[2], h)
        r05 = self.tile_update5(fea_l[3], _up1(r1))
        return self.tile_update6(fea_l[4], _up1(r05))


@register
class HITNetMF(ops.RuntimeState, nn.Module):
    """reference hitnet.py:13-122."""

    def __init__(self, backbone, initialization, propagation, loss=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.tile_init = MODELS.build(initialization)
        self.tile_update = MODELS.build(propagation)
        self.freezed = False
        self.loss = build_loss(loss) if loss is not None else None

    def extract_feat(self, img):
        return self.backbone(img)

    def _stereo_matching_pipelined(self, left_img, right_img):
        """Two-stream schedule of the same launches: the coarse-to-fine propagation only needs scale i's features and
        initialisation at step i, and the finest initialisation only the end of the U-Net decoder.  A side stream runs
        decoder scales 1/4, 1/2, 1/1 and the finest initialisation; the caller's stream runs the coarse
        initialisations and propagation steps (latency-bound ~10 us launches) beside it, waiting for the decoder
        scale-by-scale (events) -- ONE branch, one join."""
        B, dev = left_img.shape[0], left_img.device
        ti, tu = self.tile_init, self.tile_update
        g = self.backbone.stages(ops.batch_pair(left_img, right_img))
        feas = [next(g), next(g)]  # 1/16, 1/8 (on the caller's stream)
        L, R = (lambda i: feas[i][:B]), (lambda i: feas[i][B:])
        rt = self.__dict__.get("_pipe")
        if rt is None or rt[0].device != dev:
            rt = self.__dict__["_pipe"] = (torch.cuda.Stream(device=dev), torch.cuda.Event(), torch.cuda.Event())
        side, ev3, ev2 = rt
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        hyp2 = hyp3 = None
        with torch.cuda.stream(side):
            feas.append(next(g))
            if PIPE_INIT_SIDE >= 2:
                hyp2 = ti.init_level(2, L(2), R(2), L(0))
            ev3.record(side)
            feas.append(next(g))
            if PIPE_INIT_SIDE >= 1:
                hyp3 = ti.init_level(3, L(3), R(3), L(1))
            ev2.record(side)
            feas.append(next(g))
            hyp4 = ti.init_level(4, L(4), R(4), L(2))
        hyp0, hyp1 = ti.init_level(0, L(0), R(0), None), ti.init_level(1, L(1), R(1), None)
        h = tu.tile_update0(L(0), R(0), hyp0)[0]
        h = tu.tile_update1(L(1), R(1), hyp1, h)[0]
        cur.wait_event(ev3)
        h = tu.tile_update2(L(2), R(2), hyp2 if hyp2 is not None else ti.init_level(2, L(2), R(2), L(0)), h)[0]
        cur.wait_event(ev2)
        h = tu.tile_update3(L(3), R(3), hyp3 if hyp3 is not None else ti.init_level(3, L(3), R(3), L(1)), h)[0]
        cur.wait_stream(side)
        h = tu.tile_update4(L(4), R(4), hyp4, h)[0]
        r1 = tu.tile_update4_1(L(2), h)
        r05 = tu.tile_update5(L(3), _up1(r1))
        disp = tu.tile_update6(L(4), _up1(r05))
        return dict(pred_disp=disp, left_feat=L(2), right_feat=R(2), left_img=left_img)

    def stereo_matching(self, left_img, right_img, img_metas=None, state=None):
        """reference hitnet.py:75-100 (eval branch) -> dict(pred_disp, left_feat, right_feat, left_img)."""
        B = left_img.shape[0]
        if (STEREO_PIPE and not ops.Fork.serial and getattr(self, "fork_streams", True) and isinstance(self.backbone, HITUNet) and
                isinstance(self.tile_init, TileInitialization) and isinstance(self.tile_update, TilePropagation)):
            with ops.stage("stereo"):
                return self._stereo_matching_pipelined(left_img, right_img)
        with ops.stage("stereo"):  # exact-fp32 convs: the disparity itself flows through these layers (ops.stage)
            pyr = self.extract_feat(ops.batch_pair(left_img, right_img))
            fea_l = [p[:B] for p in pyr]
            fea_r = [p[B:] for p in pyr]
            self.tile_init.fork_streams = getattr(self, "fork_streams", True)
            _, init = self.tile_init(fea_l, fea_r)
            disp = self.tile_update(fea_l, fea_r, init)
        return dict(pred_disp=disp, left_feat=fea_l[2], right_feat=fea_r[2], left_img=left_img)
