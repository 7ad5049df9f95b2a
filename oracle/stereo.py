"""ORACLE (test infrastructure, not product): CPU fp32 restatement of HITNetMF.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``codd_amd``) never does.

Functional style over a flat state dict (reference key names), written from the formulas in
SURVEY.md section 9; every function cites the reference lines it restates.  Convolutions use
``torch.nn.functional.conv2d`` on CPU (the reference's own CPU path); the tile cost volume,
local correlation and plane up-sampling are explicit index arithmetic instead of the
reference's ``grid_sample`` formulation.

Pinned against the imported reference by ``tests/golden/make_golden.py`` (golden vectors in
``tests/golden/*.npz``).
"""
import torch
import torch.nn.functional as F


def lrelu(x):
    return F.leaky_relu(x, 0.2)


def conv(sd, key, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride, pad, dil)


def deconv2(sd, key, x):
    return F.conv_transpose2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=2)


# ----------------------------------------------------------------------------- backbone
def hitunet(sd, p, x):
    """reference model/stereo/hitnet/backbone.py:69-88 (helpers :8-39)."""
    def down(k, t):
        t = lrelu(conv(sd, f"{p}.{k}.0", t, 2, 1))
        return lrelu(conv(sd, f"{p}.{k}.2", t, 1, 1))

    def up(k, t):
        return lrelu(deconv2(sd, f"{p}.{k}.0", t))

    def merge(k, t):
        t = lrelu(conv(sd, f"{p}.{k}.0", t))
        t = lrelu(conv(sd, f"{p}.{k}.2", t, 1, 1))
        return lrelu(conv(sd, f"{p}.{k}.4", t, 1, 1))

    x0 = lrelu(conv(sd, f"{p}.conv1.0", x, 1, 1))
    x1 = down("down1", x0)
    x2 = down("down2", x1)
    x3 = down("down3", x2)
    t = lrelu(conv(sd, f"{p}.down4.0.0", x3, 2, 1))
    t = lrelu(conv(sd, f"{p}.down4.0.2", t, 1, 1))
    t = lrelu(conv(sd, f"{p}.down4.1", t, 1, 1))
    x4 = lrelu(conv(sd, f"{p}.down4.3", t, 1, 1))
    u4 = merge("merge4", torch.cat((x3, up("up4", x4)), 1))
    u3 = merge("merge3", torch.cat((x2, up("up3", u4)), 1))
    u2 = merge("merge2", torch.cat((x1, up("up2", u3)), 1))
    u1 = merge("merge1", torch.cat((x0, up("up1", u2)), 1))
    return [x4, u4, u3, u2, u1]


# ----------------------------------------------------------------------------- tile init
def _l1_seq(a, b):
    """sum_c |a - b| accumulated strictly in channel order (c = 0, 1, ...), so that mathematically
    equal costs are bitwise equal and the systematic ties of the zero-padded region resolve to the
    first index (torch's vectorised .sum(1) changes the association order per output element)."""
    acc = (a[:, 0] - b[:, 0]).abs()
    for c in range(1, a.shape[1]):
        acc = acc + (a[:, c] - b[:, c]).abs()
    return acc


def tile_cost_volume_min(fl, fr, D, chunk=16):
    """Fused restatement of calc_init_disp + torch.min (reference initialization.py:18-45,
    :167-171): cv[d,y,x] = sum_c |L[c,y,x] - R~[c,y,4x-d]|, R~ = 0 outside [0, W_r);
    returns (min cost, FIRST arg-min as float) each [B,1,Ht,Wt]."""
    B, C, Ht, Wt = fl.shape
    Wr = fr.shape[3]
    best = torch.full((B, Ht, Wt), float("inf"))
    arg = torch.zeros((B, Ht, Wt), dtype=torch.long)
    x4 = 4 * torch.arange(Wt)
    for d0 in range(0, D, chunk):
        d = torch.arange(d0, min(D, d0 + chunk))
        idx = x4[None, :] - d[:, None]  # [dc, Wt]
        ok = (idx >= 0) & (idx < Wr)
        g = fr[:, :, :, idx.clamp(0, Wr - 1)]  # [B,C,Ht,dc,Wt]
        g = g * ok[None, None, None].to(g.dtype)
        cv = _l1_seq(fl[:, :, :, None, :], g)  # [B,Ht,dc,Wt]
        c, a = cv.min(2)  # first minimal index inside the chunk
        upd = c < best  # strict: earlier chunk wins ties
        best = torch.where(upd, c, best)
        arg = torch.where(upd, a + d0, arg)
    return best[:, None], arg[:, None].float()


def tile_cost_volume(fl, fr, D):
    """Full (materialised) cost volume [B,D,Ht,Wt]; small sizes only (tests)."""
    B, C, Ht, Wt = fl.shape
    Wr = fr.shape[3]
    idx = 4 * torch.arange(Wt)[None, :] - torch.arange(D)[:, None]
    ok = (idx >= 0) & (idx < Wr)
    g = fr[:, :, :, idx.clamp(0, Wr - 1)] * ok[None, None, None].to(fr.dtype)  # [B,C,Ht,D,Wt]
    return _l1_seq(fl[:, :, :, None, :], g).permute(0, 2, 1, 3).contiguous()


_LEVELS = ["16x", "8x", "4x", "2x", "1x"]


def tile_init(sd, p, fea_l, fea_r, max_disp):
    """reference initialization.py:119-156 (tile_features) + :158-225 (hypothesis pyramid).
    Returns the 5 initial hypotheses [B,16,Ht,Wt], coarse (16x) first."""
    hyps = []
    for lvl, name in enumerate(_LEVELS):
        fl, fr = fea_l[lvl], fea_r[lvl]
        k = f"{p}.tile_conv{name}"
        tl = lrelu(conv(sd, k + ".2", lrelu(conv(sd, k + ".0", fl, 4, 0))))
        frp = F.pad(fr, (0, 3, 0, 0))
        tr = lrelu(conv(sd, k + ".2", lrelu(conv(sd, k + ".0", frp, (4, 1), 0))))
        D = max_disp // (16 >> lvl)
        cost, d = tile_cost_volume_min(tl, tr, D)
        feat = tl if lvl < 2 else fea_l[lvl - 2]
        dsc = lrelu(conv(sd, f"{p}.tile_fea_dscrpt{name}.0", torch.cat([cost, feat], 1)))
        z = torch.zeros_like(d)
        hyps.append(torch.cat([d, z, z, dsc], 1))
    return hyps


# ----------------------------------------------------------------------------- propagation
def to_plane(d, dx, dy, s):
    """reference propagation.py:10-23: [B,1,h,w] -> [B,1,s*h,s*w]."""
    c = torch.linspace(-(s - 1) / 2, (s - 1) / 2, s)
    up = lambda t: t.repeat_interleave(s, 2).repeat_interleave(s, 3)
    h, w = d.shape[2:]
    a = c.repeat(w)[None, None, None, :]
    b = c.repeat(h)[None, None, :, None]
    return up(d) + a * up(dx) + b * up(dy)


def upsample_hyp(h, scale=2, size=2):
    """reference propagation.py:26-32."""
    d = to_plane(h[:, 0:1], h[:, 1:2], h[:, 2:3], size) * scale
    rest = h[:, 1:].repeat_interleave(size, 2).repeat_interleave(size, 3)
    return torch.cat((d, rest), 1)


def warp_x(fr, disp):
    """Bilinear sample of fr at (x - disp, y); zero padding, pixel units (align_corners=True):
    reference propagation.py:35-58 / utils/warp.py:43-66."""
    B, C, H, W = fr.shape
    xs = torch.arange(W, dtype=torch.float32)[None, None, None, :] - disp
    x0 = torch.floor(xs)
    a = xs - x0
    x0 = x0.long()
    x1 = x0 + 1

    def take(ix):
        ok = ((ix >= 0) & (ix < W)).to(fr.dtype)
        return torch.gather(fr, 3, ix.clamp(0, W - 1).expand(B, C, H, W)) * ok

    return (1 - a) * take(x0) + a * take(x1)


def unshuffle4(x):
    return F.pixel_unshuffle(x, 4)


def tile_warping(plane, fl, fr):
    """reference propagation.py:61-86: -> [B,48,Ht,Wt], channel = (k+1)*16 + iy*4 + ix."""
    out = []
    for k in (-1, 0, 1):
        d = to_plane(plane[:, 0:1] + k, plane[:, 1:2], plane[:, 2:3], 4)
        cv = (fl - warp_x(fr, d)).abs().sum(1, keepdim=True)
        out.append(unshuffle4(cv))
    return torch.cat(out, 1)


def _resblock(sd, k, x, dil=1):
    """reference propagation.py:103-121 BasicBlock (no BN) followed by LeakyReLU."""
    t = lrelu(conv(sd, k + ".0.conv1.0.0", x, 1, dil, dil))
    t = conv(sd, k + ".0.conv2.0", t, 1, dil, dil)
    return lrelu(t + x)


def _relu_d(h):
    return torch.cat([F.relu(h[:, :1]), h[:, 1:]], 1)


def tile_update0(sd, p, fl, fr, hyp):
    """reference propagation.py:124-172."""
    fea = unshuffle4(fl.abs().sum(1, keepdim=True))
    cvv = lrelu(conv(sd, p + ".decrease.0", torch.cat([fea, tile_warping(hyp[:, :3], fl, fr)], 1)))
    t = lrelu(conv(sd, p + ".conv0.0", torch.cat([hyp, cvv], 1)))
    t = _resblock(sd, p + ".resblock0", t)
    t = _resblock(sd, p + ".resblock1", t)
    t = conv(sd, p + ".lastconv", t, 1, 1)
    return _relu_d(hyp + t)


def tile_update(sd, p, fl, fr, hyp, prev):
    """reference propagation.py:175-248; returns refined hypothesis only (inference)."""
    fea = unshuffle4(fl.abs().sum(1, keepdim=True))
    cv_c = lrelu(conv(sd, p + ".decrease.0", torch.cat([fea, tile_warping(hyp[:, :3], fl, fr)], 1)))
    up = upsample_hyp(prev, 2, 2)
    cv_p = lrelu(conv(sd, p + ".decrease.0", torch.cat([fea, tile_warping(up[:, :3], fl, fr)], 1)))
    t = lrelu(conv(sd, p + ".conv0.0", torch.cat([hyp, cv_c, up, cv_p], 1)))
    t = _resblock(sd, p + ".resblock0", t)
    t = _resblock(sd, p + ".resblock1", t)
    t = conv(sd, p + ".lastconv", t, 1, 1)
    conf = t[:, :2]
    sel = conf.argmax(1, keepdim=True).float()  # ties -> 0 (= previous)
    cur = _relu_d(hyp + t[:, 18:34])
    prv = _relu_d(up + t[:, 2:18])
    return sel * cur + (1 - sel) * prv


def post_tile_update(sd, p, fl, prev, nblk, final=False):
    """reference propagation.py:251-290 (PostTileUpdate) and :293-333 (FinalTileUpdate)."""
    t = lrelu(conv(sd, p + ".conv1.0", torch.cat([fl, prev], 1)))
    t = lrelu(conv(sd, p + ".conv1.2", t, 1, 1))
    for i in range(nblk):
        dil = 3 if (i == 1 and not final) else 1
        t = _resblock(sd, f"{p}.resblocks.{i}", t, dil)
    t = conv(sd, p + ".lastconv", t, 1, 1)
    if final:
        return F.relu(prev[:, 0:1] + t)
    return _relu_d(prev + t)


def tile_propagation(sd, p, fea_l, fea_r, init):
    """reference propagation.py:359-372, 453-454 (inference branch)."""
    h16 = tile_update0(sd, p + ".tile_update0", fea_l[0], fea_r[0], init[0])
    h8 = tile_update(sd, p + ".tile_update1", fea_l[1], fea_r[1], init[1], h16)
    h4 = tile_update(sd, p + ".tile_update2", fea_l[2], fea_r[2], init[2], h8)
    h2 = tile_update(sd, p + ".tile_update3", fea_l[3], fea_r[3], init[3], h4)
    h1 = tile_update(sd, p + ".tile_update4", fea_l[4], fea_r[4], init[4], h2)
    r1 = post_tile_update(sd, p + ".tile_update4_1", fea_l[2], h1, 4)
    r05 = post_tile_update(sd, p + ".tile_update5", fea_l[3], upsample_hyp(r1, 1, 2), 4)
    r025 = post_tile_update(sd, p + ".tile_update6", fea_l[4], upsample_hyp(r05, 1, 2), 2, final=True)
    return r025[:, 0:1]


def stereo_matching(sd, left, right, max_disp=320, p="stereo", return_intermediates=False):
    """reference model/stereo/hitnet/hitnet.py:75-100 (eval branch)."""
    fea_l = hitunet(sd, p + ".backbone", left)
    fea_r = hitunet(sd, p + ".backbone", right)
    init = tile_init(sd, p + ".tile_init", fea_l, fea_r, max_disp)
    disp = tile_propagation(sd, p + ".tile_update", fea_l, fea_r, init)
    out = dict(pred_disp=disp, left_feat=fea_l[2], right_feat=fea_r[2], left_img=left)
    if return_intermediates:
        out.update(fea_l=fea_l, fea_r=fea_r, init=init)
    return out
