"""Deterministic synthetic weights and inputs.

There is no network on the build or GPU boxes, so no published checkpoint can be loaded.
Weights come from a seedable filler keyed by the state-dict name (numpy's legacy
``RandomState`` is bit-reproducible across machines), so the same weights can be recreated
in the build container (golden generation against the imported reference), in the CPU
oracle and on the GPU box without committing multi-MB blobs.

Inputs are closed-form band-limited textures: left image = sum of
separable sinusoids, right image = left resampled by a smooth disparity field, frame t =
frame 0 translated by a smooth sub-pixel flow.  Shapes follow the reference's tensor
contract ``img, r_img : float32 [B, MF, 3, H, W]`` (reference datasets/formating.py:77-85).
"""
import math
import zlib

import numpy as np
import torch


def _rs(name):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)


# Layers that emit RESIDUALS of the recurrent estimates (HITNet's hypothesis deltas, RAFT3D's target
# residual): trained networks keep them small.  With unit-scale random weights the deltas are +-100 px, the
# ReLU after every "d + delta" (reference propagation.py:171,232-237) zeroes ~99.7 % of the disparities
# and a parity test would compare zeros with zeros; scaled down, the disparity stays around the cost-volume
# initialisation and every stage sees live data.
_RESIDUAL_LAYERS = (".lastconv.", ".update_block.delta.2.")
_RESIDUAL_SCALE = 0.02


def fill_tensor(name, shape, gain=1.0):
    """Deterministic value for state-dict entry ``name`` of ``shape`` (fp32 torch tensor)."""
    rs = _rs(name)
    if any(t in name for t in _RESIDUAL_LAYERS):
        gain = gain * _RESIDUAL_SCALE
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        v = 1.0 + 0.1 * np.abs(rs.standard_normal(shape))
    elif leaf == "running_mean":
        v = 0.05 * rs.standard_normal(shape)
    elif len(shape) >= 2:  # conv / deconv weight
        fan_in = int(np.prod(shape[1:]))
        v = gain * rs.standard_normal(shape) / math.sqrt(fan_in)
    elif leaf == "weight":  # norm scale
        v = 1.0 + 0.1 * rs.standard_normal(shape)
    else:  # bias
        v = 0.05 * rs.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


# --------------------------------------------------------------------------- conditioned weights (round 6)
# The random filler above gives a network that is NOT a stereo matcher: its strided random convolutions alias, its
# hypothesis selection (reference propagation.py:225-240) is a coin toss between the levels and its residual heads add
# ~1 px of noise per stage -- disparity error 23-36 px on the synthetic video, coarse-tile arg-mins on near-ties in
# 7-10 % of the frames, Fusion's sigmoid heads saturated at 0 / 1.  Two correct fp32 evaluations of such a network
# disagree by whole tiles.  ``mode="conditioned"`` is a closed-form weight set, still keyed by
# state-dict name and made of nothing but integer patterns and the reproducible RandomState draws (no BLAS, no QR: the
# same bits on every machine), under which HITNet IS a coarse-to-fine block matcher:
#   * backbone: anti-aliased decimation ([1 3 3 1] x [1 3 3 1] / 64 per channel on the 4x4 stride-2 convolutions),
#     nearest up-sampling on the 2x2 deconvolutions, merges that pass 0.7 x skip + 0.3 x up-path, identity 3x3
#     layers -- each plus a 5 % random perturbation on every tap, so every tap of every kernel still matters;
#   * hypothesis selection driven by the local matching cost: `decrease` channel 0 = mean warping cost of the
#     hypothesis (k = 0), `conv0` channels 0 / 1 = +-(cost_cur - cost_prev), `lastconv` confidence rows read them
#     through the (down-weighted) residual blocks: the cheaper hypothesis wins, as in a trained network -- with a
#     coarse-to-fine consistency prior: `conv0` channels 2 / 3 = lrelu(+-(d_cur - d_prev)), whose sum 0.8 |d_cur - d_prev|
#     enters the previous level's confidence with weight lam = 0.8.  Without it 7-9 tiles of a frame take a chance match far
#     away in the 320-candidate range (a 4x4 patch that happens to fit better than the true match does at its sub-pixel
#     offset): disparities up to 300 px = points at 0.7 m whose splats cover their neighbours -- the "near-camera clusters"
#     that made frame 44 of the first conditioned golden miss the bound; with it the maximum disparity of a frame is 49.6 px
#     (ground truth <= 48);
#   * residual heads (`lastconv`) at 0.005 instead of 0.02: sub-pixel perturbations of the block matcher's result;
#   * Fusion's two sigmoid heads (`weight_head.1`, `forget_head.2`) scaled so that their logits stay inside |x| < 2.
# Measured on the oracle (tools/cond_probe.py): tile initialisation within one disparity step of the ground truth on
# 99.1-99.8 % of the tiles at every level, final error median 0.38 px / mean 0.46 px, at most 0.02 % of the pixels off by more
# than 3 px, no disparity above 50 px; under 1e-5 relative input noise at most two 4x4 tiles of a frame move.
_COND = dict(eps=0.05, skip=0.7, up=0.3, res=0.005, g_res=0.3, gamma=1.0, lam=0.8, weight_head=0.01, weight_bias=1.5, forget_head=0.2)
_LP4 = np.outer([1.0, 3.0, 3.0, 1.0], [1.0, 3.0, 3.0, 1.0]) / 64.0


def _rnd(name, shape, gain):
    return gain * _rs(name).standard_normal(shape) / math.sqrt(int(np.prod(shape[1:])))


def _conditioned(name, shape, gain):
    """float64 array for state-dict entry ``name`` under mode="conditioned", or None: use the random filler."""
    C = _COND
    leaf = name.rsplit(".", 1)[-1]
    if name == "fusion.weight_head.1.weight":
        return _rnd(name, shape, gain * C["weight_head"])
    if name == "fusion.weight_head.1.bias":
        return np.full(shape, C["weight_bias"])
    if name == "fusion.forget_head.2.weight":
        return _rnd(name, shape, gain * C["forget_head"])
    if not name.startswith("stereo."):
        return None
    n = name[len("stereo."):]
    if leaf == "bias":
        if n.startswith("backbone.") or ".decrease." in n or ".conv0." in n:
            return np.zeros(shape)
        if ".lastconv." in n and shape == (34,):
            v = 0.05 * _rs(name).standard_normal(shape)
            v[:2] = 0.0
            return v
        return None
    if leaf != "weight":
        return None
    if n.startswith("backbone."):
        k = n[len("backbone."):-len(".weight")]
        if k == "conv1.0":
            return _rnd(name, shape, 1.0)
        w = _rnd(name, shape, C["eps"])
        if k.startswith("up"):  # ConvTranspose2d weight [Cin, Cout, 2, 2]: nearest up-sampling, channel i -> i
            for j in range(shape[1]):
                w[j % shape[0], j] += 1.0
            return w
        co, ci, kh, _ = shape
        for j in range(co):
            if kh == 4 and j < ci:  # anti-aliased decimation
                w[j, j] += _LP4
            elif kh == 4:  # channel growth: low-passed difference of two input channels
                w[j, (j * 5 + 1) % ci] += 0.75 * _LP4
                w[j, (j * 7 + 3) % ci] -= 0.75 * _LP4
            elif kh == 1:  # merge.0 over cat(skip, up)
                w[j, j, 0, 0] += C["skip"]
                w[j, ci // 2 + j, 0, 0] += C["up"]
            else:
                w[j, j, 1, 1] += 1.0
        return w
    if ".tile_conv" in n and n.endswith(".2.weight"):
        w = _rnd(name, shape, C["eps"])
        for j in range(shape[0]):
            w[j, j, 0, 0] += 1.0
        return w
    if n.endswith(".decrease.0.weight"):  # in = [|fea| 16, cost(k=-1) 16, cost(k=0) 16, cost(k=+1) 16]
        w = _rnd(name, shape, 1.0)
        for o, base in ((0, 32), (1, 16), (2, 48)):
            w[o] = 0.0
            w[o, base:base + 16, 0, 0] = 1.0 / 16
        return w
    if n.endswith(".conv0.0.weight") and shape[1] == 64:  # in = [hyp 16, cv_cur 16, up_prev 16, cv_prev 16]
        w = _rnd(name, shape, 1.0)
        w[0:2] = 0.0
        w[0, 16, 0, 0], w[0, 48, 0, 0] = C["gamma"], -C["gamma"]
        w[1, 16, 0, 0], w[1, 48, 0, 0] = -C["gamma"], C["gamma"]
        if C.get("lam", 0.0) > 0:  # channels 2 / 3 = lrelu(+-(d_cur - d_prev)): their sum is 0.8 |d_cur - d_prev|
            w[2:4] = 0.0
            w[2, 0, 0, 0], w[2, 32, 0, 0] = 1.0, -1.0
            w[3, 0, 0, 0], w[3, 32, 0, 0] = -1.0, 1.0
        return w
    if ".resblock" in n and any(f".tile_update{i}." in n for i in range(1, 5)):
        return _rnd(name, shape, C["g_res"])
    if n.endswith(".lastconv.weight"):
        w = _rnd(name, shape, gain * C["res"])
        if shape[0] == 34:  # rows 0 / 1 = confidence of (up-sampled previous, current): read conv0's channels 0 / 1
            w[0:2] = 0.0
            w[0, 0, 1, 1] = w[1, 1, 1, 1] = 1.0
            if C.get("lam", 0.0) > 0:  # + lam * 0.8 |d_cur - d_prev| in favour of the coarser level's hypothesis
                w[0, 2, 1, 1] = w[0, 3, 1, 1] = C["lam"]
        return w
    return None


def fill_state_dict(keys_and_shapes, gain=1.0, mode="random"):
    """``keys_and_shapes``: iterable of (name, shape) or a module / state_dict.  ``mode``: "random" (every golden up to
    round 5) or "conditioned" (the block-matcher weight set above; entries it does not design come from the random filler)."""
    assert mode in ("random", "conditioned"), mode
    if hasattr(keys_and_shapes, "state_dict"):
        keys_and_shapes = keys_and_shapes.state_dict()
    if isinstance(keys_and_shapes, dict):
        keys_and_shapes = [(k, tuple(v.shape)) for k, v in keys_and_shapes.items()]
    out = {}
    for k, s in keys_and_shapes:
        v = _conditioned(k, tuple(int(x) for x in s), gain) if mode == "conditioned" else None
        out[k] = fill_tensor(k, s, gain) if v is None else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def load_synthetic_weights(module, gain=1.0, mode="random"):
    """Fill every parameter / buffer of ``module`` in place from the deterministic filler."""
    sd = module.state_dict()
    new = fill_state_dict(sd, gain, mode)
    module.load_state_dict({k: new[k].to(sd[k].dtype) for k in sd}, strict=True)
    return module


# --------------------------------------------------------------------------- inputs
_TEX = [  # (amp, wx, px, wy, py) fixed table -> band-limited texture
    (0.80, 0.051, 0.3, 0.043, 1.1), (0.55, 0.113, 1.7, 0.097, 0.2), (0.45, 0.237, 2.9, 0.181, 2.3),
    (0.35, 0.389, 0.9, 0.411, 1.9), (0.30, 0.671, 2.1, 0.557, 0.7), (0.20, 1.013, 1.3, 0.893, 2.7),
]


def _texture(x, y, ch):
    out = torch.zeros_like(x)
    for k, (a, wx, px, wy, py) in enumerate(_TEX):
        out = out + a * torch.sin(wx * x + px + 0.9 * ch) * torch.cos(wy * y + py + 0.4 * ch * (k + 1))
    return out


def _plane_wave_table(n=48, seed=12345):
    """n plane waves (amp, kx, ky, phase, channel phase step) from a fixed LCG: wave vectors of random direction with
    |k| log-uniform in [0.03, 1.3] rad/px, amplitude ~ |k|^-0.5.  Unlike _TEX (six x/y-separable sinusoids, nearly
    periodic along x at the coarse tile scales) the sum has no short self-similarity along the disparity axis, so the
    tile cost volume's minima are well separated (tools/video_margin_scan.py)."""
    st, rows = seed, []

    def u():
        nonlocal st
        st = (1103515245 * st + 12345) % (1 << 31)
        return st / float(1 << 31)
    for _ in range(n):
        k = 0.03 * math.exp(u() * math.log(1.3 / 0.03))
        th = 2 * math.pi * u()
        rows.append((0.5 * (0.03 / k) ** 0.5 * 3.0, k * math.cos(th), k * math.sin(th), 2 * math.pi * u(), 0.3 + 1.7 * u()))
    return rows


_WAVES = _plane_wave_table()


def _texture_waves(x, y, ch):
    out = torch.zeros_like(x)
    for a, kx, ky, ph, cs in _WAVES:
        out = out + a * torch.sin(kx * x + ky * y + ph + cs * ch)
    return out


def disparity_field(H, W, t=0.0, dmin=1.0, dmax=48.0, left_taper=0.0):
    """Smooth positive disparity field [H, W] (px) at time ``t``.  ``left_taper`` > 0: the field falls linearly to 0
    over the first ``left_taper`` columns, so that every left pixel has its match INSIDE the right image (x - d >= 0)
    and no unmatched band with arbitrary arg-mins exists at the left border."""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    s = 0.5 + 0.25 * torch.sin(2 * math.pi * (x / W) + 0.1 * t) + 0.25 * torch.cos(2 * math.pi * (y / H) * 1.5)
    d = dmin + (dmax - dmin) * s.clamp(0, 1)
    return d * (x / left_taper).clamp(0, 1) if left_taper > 0 else d


def stereo_sequence(H, W, MF, dmax=48.0, flow=(0.75, 0.25), texture="sines", dphase=1.0, left_taper=0.0):
    """Synthetic stereo video: returns (img, r_img) float32 [1, MF, 3, H, W] and gt disparity
    [1, MF, 1, H, W].  Left frame t samples the texture at (x - t*fx, y - t*fy); the right
    image samples the same texture at (x + d(x, y)) so that right(x - d) ~ left(x).
    ``texture``: "sines" (the six separable sinusoids every golden up to round 4 was made with) or "waves" (48 plane
    waves of random direction: no near-periodicity along the disparity axis)."""
    assert texture in ("sines", "waves"), texture
    _texture = {"sines": globals()["_texture"], "waves": _texture_waves}[texture]
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    ls, rs, ds = [], [], []
    for t in range(MF):
        d = disparity_field(H, W, float(t) * dphase, dmax=dmax, left_taper=left_taper)  # (dphase: speed of the disparity field's drift, 0.1 rad/frame * dphase)
        xs, ys = x - t * flow[0], y - t * flow[1]
        left = torch.stack([_texture(xs, ys, c) for c in range(3)])
        right = torch.stack([_texture(xs + d, ys, c) for c in range(3)])
        ls.append(left)
        rs.append(right)
        ds.append(d[None])
    img = torch.stack(ls)[None].contiguous()
    r_img = torch.stack(rs)[None].contiguous()
    gt = torch.stack(ds)[None].contiguous()
    return img, r_img, gt


def default_metas(H, W, img_shape=None, intrinsics=(1050.0, 1050.0, 480.0, 270.0)):
    """``img_metas`` as produced by the reference pipeline (configs/datasets/custom.py:24-35,
    configs/datasets/scene_flow.py:13-15)."""
    if img_shape is None:
        img_shape = (H, W, 3)
    return [[dict(img_shape=tuple(img_shape), ori_shape=tuple(img_shape), pad_shape=(H, W, 3),
                  disp_range=(1, 210), depth_range=(1.0, 210.0), calib=210.0,
                  intrinsics=list(intrinsics), filename="synthetic")]]
