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


def fill_state_dict(keys_and_shapes, gain=1.0):
    """``keys_and_shapes``: iterable of (name, shape) or a module / state_dict."""
    if hasattr(keys_and_shapes, "state_dict"):
        keys_and_shapes = keys_and_shapes.state_dict()
    if isinstance(keys_and_shapes, dict):
        keys_and_shapes = [(k, tuple(v.shape)) for k, v in keys_and_shapes.items()]
    return {k: fill_tensor(k, s, gain) for k, s in keys_and_shapes}


def load_synthetic_weights(module, gain=1.0):
    """Fill every parameter / buffer of ``module`` in place from the deterministic filler."""
    sd = module.state_dict()
    new = fill_state_dict(sd, gain)
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


def disparity_field(H, W, t=0.0, dmin=1.0, dmax=48.0):
    """Smooth positive disparity field [H, W] (px) at time ``t``."""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    s = 0.5 + 0.25 * torch.sin(2 * math.pi * (x / W) + 0.1 * t) + 0.25 * torch.cos(2 * math.pi * (y / H) * 1.5)
    return dmin + (dmax - dmin) * s.clamp(0, 1)


def stereo_sequence(H, W, MF, dmax=48.0, flow=(0.75, 0.25), texture="sines", dphase=1.0):
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
        d = disparity_field(H, W, float(t) * dphase, dmax=dmax)  # (dphase: speed of the disparity field's drift, 0.1 rad/frame * dphase)
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
