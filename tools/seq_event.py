"""dev: locate the first large divergence between the default (split-bf16) and the all-fp32 product paths in the
16-frame recurrence, and show which intermediate (warped disparity, weights, fused disparity) carries it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth

DEV = "cuda:0"
N = int(os.environ.get("N", "9"))
H, W, intr, img_shape, _, _ = T.CASES["cfg3_codd_960x576"]
img, r_img, _ = synth.stereo_sequence(H, W, N)
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
KEYS = ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp")


NOISE_B = float(os.environ.get("NOISE_B", "0"))  # second run: default arithmetic, inputs perturbed by this relative noise
FIRST = int(os.environ.get("FIRST", "1"))


def run(precision, noise=0.0):
    est = T._build(False, 16)[0].to(DEV)
    g = torch.Generator().manual_seed(1)
    prev = ops.set_conv_precision(precision)
    ops.enable_autotune(True, shipped=True)
    frames = []
    try:
        state = {}
        for f in range(N):
            l, r = img[:, f].clone(), r_img[:, f].clone()
            if noise:
                l = l * (1 + noise * torch.randn(l.shape, generator=g))
                r = r * (1 + noise * torch.randn(r.shape, generator=g))
            out = est.consistent_online_depth_estimation(l.to(DEV).contiguous(), r.to(DEV).contiguous(), metas[0], state)
            frames.append({k: out[k].detach().float().cpu().reshape(H, W) for k in KEYS if k in out})
    finally:
        ops.enable_autotune(False)
        ops.set_conv_precision(prev)
    return frames


a, b = run("split"), (run("split", NOISE_B) if NOISE_B else run("fp32"))
for f in range(FIRST, N):
    line = f"frame {f}:"
    for k in KEYS:
        if k in a[f]:
            d = (a[f][k] - b[f][k]).abs()
            line += f"  {k} mean {d.mean().item():.1e} max {d.max().item():.1e} n>{0.25 if 'pred' in k else 0.05} {(d > (0.25 if 'pred' in k else 0.05)).sum().item()}"
    print(line)
    d = (a[f]["pred_disp"] - b[f]["pred_disp"]).abs()
    ys, xs = torch.nonzero(d > 0.25, as_tuple=True)
    if len(ys):
        print(f"   fused flips: {len(ys)} px, rows {ys.min().item()}..{ys.max().item()} cols {xs.min().item()}..{xs.max().item()}")
        for y, x in list(zip(ys.tolist(), xs.tolist()))[:6]:
            print("   (%d,%d): " % (y, x) + "  ".join(f"{k} {a[f][k][y, x].item():.4f}|{b[f][k][y, x].item():.4f}" for k in KEYS if k in a[f]))
