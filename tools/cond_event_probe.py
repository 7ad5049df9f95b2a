"""dev (GPU): where does the product leave the tracked oracle on a frame of a conditioned case, and what do its own stereo / warp /
fusion-weight maps look like there?  Runs the product eagerly (no graph), and for frames FROM..TO lists the sampled pixels that
differ from the golden by more than 0.25 px with pred_curr, pred_warp, the two weights and the fused value at that pixel.
    python tools/cond_event_probe.py cfg3_50c 42 46
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth

name, f0, f1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
base, iters, MF, sub = T.COND_CASES[name]
H, W, intr, img_shape, _, _ = T.CASES[base]
z = np.load(T.COND_GOLDEN)
dev = "cuda:0"
est = T._build(False, iters, mode="conditioned")[0].to(dev)
img, r_img, _ = T.conditioned_video(H, W, f1 + 1)
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
ops.enable_autotune(True, shipped=True)
state = {}
with torch.no_grad():
    for f in range(f1 + 1):
        out = est.consistent_online_depth_estimation(img[:, f].to(dev).contiguous(), r_img[:, f].to(dev).contiguous(), metas[0], state)
        if f < f0:
            continue
        d = out["pred_disp"][0, 0].cpu()
        g = torch.from_numpy(z[f"{name}_f{f}"])
        diff = (d[::sub, ::sub] - g).abs()
        ys, xs = torch.nonzero(diff > 0.25, as_tuple=True)
        print(f"frame {f}: sub-grid mean |delta| {diff.mean():.3e}, {len(ys)} sampled pixels off by > 0.25 px; product max disparity {d.max():.1f}, "
              f"pixels with disparity > 80: {(d > 80).sum().item()}; pred_warp > 80: {(out['pred_warp'] > 80).sum().item() if 'pred_warp' in out else -1}")
        for y, x in zip(ys.tolist(), xs.tolist()):
            Y, X = y * sub, x * sub
            pc, pw = out["pred_curr"][0, 0, Y, X].item(), out["pred_warp"][0, 0, Y, X].item()
            wf, wr = out["fusion_weights"][0, 0, Y, X].item(), out["reset_weights"][0, 0, Y, X].item()
            print(f"   pixel ({Y}, {X}): golden {g[y, x]:.3f} product {d[Y, X]:.3f} | pred_curr {pc:.3f} pred_warp {pw:.3f} wf {wf:.3f} wr {wr:.3f}")
            pcw = out["pred_curr"][0, 0, max(0, Y - 4):Y + 5, max(0, X - 4):X + 5]
            pww = out["pred_warp"][0, 0, max(0, Y - 4):Y + 5, max(0, X - 4):X + 5]
            print(f"      9x9 window: pred_curr min {pcw.min():.1f} max {pcw.max():.1f}; pred_warp min {pww.min():.1f} max {pww.max():.1f} holes {(pww <= 0).sum().item()}")
