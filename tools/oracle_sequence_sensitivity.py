"""dev (CPU only): the ORACLE's full recurrence (stereo -> motion -> fusion) on clean inputs against the same recurrence on
inputs perturbed by 1e-7 relative noise -- per-frame mean |delta| / flipped fraction of the fused disparity.
Usage: python tools/oracle_sequence_sensitivity.py <case> <frames> [noise]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc

case, MF = sys.argv[1], int(sys.argv[2])
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-7
H, W, intr, _, _, _ = T.CASES[case]
sd = T._build(False)[1]
img, r_img, _ = synth.stereo_sequence(H, W, MF)
torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
g = torch.Generator().manual_seed(1)
sa, sb = {}, {}
with torch.no_grad():
    for f in range(MF):
        l, r = img[:, f], r_img[:, f]
        a = oc.frame(sd, l, r, sa, intr, iters=T.ITERS, with_motion=True, with_fusion=True)["pred_disp"]
        ln = l * (1 + noise * torch.randn(l.shape, generator=g))
        rn = r * (1 + noise * torch.randn(r.shape, generator=g))
        b = oc.frame(sd, ln, rn, sb, intr, iters=T.ITERS, with_motion=True, with_fusion=True)["pred_disp"]
        d = (a - b).abs()
        print(f"oracle vs oracle({noise:g} input noise), {case} frame {f}: mean |delta| {d.mean().item():.3e}  "
              f"flipped(>0.25px) {(d > 0.25).float().mean().item():.3e}  max {d.max().item():.3e}", flush=True)
