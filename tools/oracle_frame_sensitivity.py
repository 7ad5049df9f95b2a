"""dev (CPU, no GPU, no reference): how far does the ORACLE's stereo stage move when the input images of one frame
are perturbed by 1e-7 relative noise?  The stereo network is not recurrent, so one frame at a time is the whole story.
Usage: python tools/oracle_frame_sensitivity.py [frame ...]   (default: 13 14 15)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc

frames = [int(a) for a in sys.argv[1:] if a.isdigit()] or [13, 14, 15]
CASE = next((a for a in sys.argv[1:] if a in T.CASES), "cfg3_codd_960x576")
print("case", CASE)
H, W, intr, _, _, _ = T.CASES[CASE]
sd = T._build(True)[1]
img, r_img, _ = synth.stereo_sequence(H, W, max(frames) + 1)
torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
g = torch.Generator().manual_seed(1)
for f in frames:
    with torch.no_grad():
        l, r = img[:, f], r_img[:, f]
        a = oc.frame(sd, l, r, {}, intr, iters=16, with_motion=False, with_fusion=False)["pred_disp"]
        for noise in (1e-7, 1e-6):
            ln = l * (1 + noise * torch.randn(l.shape, generator=g))
            rn = r * (1 + noise * torch.randn(r.shape, generator=g))
            b = oc.frame(sd, ln, rn, {}, intr, iters=16, with_motion=False, with_fusion=False)["pred_disp"]
            d = (a - b).abs()
            print(f"oracle stereo stage, frame {f}, input noise {noise:g}: mean |delta| {d.mean().item():.3e}  "
                  f"flipped(>0.25px) {(d > 0.25).float().mean().item():.3e}  max {d.max().item():.3e}", flush=True)
