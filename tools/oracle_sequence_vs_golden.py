"""dev (CPU only): the ORACLE's full recurrence (stereo -> motion -> fusion) on inputs perturbed by 1e-7 relative noise
against the TRACKED oracle frames (tests/golden/headline_oracle_long_sub4.npz, clean inputs) on their sub-grid -- is the
oracle's own trajectory stable where the product's is not (frames 6..13 of cfg3_long)?
Usage: python tools/oracle_sequence_vs_golden.py <long case: cfg3_long | cfg5_long> <frames> [noise] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc

name, MF = sys.argv[1], int(sys.argv[2])
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-7
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
case = {"cfg3_long": "cfg3_codd_960x576", "cfg5_long": "cfg5_tartanair_640x512"}[name]
H, W, intr, _, _, _ = T.CASES[case]
z = np.load(T.LONG_GOLDEN)
sub = int(z["sub"])
sd = T._build(False)[1]
img, r_img, _ = synth.stereo_sequence(H, W, MF)
torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", max(1, min(os.cpu_count() or 1, 16)))))
g = torch.Generator().manual_seed(seed)
st = {}
with torch.no_grad():
    for f in range(MF):
        l, r = img[:, f], r_img[:, f]
        ln = l * (1 + noise * torch.randn(l.shape, generator=g))
        rn = r * (1 + noise * torch.randn(r.shape, generator=g))
        b = oc.frame(sd, ln, rn, st, intr, iters=T.ITERS, with_motion=True, with_fusion=True)["pred_disp"][0, 0, ::sub, ::sub]
        d = (torch.from_numpy(z[f"{name}_f{f}"]) - b).abs()
        print(f"oracle(clean, tracked) vs oracle({noise:g} input noise, seed {seed}), {name} frame {f}: mean |delta| {d.mean().item():.3e}  "
              f"median {d.median().item():.3e}  flipped(>0.25px) {(d > 0.25).float().mean().item():.3e}  max {d.max().item():.3e}", flush=True)
