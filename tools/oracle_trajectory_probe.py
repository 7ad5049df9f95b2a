"""dev (CPU, meant for the GPU box's host cores): how stable is the ORACLE's own trajectory over several frames?

The one-frame probes (tools/oracle_frame_probe.py) show the oracle's frame function well conditioned on most frames -- and
the product still drifts away from it inside a small image region over 5-6 frames (round 5: rows 4..47 / cols 430..535 of
the 50-frame video from frame ~15 on, 15 px on 19 pixels at frame 19).  That is the signature of a locally UNSTABLE
recurrence (Fusion's weight heads read the warped memory their own output fed; untrained weights give the loop a gain > 1
where |pred_warp - pred_curr| is ~150 px), which only a multi-frame probe can show: restart the oracle from its saved state
behind frame F0 - 1, perturb that state ONCE by a relative noise of the size of the product's documented arithmetic
(split-bf16: 2^-17 ~ 1e-5) and follow frames F0 .. F1 against the tracked trajectory.

    STATE=path/to/state_behind_F0-1.pt GOLDEN=path.npz F0=15 F1=24 NOISE=1e-5 SEEDS=2 python tools/oracle_trajectory_probe.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc

LONG = os.environ.get("LONG", "cfg3_50")
F0, F1 = int(os.environ.get("F0", "15")), int(os.environ.get("F1", "24"))
NOISE, SEEDS = float(os.environ.get("NOISE", "1e-5")), int(os.environ.get("SEEDS", "2"))
case = T.LONG_CASES[LONG]
H, W, intr, _, _, _ = T.CASES[case[0]]
golden = np.load(os.environ["GOLDEN"])
sub = int(golden["sub"])
torch.set_num_threads(int(os.environ.get("THREADS", max(1, min(os.cpu_count() or 1, 32)))))
sd = T._build(False, case[1])[1]
img, r_img, _ = synth.stereo_sequence(H, W, F1 + 1, **({"flow": case[3]} if len(case) > 3 else {}))
ck = torch.load(os.environ["STATE"], map_location="cpu")
assert ck["f"] == F0 - 1, (ck["f"], F0)


def noisy(v, gen):
    if torch.is_tensor(v):
        return v * (1 + NOISE * torch.randn(v.shape, generator=gen)) if v.is_floating_point() else v
    if isinstance(v, (list, tuple)):
        return type(v)(noisy(x, gen) for x in v)
    if isinstance(v, dict):
        return {k: noisy(x, gen) for k, x in v.items()}
    return v


for seed in range(-1 if os.environ.get("CONTROL", "1") == "1" else 0, SEEDS):  # seed -1: unperturbed control (must reproduce the golden)
    st = torch.load(os.environ["STATE"], map_location="cpu")["state"]
    if seed >= 0:
        st = noisy(st, torch.Generator().manual_seed(77 + seed))
    for f in range(F0, F1 + 1):
        t0 = time.time()
        with torch.no_grad():
            d = oc.frame(sd, img[:, f], r_img[:, f], st, intr, iters=case[1])["pred_disp"][0, 0]
        if f"{LONG}_f{f}" not in golden.files:
            break
        e = (d[::sub, ::sub] - torch.from_numpy(golden[f"{LONG}_f{f}"])).abs()
        ys, xs = torch.nonzero(e > 0.25, as_tuple=True)
        box = f"rows {int(ys.min()) * sub}..{int(ys.max()) * sub} cols {int(xs.min()) * sub}..{int(xs.max()) * sub}" if len(ys) else "-"
        print(f"{LONG} {'control (no noise)' if seed < 0 else f'state noise {NOISE:g} at frame {F0 - 1}, seed {seed}'}: frame {f} oracle vs its tracked trajectory "
              f"(sub-grid): mean {e.mean():.2e}  flipped {(e > 0.25).float().mean():.2e} ({len(ys)} px, {box})  max {e.max():.2e}  [{time.time() - t0:.0f} s]", flush=True)
        if seed < 0 and f == F0 + 1:
            break  # two control frames are enough
