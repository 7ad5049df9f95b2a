"""dev (CPU only): is a frame of a recurrent parity case ILL-CONDITIONED, i.e. does the ORACLE itself leave its tracked
trajectory under a legitimate re-evaluation of the same fp32 arithmetic?  

Runs the oracle's full recurrence (oracle/codd.py: stereo -> motion -> fusion) once per VARIANT on the clean inputs and
compares every frame with the TRACKED oracle frames (tests/golden/headline_oracle_long_sub4.npz) on their sub-grid, and
at full resolution with the "default" variant when that is among the variants.

  default    the oracle as the golden generators run it (oneDNN convolutions): must reproduce the golden bit for bit
  nomkldnn   torch.backends.mkldnn.flags(enabled=False): every convolution through ATen's im2col + sgemm path --
             another summation order of the same fp32 sums
  conv64     every convolution evaluated in fp64 and rounded to fp32 once (error <= 0.5 ulp per output: the
             correctly-rounded convolution any fp32 implementation approximates)
  noise      1e-7 relative noise on the input images (the round-4 probe; below 1 ulp for most pixels)

Usage: python tools/oracle_reassoc_probe.py <long case> <frames> <variant> [<variant> ...]
       env CODD_GOLDEN_THREADS, CODD_PROBE_SAVE=dir (full-resolution frames of every variant as <case>_<variant>.pt)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc


def run(name, MF, variant, save_dir=None):
    base, iters, _ = T.LONG_CASES[name][:3]
    flow = T.LONG_CASES[name][3] if len(T.LONG_CASES[name]) > 3 else None
    H, W, intr, _, _, _ = T.CASES[base]
    z = np.load(T.LONG_GOLDEN)
    sub = int(z["sub"])
    sd = T._build(False, iters)[1]
    img, r_img, _ = synth.stereo_sequence(H, W, MF) if flow is None else synth.stereo_sequence(H, W, MF, flow=flow)
    torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", max(1, min(os.cpu_count() or 1, 16)))))
    g = torch.Generator().manual_seed(1)
    orig = (F.conv2d, F.conv_transpose2d)
    ctx = None
    if variant == "nomkldnn":
        ctx = torch.backends.mkldnn.flags(enabled=False)
        ctx.__enter__()
    elif variant == "conv64":
        def up(a):
            return a.double() if torch.is_tensor(a) and a.is_floating_point() else a
        F.conv2d = lambda *a, **k: orig[0](*[up(x) for x in a], **{q: up(v) for q, v in k.items()}).float()
        F.conv_transpose2d = lambda *a, **k: orig[1](*[up(x) for x in a], **{q: up(v) for q, v in k.items()}).float()
    st, frames = {}, []
    try:
        with torch.no_grad():
            for f in range(MF):
                t0 = time.time()
                l, r = img[:, f], r_img[:, f]
                if variant == "noise":
                    l = l * (1 + 1e-7 * torch.randn(l.shape, generator=g))
                    r = r * (1 + 1e-7 * torch.randn(r.shape, generator=g))
                full = oc.frame(sd, l, r, st, intr, iters=iters, with_motion=True, with_fusion=True)["pred_disp"].clone()
                frames.append(full)
                key = f"{name}_f{f}"
                if key in z.files:
                    d = (torch.from_numpy(z[key]) - full[0, 0, ::sub, ::sub]).abs()
                    print(f"oracle[{variant}] vs tracked oracle, {name} frame {f:2d} sub-grid: mean |delta| {d.mean().item():.3e}  "
                          f"median {d.median().item():.3e}  flipped(>0.25px) {(d > 0.25).float().mean().item():.3e} "
                          f"({int((d > 0.25).sum())} px)  max {d.max().item():.3e}   [{time.time() - t0:.0f} s]", flush=True)
                else:
                    print(f"oracle[{variant}] {name} frame {f:2d}: mean disparity {full.mean().item():.4f}  [{time.time() - t0:.0f} s]", flush=True)
                if save_dir:
                    os.makedirs(save_dir, exist_ok=True)
                    torch.save(frames, os.path.join(save_dir, f"{name}_{variant}.pt"))
    finally:
        F.conv2d, F.conv_transpose2d = orig
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return frames


def main():
    name, MF, variants = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    save = os.environ.get("CODD_PROBE_SAVE")
    out = {v: run(name, MF, v, save) for v in variants}
    vs = list(out)
    for i in range(len(vs)):
        for j in range(i + 1, len(vs)):
            for f in range(MF):
                d = (out[vs[i]][f] - out[vs[j]][f]).abs()
                print(f"oracle[{vs[i]}] vs oracle[{vs[j]}], {name} frame {f:2d} FULL resolution: mean |delta| {d.mean().item():.3e}  "
                      f"flipped(>0.25px) {(d > 0.25).float().mean().item():.3e} ({int((d > 0.25).sum())} px)  max {d.max().item():.3e}")


if __name__ == "__main__":
    main()
