"""dev (CPU): conditioning of ONE frame of the recurrence, measured on the oracle itself.

Given the oracle's recurrent state behind frame F - 1 of a LONG case (saved per frame by tests/golden/make_long_golden.py,
"<out>.<case>.state.pt", kept per frame by the snapshot loop of round 5), evaluate frame F
  * on the exact images (must reproduce the tracked golden frame), and
  * K times on images carrying NOISE relative noise (default 1e-6: ~10x the differences between two fp32 evaluation orders),
    optionally also with every convolution on ATen's im2col + sgemm path (VARIANT=nomkldnn),
and report how far the frame's OUTPUT (fused disparity) moves: mean |delta| and fraction > 0.25 px, over all pixels and
on the golden's sub-grid.  Unlike the stereo-stage record of the golden this covers the motion stage (nearest-z splat,
`disp_warp > W -> 0`) and Fusion's saturated weight heads -- the selections that move isolated 4 x 4 blocks by tens of pixels.

    python tools/oracle_frame_probe.py STATE_DIR GOLDEN.npz F [F ...]      # K=3 NOISE=1e-6 OUT=probe.npz
"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import codd as oc

LONG = os.environ.get("LONG", "cfg3_50")
K = int(os.environ.get("K", "3"))
NOISE = float(os.environ.get("NOISE", "1e-6"))
OUT = os.environ.get("OUT", "")
# STATE_NOISE: the same relative noise on every floating tensor of the recurrent state as well (previous fused disparity,
# feature memory, RAFT3D's feature map / context features): two implementations reach frame F with states that differ at
# rounding level, and Fusion's weight heads read the warped MEMORY -- an image-only probe cannot see that dependence
STATE_NOISE = float(os.environ.get("STATE_NOISE", "0"))
case = T.LONG_CASES[LONG]
H, W, intr, _, _, _ = T.CASES[case[0]]
torch.set_num_threads(int(os.environ.get("THREADS", max(1, min(os.cpu_count() or 1, 16)))))


def states(d):
    """frame index -> snapshot path (the snapshot behind frame f holds {"f": f, "state": ...})"""
    out = {}
    for p in sorted(glob.glob(os.path.join(d, "*.pt"))):
        try:
            out[int(torch.load(p, map_location="cpu")["f"])] = p
        except Exception as e:  # a snapshot copied while it was being replaced
            print("skipping", p, e)
    return out


def main():
    sdir, golden = sys.argv[1], np.load(sys.argv[2])
    frames = [int(a) for a in sys.argv[3:]]
    sub = int(golden["sub"])
    snap = states(sdir)
    sd = T._build(False, case[1])[1]
    img, r_img, _ = synth.stereo_sequence(H, W, max(frames) + 1, **({"flow": case[3]} if len(case) > 3 else {}))
    res = {}
    for f in frames:
        if f - 1 not in snap:
            print(f"frame {f}: no snapshot behind frame {f - 1} in {sdir}")
            continue
        def noisy(v, gen):
            if torch.is_tensor(v):
                return v * (1 + STATE_NOISE * torch.randn(v.shape, generator=gen)) if v.is_floating_point() else v
            if isinstance(v, (list, tuple)):
                return type(v)(noisy(x, gen) for x in v)
            if isinstance(v, dict):
                return {k: noisy(x, gen) for k, x in v.items()}
            return v

        def run(l, r, gen=None):
            st = torch.load(snap[f - 1], map_location="cpu")["state"]
            if gen is not None and STATE_NOISE:
                st = noisy(st, gen)
            with torch.no_grad():
                return oc.frame(sd, l, r, st, intr, iters=case[1])["pred_disp"][0, 0]
        t0 = time.time()
        base = run(img[:, f], r_img[:, f])
        g = torch.from_numpy(golden[f"{LONG}_f{f}"]) if f"{LONG}_f{f}" in golden.files else None
        rep = (base[::sub, ::sub] - g).abs().max().item() if g is not None else float("nan")
        worst = np.zeros(4, np.float32)
        for k in range(K):
            gen = torch.Generator().manual_seed(1000 * f + k)
            l = img[:, f] * (1 + NOISE * torch.randn(img[:, f].shape, generator=gen))
            r = r_img[:, f] * (1 + NOISE * torch.randn(img[:, f].shape, generator=gen))
            d = (run(l, r, gen) - base).abs()
            ds = d[::sub, ::sub]
            row = np.array([d.mean().item(), (d > 0.25).float().mean().item(), ds.mean().item(), (ds > 0.25).float().mean().item()], np.float32)
            print(f"{LONG} frame {f} seed {k}: oracle output under {NOISE:g} input" + (f" + {STATE_NOISE:g} state" if STATE_NOISE else "") + f" noise moves by mean {row[0]:.2e} px, flipped {row[1]:.2e} "
                  f"(all pixels); sub-grid mean {row[2]:.2e}, flipped {row[3]:.2e}", flush=True)
            worst = np.maximum(worst, row)
        res[f] = worst
        print(f"{LONG} frame {f}: worst of {K}: all-pixel mean {worst[0]:.2e} flipped {worst[1]:.2e} | sub-grid mean {worst[2]:.2e} flipped {worst[3]:.2e}; "
              f"exact-input evaluation vs tracked golden max |delta| {rep:.1e}  [{time.time() - t0:.0f} s]", flush=True)
    if OUT:
        old = dict(np.load(OUT)) if os.path.exists(OUT) else {}
        old.update({f"{LONG}_frame_sens_f{f}": v for f, v in res.items()})
        old["frame_sens_noise"] = np.array([NOISE, STATE_NOISE], np.float32)
        np.savez(OUT, **old)


if __name__ == "__main__":
    main()
