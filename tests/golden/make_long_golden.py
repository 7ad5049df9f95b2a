"""Generator of tests/golden/headline_oracle_long_sub4.npz -- TRACKED oracle outputs for the recurrence-length and
`iters = 1` parity cases of tests/test_gpu_headline_parity.py (round 4).

BASELINE.json configs[2] is "num_frames = -1": the reference walks up to 50 frames of one video through
`state["memory"]` (reference datasets/custom_stereo_mf.py:23,212-231; state loop model/codd.py:322-366), so a
selection flipped in frame t feeds every later frame.  The three-frame golden (make_headline_golden.py) cannot show
drift; this file holds the CPU oracle's (oracle/codd.py) disparity for

  * cfg3_long  : full CODD 960x576, iters = 16, frames 0 .. 15 of the synthetic video (the bench configuration)
  * cfg5_long  : full CODD 640x512 (TartanAir shape), iters = 16, frames 0 .. 15
  * cfg5_it1   : full CODD 640x512 (TartanAir shape), iters = 1 -- the value the reference configures for TartanAir
                 (reference configs/models/codd.py:6) -- frames 0 .. 5

on the sub-grid [::4, ::4] (fp32).  Frames of the synthetic video do not depend on the sequence length
(codd_amd/synth.py: frame t is a closed form of t), so cfg3_long's frames 0-2 must reproduce the committed
three-frame golden -- checked at the end of the run.

    python tests/golden/make_long_golden.py [cfg3_long] [cfg5_it1]        # ~70 min + ~5 min on 8 cores

Needs neither a GPU nor /root/reference.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# CODD_GOLDEN_OUT / CODD_GOLDEN_THREADS: write somewhere else with another CPU thread count -- used once to measure how
# far TWO ORACLE RUNS drift apart over the recurrence (different thread counts = different fp32 summation orders in
# torch's CPU convolutions): profiles/r04_oracle_self_divergence.log
OUT = os.environ.get("CODD_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden", "headline_oracle_long_sub4.npz"))
SUB = 4

# name -> (case of test_gpu_headline_parity.CASES that gives shape / intrinsics, iters, frames[, per-frame flow of the video])
LONG_CASES = {
    "cfg3_long": ("cfg3_codd_960x576", 16, 16),
    "cfg5_it1": ("cfg5_tartanair_640x512", 1, 6),
    "cfg5_long": ("cfg5_tartanair_640x512", 16, 16),
    # round 5: the reference's sequence cap (datasets/custom_stereo_mf.py:23: 50 frames per chunk) on a second synthetic
    # video whose per-frame shift (0.737, 0.263) px meets no half-pixel phase of the texture before t = 500
    "cfg3_50": ("cfg3_codd_960x576", 16, 50, (0.737, 0.263)),
}
# CODD_GOLDEN_VARIANT=nomkldnn: the same recurrence with every convolution on ATen's im2col + sgemm path instead of
# oneDNN (another fp32 summation order), stored as "<case>@nomkldnn_f<t>": the oracle's OWN sensitivity per frame, which
# the test uses to tell an ill-conditioned frame from a product defect
VARIANT = os.environ.get("CODD_GOLDEN_VARIANT", "")
FINE_FRAMES = 3
# Stereo-stage conditioning, recorded per frame beside the frames of the 50-frame case ("<case>_stereo_sens_f<t>" =
# [mean |delta|, fraction > 0.25 px] over all pixels, the larger of SENS_K seeds): how far the ORACLE's own stereo output
# (a per-frame function of the two images: tile cost-volume arg-mins of an untrained network) moves when the images
# carry SENS_NOISE relative noise -- ~10x the differences between two fp32 evaluation orders.  Costs two extra stereo
# evaluations (~2 x 4 s) per frame.  tools/video_margin_scan.py: 7-10 % of the frames of EVERY synthetic video tried
# (four flows, two textures) have such a near-tie, so the test reads this record instead of hoping for a clean video.
SENS_NOISE, SENS_K = 1e-6, 2


def stereo_sensitivity(sd, left, right, base, f):
    from oracle import stereo as ostereo
    worst = np.zeros(2, np.float32)
    for k in range(SENS_K):
        g = torch.Generator().manual_seed(100 * f + k)
        l = left * (1 + SENS_NOISE * torch.randn(left.shape, generator=g))
        r = right * (1 + SENS_NOISE * torch.randn(right.shape, generator=g))
        d = (ostereo.stereo_matching(sd, l, r, 320)["pred_disp"] - base).abs()
        worst = np.maximum(worst, np.array([d.mean().item(), (d > 0.25).float().mean().item()], np.float32))
    return worst


def merge(dst, src):
    """python make_long_golden.py --merge DST.npz SRC.npz: copy SRC's "<case>@<variant>_*" frames into DST."""
    a, b = np.load(dst), np.load(src)
    arrays = {k: a[k] for k in a.files}
    n = 0
    for k in b.files:
        if "@" in k:
            arrays[k] = b[k]
            n += 1
    np.savez_compressed(dst, **arrays)
    print(f"merged {n} variant arrays of {src} into {dst} ({os.path.getsize(dst)} bytes)")


def main():
    import test_gpu_headline_parity as T
    from codd_amd import synth
    from oracle import codd as oc
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", max(1, min(os.cpu_count() or 1, 16)))))
    arrays = {}
    if os.path.exists(OUT):
        old = np.load(OUT)
        arrays = {k: old[k] for k in old.files}
    import contextlib
    ctx = torch.backends.mkldnn.flags(enabled=False) if VARIANT == "nomkldnn" else contextlib.nullcontext()
    assert VARIANT in ("", "nomkldnn"), VARIANT
    for name, case in LONG_CASES.items():
        base, iters, MF = case[:3]
        if (only and name not in only) or (not only and len(case) > 3):
            continue
        H, W, intr, _, stereo_only, _ = T.CASES[base]
        sd = T._build(stereo_only)[1]
        img, r_img, _ = synth.stereo_sequence(H, W, MF, **({"flow": case[3]} if len(case) > 3 else {}))
        MF = min(MF, int(os.environ.get("CODD_GOLDEN_FRAMES", MF)))
        key = name + ("@" + VARIANT if VARIANT else "")
        state, f0 = {}, 0
        # resume: the recurrent state is saved beside OUT after every frame (git-ignored scratch; a 50-frame run is 2-3 h
        # of CPU and background jobs do not survive a session restart)
        resume = OUT + f".{key}.state.pt"
        if os.path.exists(resume):
            ck = torch.load(resume)
            if all(f"{key}_f{q}" in arrays for q in range(ck["f"] + 1)):
                state, f0 = ck["state"], ck["f"] + 1
                print(f"{key}: resuming behind frame {ck['f']} from {resume}", flush=True)
        with torch.no_grad(), ctx:
            for f in range(f0, MF):
                t0 = time.time()
                o = oc.frame(sd, img[:, f], r_img[:, f], state, intr, iters=iters, with_motion=True, with_fusion=True)
                a = o["pred_disp"][0, 0, ::SUB, ::SUB].contiguous().numpy().astype(np.float32)
                arrays[f"{key}_f{f}"] = a
                if len(case) > 3 and f < FINE_FRAMES:  # the first frames of the 50-frame case also on the finer sub-grid [::2, ::2]
                    arrays[f"{key}_sub2_f{f}"] = o["pred_disp"][0, 0, ::2, ::2].contiguous().numpy().astype(np.float32)
                if len(case) > 3 and not VARIANT:
                    arrays[f"{key}_stereo_sens_f{f}"] = stereo_sensitivity(sd, img[:, f], r_img[:, f], o["pred_curr"] if "pred_curr" in o else o["pred_disp"], f)
                    print(key, f, "stereo-stage movement under %g input noise: mean %.2e px, flipped %.2e" % (SENS_NOISE, *arrays[f"{key}_stereo_sens_f{f}"]), flush=True)
                print(key, f, a.shape, float(a.mean()), f"{time.time() - t0:.0f} s", flush=True)
                # checkpoint after every frame: the run takes an hour
                np.savez_compressed(OUT + ".tmp.npz", **{**arrays, "sub": np.array(SUB), "src_hash": np.array(T._src_hash())})
                os.replace(OUT + ".tmp.npz", OUT)
                torch.save(dict(f=f, state=state), resume + ".tmp")
                os.replace(resume + ".tmp", resume)
        if iters == T.ITERS and not VARIANT and len(case) == 3:
            short = np.load(T.GOLDEN)
            for f in range(T.CASES[base][5]):
                d = np.abs(short[f"{base}_f{f}"] - arrays[f"{name}_f{f}"])
                print(f"{name} frame {f} vs committed three-frame golden: max |delta| {d.max():.3e}")
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--merge":
        merge(sys.argv[2], sys.argv[3])
    else:
        main()
