"""Generator of tests/golden/headline_oracle_conditioned.npz -- TRACKED oracle frames of the WELL-CONDITIONED parity cases of
tests/test_gpu_headline_parity.py::test_conditioned_sequence_meets_north_star_bound (round 6).

Same recurrence as make_long_golden.py (oracle/codd.py: stereo -> motion -> fusion, iters = 16, max_disp = 320, reference
model/codd.py:322-366) on
  * weights  codd_amd.synth.load_synthetic_weights(mode="conditioned"): the closed-form block-matcher weight set
  * video    synth.stereo_sequence(texture="waves", left_taper=96, flow=(0.737, 0.263)): broadband texture, every left pixel
             has its match inside the right image
so that the oracle's own fp32 evaluations agree to ~1e-5 px on every frame (tools/cond_probe.py) and every frame can be
held to the north-star bound against ONE tracked trajectory, without conditioning rules.

  cfg3_50c  960x576 (the benchmarked shape), 50 frames = the reference's sequence cap (datasets/custom_stereo_mf.py:23), sub-grid [::6, ::6]
  cfg5_16c  640x512 (TartanAir shape), 16 frames, sub-grid [::4, ::4]

    python tests/golden/make_conditioned_golden.py <case> ...      env: CODD_GOLDEN_OUT, CODD_GOLDEN_THREADS, CODD_GOLDEN_FRAMES,
                                                                        CODD_GOLDEN_VARIANT=nomkldnn (ATen im2col + sgemm convolutions)
    python tests/golden/make_conditioned_golden.py --pack OUT.npz TRACKED.npz [name=VARIANT.npz[:key suffix] ...]
Runs are resumable (state saved beside OUT after every frame).  Needs neither a GPU nor /root/reference.
"""
import contextlib
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.environ.get("CODD_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden", "headline_oracle_conditioned.npz"))
VARIANT = os.environ.get("CODD_GOLDEN_VARIANT", "")


def main():
    import test_gpu_headline_parity as T
    from oracle import codd as oc
    torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", max(1, min(os.cpu_count() or 1, 16)))))
    assert VARIANT in ("", "nomkldnn"), VARIANT
    if os.environ.get("CODD_GOLDEN_FTZ") == "1":
        # ATen's sgemm path computes with denormals enabled, and the conditioned network's leaky-ReLU chains push some activations
        # below 1e-38: from frame ~11 on a frame of the 960x576 case took 12 minutes instead of 2.  Flush-to-zero is a
        # legitimate fp32 evaluation too (differences ~1e-38); the resumed part of the @nomkldnn table was computed with it.
        torch.set_flush_denormal(True)
    arrays = {}
    if os.path.exists(OUT):
        old = np.load(OUT)
        arrays = {k: old[k] for k in old.files}
    for name in sys.argv[1:]:
        base, iters, MF, sub = T.COND_CASES[name]
        H, W, intr, _, _, _ = T.CASES[base]
        sd = T._build(False, iters, mode="conditioned")[1]
        img, r_img, _ = T.conditioned_video(H, W, MF)
        MF = min(MF, int(os.environ.get("CODD_GOLDEN_FRAMES", MF)))
        key = name + ("@" + VARIANT if VARIANT else "")
        state, f0 = {}, 0
        # (CODD_GOLDEN_STATE_DIR: keep the ~200 MB resume files out of gpurun_out/ on the GPU box -- its merge-back limit is 64 MiB)
        resume = os.path.join(os.environ.get("CODD_GOLDEN_STATE_DIR", os.path.dirname(OUT)), os.path.basename(OUT) + f".{key}.state.pt")
        if os.path.exists(resume):
            ck = torch.load(resume)
            if all(f"{key}_f{q}" in arrays for q in range(ck["f"] + 1)):
                state, f0 = ck["state"], ck["f"] + 1
                print(f"{key}: resuming behind frame {ck['f']}", flush=True)
        ctx = torch.backends.mkldnn.flags(enabled=False) if VARIANT == "nomkldnn" else contextlib.nullcontext()
        with torch.no_grad(), ctx:
            for f in range(f0, MF):
                t0 = time.time()
                o = oc.frame(sd, img[:, f], r_img[:, f], state, intr, iters=iters, with_motion=True, with_fusion=True)
                a = o["pred_disp"][0, 0, ::sub, ::sub].contiguous().numpy().astype(np.float32)
                arrays[f"{key}_f{f}"] = a
                extra = ""
                if f > 0:
                    extra = f" fusion weight {o['fusion_weights'].mean().item():.3f} reset weight {o['reset_weights'].mean().item():.3f} |fused - curr| {(o['pred_disp'] - o['pred_curr']).abs().mean().item():.3f} px"
                print(f"{key} frame {f}: mean disparity {float(a.mean()):.4f}{extra}  [{time.time() - t0:.0f} s on {torch.get_num_threads()} threads]", flush=True)
                np.savez_compressed(OUT + ".tmp.npz", **{**arrays, f"{name}_sub": np.array(sub), "src_hash": np.array(T._src_hash())})
                os.replace(OUT + ".tmp.npz", OUT)
                torch.save(dict(f=f, state=state), resume + ".tmp")
                os.replace(resume + ".tmp", resume)
    print("wrote", OUT, os.path.getsize(OUT))


def pack():
    """tracked frames + "<case>@<variant>_env" = per-frame [mean |delta|, fraction > 0.25 px] of every other fp32 evaluation of the
    oracle against the tracked one (NaN: not computed on that frame)"""
    out, tracked = sys.argv[2:4]
    m = np.load(tracked)
    arrays = {k: m[k] for k in m.files if "@" not in k}
    cases = sorted({k[:-4] for k in arrays if k.endswith("_sub")})
    for spec in sys.argv[4:]:
        vname, rest = spec.split("=")
        path, _, suffix = rest.partition(":")
        v = np.load(path)
        for c in cases:
            MF = sum(1 for k in arrays if re.fullmatch(re.escape(c) + r"_f\d+", k))
            env = np.full((MF, 2), np.nan, np.float32)
            for f in range(MF):
                k = f"{c}{suffix}_f{f}"
                if k in v.files:
                    d = np.abs(v[k] - arrays[f"{c}_f{f}"])
                    env[f] = (d.mean(), (d > 0.25).mean())
            n = int(np.isfinite(env[:, 0]).sum())
            if n:
                arrays[f"{c}@{vname}_env"] = env
                print(f"{c}@{vname}: {n} of {MF} frames; oracle-vs-oracle mean |delta| max {np.nanmax(env[:, 0]):.2e} px, flipped fraction max {np.nanmax(env[:, 1]):.2e}")
    np.savez_compressed(out, **arrays)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    pack() if len(sys.argv) > 3 and sys.argv[1] == "--pack" else main()
