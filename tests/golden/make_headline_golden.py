"""Generator of tests/golden/headline_oracle_sub4.npz -- the TRACKED oracle outputs behind tests/test_gpu_headline_parity.py.

Runs the CPU oracle (oracle/codd.py: restatement of reference model/codd.py:80-126, iters = 16, max_disp = 320) on
the seedless synthetic sequences of the four benchmarked configurations (BASELINE.json configs[1..4]) with the
closed-form weight filler (codd_amd/synth.py), and stores every frame's disparity SUB-SAMPLED to every 4th pixel in
both directions (fp32, [H/4, W/4]): 11 frames, ~1.4 MB.  The GPU test compares the same sub-grid of the product
path's output against it, so the headline parity evidence no longer rests on the untracked full-resolution cache
(tests/_oracle_cache/, still used when present and recomputed for the smallest case on a fresh box).

    python tests/golden/make_headline_golden.py            # recompute everything (~30 min on 8 cores)
    python tests/golden/make_headline_golden.py --check    # recompute and compare with the committed file

Needs neither a GPU nor /root/reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "headline_oracle_sub4.npz")
SUB = 4


def main():
    import test_gpu_headline_parity as T
    from codd_amd import synth
    from oracle import codd as oc
    check = "--check" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    arrays = {}
    for name, (H, W, intr, _, stereo_only, MF) in T.CASES.items():
        if only and name not in only:
            continue
        sd = T._build(stereo_only)[1]
        img, r_img, _ = synth.stereo_sequence(H, W, MF)
        state = {}
        with torch.no_grad():
            for f in range(MF):
                o = oc.frame(sd, img[:, f], r_img[:, f], state, intr, iters=T.ITERS, with_motion=not stereo_only,
                             with_fusion=not stereo_only)
                arrays[f"{name}_f{f}"] = o["pred_disp"][0, 0, ::SUB, ::SUB].contiguous().numpy().astype(np.float32)
                print(name, f, arrays[f"{name}_f{f}"].shape, float(arrays[f"{name}_f{f}"].mean()), flush=True)
    if check:
        ref = np.load(OUT)
        for k, v in arrays.items():
            d = np.abs(ref[k] - v)
            print(f"{k}: max |delta| vs committed {d.max():.3e}  mean {d.mean():.3e}")
        return
    if only and os.path.exists(OUT):  # partial regeneration keeps the other cases
        old = np.load(OUT)
        arrays = {**{k: old[k] for k in old.files}, **arrays}
    arrays["sub"] = np.array(SUB)
    arrays["src_hash"] = np.array(T._src_hash())
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
