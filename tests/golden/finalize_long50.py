"""Packs the 50-frame golden of tests/test_gpu_headline_parity.py::test_recurrent_sequence_matches_oracle[cfg3_50] from the
generator's outputs (make_long_golden.py):
    MAIN      the tracked oracle trajectory ("cfg3_50_f<t>", "_sub2_f<t>", "_stereo_sens_f<t>", "_full_f<t>")
    VARIANTS  name=file.npz[:key prefix] ...   other fp32 evaluations of the SAME oracle recurrence:
              nomkldnn        every convolution on ATen's im2col + sgemm path instead of oneDNN (this container's CPU)
              hostB           the default evaluation on the GPU box's host CPU (another oneDNN code path; frames 0..22)
              hostB_nomkldnn  the sgemm evaluation on that host (frames 0..21)
For every variant the golden keeps a per-frame table "<case>@<variant>_env" = [mean |delta|, fraction > 0.25 px] against the
tracked frames (NaN where the variant was not computed) and the variant's FRAMES only where they carry information
(mean |delta| >= 1e-3 / 3: the frames on which the oracle's own evaluations disagree) -- 138 KB per frame saved elsewhere.
    python tests/golden/finalize_long50.py OUT.npz MAIN.npz nomkldnn=ALT.npz hostB=BOX_MAIN.npz:cfg3_50 hostB_nomkldnn=BOX_ALT.npz:cfg3_50@nomkldnn
"""
import sys
import numpy as np

NAME, KEEP = "cfg3_50", 1e-3 / 3


def main():
    out, main_p = sys.argv[1:3]
    m = np.load(main_p)
    arrays = {k: m[k] for k in m.files if "@" not in k}
    import re
    MF = sum(1 for k in arrays if re.fullmatch(NAME + r"_f\d+", k))
    for spec in sys.argv[3:]:
        vname, rest = spec.split("=")
        path, _, prefix = rest.partition(":")
        prefix = prefix or f"{NAME}@{vname}"
        v = np.load(path)
        env = np.full((MF, 2), np.nan, np.float32)
        kept = []
        for f in range(MF):
            if f"{prefix}_f{f}" in v.files:
                d = np.abs(v[f"{prefix}_f{f}"] - arrays[f"{NAME}_f{f}"])
                env[f] = (d.mean(), (d > 0.25).mean())
                if env[f, 0] >= KEEP:
                    arrays[f"{NAME}@{vname}_f{f}"] = v[f"{prefix}_f{f}"]
                    kept.append(f)
                if f < 3 and f"{prefix}_sub2_f{f}" in v.files and env[f, 0] >= KEEP:
                    arrays[f"{NAME}@{vname}_sub2_f{f}"] = v[f"{prefix}_sub2_f{f}"]
        arrays[f"{NAME}@{vname}_env"] = env
        n = int(np.isfinite(env[:, 0]).sum())
        print(f"{vname}: {n} frames compared, {len(kept)} kept (oracle-vs-oracle >= {KEEP:.1e} px): {kept}")
    np.savez_compressed(out, **arrays)
    import os
    print("wrote", out, os.path.getsize(out), "bytes;", MF, "tracked frames")


if __name__ == "__main__":
    main()
