"""Adds the per-frame stereo-stage conditioning record ("<case>_stereo_sens_f<t>", see make_long_golden.py) to an existing
long golden WITHOUT re-running its recurrence: the stereo stage is a per-frame function of the two images.
    python tests/golden/add_stereo_sens.py cfg3_long cfg5_long          # ~10 min on 8 cores
"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import test_gpu_headline_parity as T
    import make_long_golden as G
    from codd_amd import synth
    from oracle import stereo as ostereo
    torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", "8")))
    for name in sys.argv[1:]:
        case = T.LONG_CASES[name]
        path = case[4] if len(case) > 4 else T.LONG_GOLDEN
        z = np.load(path)
        arrays = {k: z[k] for k in z.files}
        H, W = T.CASES[case[0]][:2]
        MF = T.n_frames(z, name)
        sd = T._build(False, case[1])[1]
        img, r_img, _ = synth.stereo_sequence(H, W, MF, **({"flow": case[3]} if len(case) > 3 else {}))
        with torch.no_grad():
            for f in range(MF):
                base = ostereo.stereo_matching(sd, img[:, f], r_img[:, f], 320)["pred_disp"]
                arrays[f"{name}_stereo_sens_f{f}"] = G.stereo_sensitivity(sd, img[:, f], r_img[:, f], base, f)
                print(name, f, "stereo-stage movement under %g input noise: mean %.2e px, flipped %.2e" % (G.SENS_NOISE, *arrays[f"{name}_stereo_sens_f{f}"]), flush=True)
        np.savez_compressed(path, **arrays)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
