"""Adds full-resolution oracle frames ("<case>_full_f<t>", fp32 [H, W]) of selected frames to a long golden: frame t is
re-evaluated from the oracle's saved recurrent state behind frame t - 1 (make_long_golden.py keeps "<out>.<case>.state.pt";
round 5 kept one snapshot per frame) and must reproduce the tracked sub-grid frame bit for bit.
    python tests/golden/add_full_frames.py STATE_DIR GOLDEN.npz cfg3_50 20 [21 ...]        # ~4 min per frame on 8 cores
"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import test_gpu_headline_parity as T
    import oracle_frame_probe as P
    from codd_amd import synth
    from oracle import codd as oc
    sdir, path, name = sys.argv[1:4]
    frames = [int(a) for a in sys.argv[4:]]
    torch.set_num_threads(int(os.environ.get("CODD_GOLDEN_THREADS", "8")))
    case = T.LONG_CASES[name]
    H, W, intr = T.CASES[case[0]][:3]
    z = np.load(path)
    arrays = {k: z[k] for k in z.files}
    sub = int(z["sub"])
    snap = P.states(sdir)
    sd = T._build(False, case[1])[1]
    img, r_img, _ = synth.stereo_sequence(H, W, max(frames) + 1, **({"flow": case[3]} if len(case) > 3 else {}))
    for f in frames:
        st = torch.load(snap[f - 1], map_location="cpu")["state"]
        with torch.no_grad():
            d = oc.frame(sd, img[:, f], r_img[:, f], st, intr, iters=case[1])["pred_disp"][0, 0].contiguous().numpy().astype(np.float32)
        dev = np.abs(d[::sub, ::sub] - arrays[f"{name}_f{f}"]).max()
        print(f"{name} frame {f}: full-resolution frame from the saved state; sub-grid vs tracked golden max |delta| {dev:.1e}", flush=True)
        assert dev == 0.0, "the saved state does not reproduce the tracked trajectory"
        arrays[f"{name}_full_f{f}"] = d
        np.savez_compressed(path + ".tmp.npz", **arrays)
        os.replace(path + ".tmp.npz", path)


if __name__ == "__main__":
    main()
