import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner
ops.enable_autotune(True, shipped=True)
n0 = len(ops.TUNE_DB)
for name, (H, W, intr, img_shape, stereo_only, MF) in T.CASES.items():
    for iters in ((16, 1) if name.startswith("cfg5") else (16,)):
        est = T._build(stereo_only, iters)[0].to("cuda:0")
        img, r_img, _ = synth.stereo_sequence(H, W, 3)
        metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
        runner = FrameRunner(est, metas[0], use_graph=False)
        for f in range(3):
            runner.step(img[:, f].to("cuda:0").contiguous(), r_img[:, f].to("cuda:0").contiguous())
        print(name, iters, "db entries", len(ops.TUNE_DB), "tuned on the fly so far", len(ops.TUNE_DB) - n0, flush=True)
for r in ops.AUTOTUNE_LOG[:20]:
    print(r[0])
print("all-pairs picks (D, h, w) -> (xb, th, ck, mb, layout, pgw, cgw, terms, ks):")
for k, v in sorted(ops._ALLPAIRS_PICK.items()):
    print("  ", k, v)
