"""dev: per-layer time of the best exact-fp32 configuration against the best split-bf16 (3-term) one for EVERY
convolution of the frame (stage policy off, tuner from scratch) -- the autotuner measures both for each layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner

ops._STAGE_PRECISION.clear()
ops.USE_ROLL = False
ops.enable_autotune(True, shipped=False)
H, W, intr, img_shape, _, _ = T.CASES["cfg3_codd_960x576"]
est = T._build(False, 16)[0].to("cuda:0")
img, r_img, _ = synth.stereo_sequence(H, W, 2)
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
runner = FrameRunner(est, metas[0], use_graph=False)
for f in range(2):
    runner.step(img[:, f].to("cuda:0").contiguous(), r_img[:, f].to("cuda:0").contiguous())
torch.cuda.synchronize()
rows = [r for r in ops.AUTOTUNE_LOG if r[0].startswith("choice")]
print("layer | best fp32 us | best of (fp32, split incl. re-layout) us | chosen")
tot32 = totb = 0.0
for name, f32cfg, t32, cfg, tbest in sorted(rows, key=lambda r: -r[2]):
    print(f"{name:44s} {t32:8.1f} {tbest:8.1f}  {cfg}")
