"""dev: which launch configuration does every exact-fp32 convolution of one frame run with?  (Launches whose
configuration is the multi-job class (npb 1, nw 4, mb 1, quad layout) can be merged into multi-job launches without
changing a bit.)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner
ops.enable_autotune(True, shipped=True)
H, W, intr, img_shape, _, _ = T.CASES["cfg3_codd_960x576"]
est = T._build(False, 16)[0].to("cuda:0")
img, r_img, _ = synth.stereo_sequence(H, W, 3)
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
runner = FrameRunner(est, metas[0], use_graph=False)
for f in range(2):
    runner.step(img[:, f].to("cuda:0").contiguous(), r_img[:, f].to("cuda:0").contiguous())
recs = []
orig = ops._launch_conv
def hook(lib, p, stream):
    recs.append((p.layout, p.npb, p.nw, p.mb, p.ck, p.C0 + p.C1, p.Cout, p.kh, p.Hout, p.Wout, p.sy, p.terms))
    return orig(lib, p, stream)
ops._launch_conv = hook
runner.step(img[:, 2].to("cuda:0").contiguous(), r_img[:, 2].to("cuda:0").contiguous())
ops._launch_conv = orig
fp32 = [r for r in recs if r[0] != 2]
inclass = [r for r in fp32 if r[:4] == (1, 1, 4, 1)]
print("single conv launches:", len(recs), "exact-fp32:", len(fp32), "in the multi-job class (layout 1, npb 1, nw 4, mb 1):", len(inclass))
c = collections.Counter((r[5], r[6], r[7], r[8], r[9], r[10], r[4]) for r in inclass)
for k, n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"  n={n:2d}  cin {k[0]:3d} cout {k[1]:3d} k{k[2]} out {k[3]}x{k[4]} s{k[5]} ck{k[6]}")
