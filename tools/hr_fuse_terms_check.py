import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_headline_parity as T
from codd_amd import synth, hrnet, ops
est = T._build(False, 16)[0].to("cuda:0")
img, _, _ = synth.stereo_sequence(512, 640, 1)
x = img[:, 0].to("cuda:0")
r3 = est.motion.raft3d
outs = []
for flag in (False, True):
    hrnet.FUSE_TERMS = flag
    with ops.stage("context"):
        outs.append(r3.context(x).clone())
print("bit-identical:", torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item())
