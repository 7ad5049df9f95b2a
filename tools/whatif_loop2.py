"""Dev tool: upper bounds for co-launching the update block's independent convolutions -- the frame rate with the flow
encoder's 7x7 convolution / the next update's z|r convolution REMOVED (results are wrong on purpose)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import configs, motion, ops, synth  # noqa: E402
from codd_amd.registry import build_estimator  # noqa: E402
from codd_amd.runtime import FrameRunner  # noqa: E402

H, W = 576, 960
ops.enable_autotune(True, shipped=True)
est = build_estimator(configs.codd()).eval()
synth.load_synthetic_weights(est, 1.4)
est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 6)
img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
ub = est.motion.raft3d.update_block


def fps(tag, n=100):
    est.__dict__.pop("_runners", None)
    r = FrameRunner(est, metas[0], use_graph=True)
    for i in range(30):
        r.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        r.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag:64s} {dt:7.3f} ms/frame  {1e3 / dt:6.1f} frames/s", flush=True)


fps("baseline (loop forks on)")
ub._fk = None
motion.LOOP_FORK_ENC = motion.LOOP_FORK_ZR = False
fps("loop forks off")
orig_cv = motion.cv


def cv_skip(m, *a, **k):
    if m is ub.flow_enc[0]:
        return None
    return orig_cv(m, *a, **k)


motion.cv = cv_skip
fps("loop forks off, without the flow encoder's 7x7 convolution")
orig_gate = ops.conv_gate


def gate_skip(pc, xs, gate, **k):
    if gate == 1:
        return k["out"]
    return orig_gate(pc, xs, gate, **k)


ops.conv_gate = gate_skip
fps("loop forks off, without flow 7x7 and the z|r convolution")
motion.cv = orig_cv
fps("loop forks off, without the z|r convolution")
