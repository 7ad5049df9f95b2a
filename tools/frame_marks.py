"""dev: UN-PROFILED phase times of the captured frame -- one-thread timestamp launches (codd_timestamp: the device's 100 MHz
wall clock) at the seams of the frame on the caller's stream: frame start, stereo network issued / done, update loop start
/ end, frame end.  rocprofv3 inflates the per-node launch cost 3-4x and with it the head of the frame (DESIGN finding 47);
these six extra nodes do not.   python tools/frame_marks.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from codd_amd import configs, ops, synth
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H, W = 576, 960
dev = "cuda:0"
ops.enable_autotune(True, shipped=True)
est = build_estimator(configs.codd()).eval()
synth.load_synthetic_weights(est, 1.4)
est = est.to(dev)
img, r_img, _ = synth.stereo_sequence(H, W, 8)
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
marks = torch.zeros(16, dtype=torch.int64, device=dev)
NAMES = ["frame start", "stereo done", "loop start (first lookup)", "loop end", "motion done", "frame end"]

stereo_fn = est.stereo.stereo_matching
def stereo(*a, **k):
    out = stereo_fn(*a, **k)
    ops.timestamp(marks, 1)
    return out
est.stereo.stereo_matching = stereo
look = ops.raft_geometry_lookup
state = dict(n=0)
def lookup(*a, **k):
    if state["n"] == 0:
        ops.timestamp(marks, 2)
    state["n"] += 1
    return look(*a, **k)
ops.raft_geometry_lookup = lookup
cvx = ops.cvx_upsample_se3_weight
def cvx_(*a, **k):
    ops.timestamp(marks, 3)
    return cvx(*a, **k)
ops.cvx_upsample_se3_weight = cvx_
motion_fn = est.motion.forward
def motion(*a, **k):
    state["n"] = 0
    out = motion_fn(*a, **k)
    ops.timestamp(marks, 4)
    return out
est.motion.forward = motion
frame_fn = est.consistent_online_depth_estimation
def frame(*a, **k):
    ops.timestamp(marks, 0)
    out = frame_fn(*a, **k)
    ops.timestamp(marks, 5)
    return out
est.consistent_online_depth_estimation = frame

runner = FrameRunner(est, metas[0], use_graph=True)
acc = []
for f in range(N):
    l, r = img[:, f % 8].to(dev).contiguous(), r_img[:, f % 8].to(dev).contiguous()
    runner.step(l, r)
    if f >= N // 2:
        torch.cuda.synchronize()
        acc.append(marks[:6].cpu().numpy().copy())
a = np.array(acc, dtype=np.float64)
rel = (a - a[:, :1]) / 100.0  # us
med = np.median(rel, axis=0)
print("un-profiled frame (graph replay, synchronised after every frame: the launch of frame t + 1 does not overlap frame t):")
for n, v in zip(NAMES, med):
    print(f"  {n:28s} +{v:8.1f} us")
print(f"  stereo {med[1]:.0f} us | wait for the loop {med[2] - med[1]:.0f} | loop {med[3] - med[2]:.0f} ({(med[3] - med[2]) / 16:.1f} per update) | "
      f"up-sampling + splats {med[4] - med[3]:.0f} | fusion + write-back {med[5] - med[4]:.0f}")
# back-to-back replays (no synchronisation): frame period from consecutive frame-start marks cannot be read from one buffer;
# time it with events instead
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for f in range(40):
    runner.step(img[:, f % 8].to(dev).contiguous(), r_img[:, f % 8].to(dev).contiguous())
e.record(); torch.cuda.synchronize()
print(f"back-to-back: {s.elapsed_time(e) / 40:.3f} ms per frame")
