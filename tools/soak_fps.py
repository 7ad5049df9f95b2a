import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, synth, ops
ops.enable_autotune(True, shipped=True)
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner
H, W = 576, 960
est = build_estimator(configs.codd()).eval(); synth.load_synthetic_weights(est, 1.4); est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 6); img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
runner = FrameRunner(est, metas[0], use_graph=True)
for i in range(3):
    runner.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
torch.cuda.synchronize()
t00 = time.perf_counter()
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t0 = time.perf_counter()
    for i in range(50):
        runner.step(img[:, i % 6], r_img[:, i % 6])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"t={time.perf_counter()-t00:6.1f}s  block {blk}: {50/dt:.2f} fps", flush=True)
# host-side cost of one step (input copies + graph launch), GPU idle at the start of each measurement
ts = []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.step(img[:, i % 6], r_img[:, i % 6])
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host time of step() with an idle GPU: min %.2f ms  median %.2f ms" % (min(ts) * 1e3, sorted(ts)[10] * 1e3))
