"""Dev probe: per-frame wall time of the first frames after a device sync (does a K = 20 timed region pay a ramp?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, ops, synth
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner
ops.enable_autotune(True, shipped=True)
est = build_estimator(configs.codd()).eval(); synth.load_synthetic_weights(est, 1.4); est = est.cuda()
H, W = 576, 960
img, r_img, _ = synth.stereo_sequence(H, W, 6); img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
r = FrameRunner(est, metas[0], use_graph=True)
for i in range(160):
    r.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
for idle_ms in (0, 5, 50):
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(40):
        r.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
        if i in (0, 4, 9, 19, 39):
            torch.cuda.synchronize()
            marks.append((i + 1, time.perf_counter() - t0))
    print(f"idle {idle_ms} ms before: " + "  ".join(f"{n} frames {1e3 * t / n:.2f} ms/frame" for n, t in marks))
