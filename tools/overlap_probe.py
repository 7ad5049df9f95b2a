"""dev: do the frame's image-only chains (stereo, fnet, cnet) overlap when launched as separate graphs on
separate streams?  Times each graph alone and the three together."""
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
runner = FrameRunner(est, metas[0], use_graph=True, split=True)
for i in range(4):
    runner.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
torch.cuda.synchronize()
st = runner._static
s0, s1, s2 = st["streams"]
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def on(s, g):
    with torch.cuda.stream(s): g.replay()
print("stereo alone   %.3f ms" % timeit(lambda: on(s0, st["g_s"])))
print("fnet alone     %.3f ms" % timeit(lambda: on(s1, st["g_f"])))
print("cnet alone     %.3f ms" % timeit(lambda: on(s2, st["g_c"])))
print("motion+fusion  %.3f ms" % timeit(lambda: on(s0, st["g_m"])))
print("fnet + cnet    %.3f ms" % timeit(lambda: (on(s1, st["g_f"]), on(s2, st["g_c"]))))
print("all three      %.3f ms" % timeit(lambda: (on(s0, st["g_s"]), on(s1, st["g_f"]), on(s2, st["g_c"]))))
print("stereo ; fnet ; cnet on ONE stream  %.3f ms" % timeit(lambda: (on(s0, st["g_s"]), on(s0, st["g_f"]), on(s0, st["g_c"]))))
print("motion + cnet concurrently %.3f ms" % timeit(lambda: (on(s0, st["g_m"]), on(s2, st["g_c"]))))
print("motion + stereo concurrently %.3f ms" % timeit(lambda: (on(s0, st["g_m"]), on(s1, st["g_s"]))))
print("motion + stereo + fnet + cnet concurrently %.3f ms" % timeit(lambda: (on(s0, st["g_m"]), on(s1, st["g_s"]), on(s2, st["g_c"]), on(s2, st["g_f"]))))
