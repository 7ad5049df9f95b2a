"""dev (GPU): two videos through the SAME launches (B = 2, lock-step) -- VERDICT r4 item 4.
Runs video 0 and video 1 alone (B = 1, two FrameRunners, graph replay) and stacked as one B = 2 batch through one
FrameRunner; prints per-frame equality of each video's disparities and the three frame rates.
Usage: python tools/b2_probe.py [frames] [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codd_amd  # noqa: F401
from codd_amd import configs, ops, synth
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (576, 960)
dev = torch.device("cuda:0")
MF = 6
vids = [synth.stereo_sequence(H, W, MF, flow=(0.75 + 0.125 * v, 0.25)) for v in range(2)]
metas = synth.default_metas(H, W, img_shape=(540, 960, 3) if (H, W) == (576, 960) else (H, W, 3))
ops.enable_autotune(True, shipped=True)


def model():
    est = build_estimator(configs.codd(iters=16)).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    return est.to(dev)


def frames(v, i):
    return vids[v][0][:, i % MF].to(dev).contiguous(), vids[v][1][:, i % MF].to(dev).contiguous()


singles = []
for v in range(2):
    r = FrameRunner(model(), metas[0], use_graph=True)
    outs = [r.step(*frames(v, i)).clone() for i in range(5)]
    singles.append((r, outs))
rb = FrameRunner(model(), metas[0], use_graph=True)
for i in range(5):
    l = torch.cat([frames(0, i)[0], frames(1, i)[0]], 0)
    rr = torch.cat([frames(0, i)[1], frames(1, i)[1]], 0)
    d = rb.step(l, rr)
    for v in range(2):
        a, b = singles[v][1][i], d[v:v + 1]
        diff = (a - b).abs()
        print(f"frame {i} video {v}: B=2 vs B=1 equal {torch.equal(a, b)}  mean |delta| {diff.mean().item():.3e}  max {diff.max().item():.3e}  "
              f"flipped {(diff > 0.25).float().mean().item():.2e}", flush=True)


def rate(fn, n):
    for i in range(10):
        fn(5 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(15 + i)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


ins1 = [frames(0, i) for i in range(MF)]
ins2 = [(torch.cat([frames(0, i)[0], frames(1, i)[0]], 0), torch.cat([frames(0, i)[1], frames(1, i)[1]], 0)) for i in range(MF)]
f1 = rate(lambda i: singles[0][0].step(*ins1[i % MF]), N)
f2 = rate(lambda i: rb.step(*ins2[i % MF]), N)
print(f"B=1: {f1:.2f} frames/s   B=2 lock-step: {f2:.2f} steps/s = {2 * f2:.2f} frames/s per GPU  (x{2 * f2 / f1:.3f})")
