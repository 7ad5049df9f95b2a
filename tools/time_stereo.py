"""dev: time the HITNetMF forward at 960x576 on the GPU (eager and hipGraph replay)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codd_amd
from codd_amd import synth
from codd_amd.registry import build_estimator

H, W = 576, 960
cfg = dict(type="ConsistentOnlineDynamicDepth", stereo=dict(type="HITNetMF", backbone=dict(type="HITUNet"),
           initialization=dict(type="TileInitialization", max_disp=320), propagation=dict(type="TilePropagation")))
est = build_estimator(cfg).eval()
synth.load_synthetic_weights(est, 1.4)
est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 1)
l, r = img[:, 0].cuda(), r_img[:, 0].cuda()
for _ in range(3):
    out = est.stereo.stereo_matching(l, r)
torch.cuda.synchronize()
t = time.perf_counter()
N = 10
for _ in range(N):
    out = est.stereo.stereo_matching(l, r)
torch.cuda.synchronize()
print("eager ms/frame", (time.perf_counter() - t) / N * 1e3)
if "--graph" in sys.argv:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            out = est.stereo.stereo_matching(l, r)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = est.stereo.stereo_matching(l, r)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        g.replay()
    torch.cuda.synchronize()
    print("graph ms/frame", (time.perf_counter() - t) / N * 1e3)
    print("disp mean", out["pred_disp"].mean().item())
