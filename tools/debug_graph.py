"""dev: graph-replay soak test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, synth
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner
H, W = 576, 960
est = build_estimator(configs.codd()).eval(); synth.load_synthetic_weights(est, 1.4); est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 6); img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
runner = FrameRunner(est, metas[0], use_graph=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sync_every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for i in range(N):
    d = runner.step(img[:, i % 6].contiguous(), r_img[:, i % 6].contiguous())
    if i % sync_every == 0:
        torch.cuda.synchronize()
        print(i, float(d.mean()), flush=True)
