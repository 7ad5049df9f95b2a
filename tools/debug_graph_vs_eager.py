import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, synth, ops
from codd_amd.registry import build_estimator
from codd_amd.runtime import FrameRunner
H, W, iters = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (128, 384, 4)
serial = len(sys.argv) > 4 and sys.argv[4] == "serial"
ops.Fork.serial = serial
est = build_estimator(configs.codd(iters=iters)).eval(); synth.load_synthetic_weights(est, 1.4); est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 6); img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, intrinsics=(240.0, 240.0, W / 2.0, H / 2.0))
re, rg = FrameRunner(est, metas[0], use_graph=False), FrameRunner(est, metas[0], use_graph=True)
for f in range(8):
    l, r = img[:, f % 6].contiguous(), r_img[:, f % 6].contiguous()
    de = re.step(l, r).clone()
    dg = rg.step(l, r).clone()
    torch.cuda.synchronize()
    print(f, "graph" if rg.graph is not None else "eager", "max diff", (de - dg).abs().max().item(), "mean", (de - dg).abs().mean().item(), flush=True)
