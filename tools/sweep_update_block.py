"""dev (GPU): every launch configuration the tuner times for the split-bf16 convolutions of ONE steady-state frame, per layer --
configuration (xb, th, ck, mb, layout, pgw, cgw, terms, ksplit), workgroups of the launch, microseconds (best of two bursts
of three launches) -- so that a layer's dispatch-round quantisation can be read off directly.

    python tools/sweep_update_block.py [H W]          # default 576 960; FILTER=<substring of the layer description>
"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import configs, ops, synth  # noqa: E402
from codd_amd.registry import build_estimator  # noqa: E402
from codd_amd.runtime import FrameRunner  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (576, 960)
flt = os.environ.get("FILTER", "")
dev = torch.device("cuda:0")
est = build_estimator(configs.codd(iters=16)).eval()
synth.load_synthetic_weights(est, gain=1.4)
est = est.to(dev)
img, r_img, _ = synth.stereo_sequence(H, W, 2)
metas = synth.default_metas(H, W)
ops.enable_autotune(True, shipped=False)  # time everything
ops.AUTOTUNE_TRACE = []
runner = FrameRunner(est, metas[0], use_graph=False)
for f in range(2):
    runner.step(img[:, f].to(dev).contiguous(), r_img[:, f].to(dev).contiguous())
torch.cuda.synchronize()
layers = OrderedDict()
for desc, cfg, grid, us in ops.AUTOTUNE_TRACE:
    layers.setdefault(desc, []).append((us, grid, cfg))
for desc, rows in layers.items():
    if flt and flt not in desc:
        continue
    rows.sort()
    print(f"== {desc}: {len(rows)} configurations")
    for us, grid, cfg in rows[:12]:
        print(f"   {us:7.1f} us  {grid:4d} workgroups ({grid / 256:.2f} rounds)  cfg {cfg}")
