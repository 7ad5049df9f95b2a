"""dev: HRNet branch convolutions (18/36/72/144 channels) under every quad-layout launch class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
from codd_amd import ops
dev = "cuda:0"
ops.set_conv_precision("fp32")
warnings.simplefilter("error")

def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

g = torch.Generator().manual_seed(0)
for C, H, W in [(18, 144, 240), (36, 72, 120), (72, 36, 60), (144, 18, 30)]:
    x = torch.randn(1, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev)
    pc = ops.PackedConv(w, torch.zeros(C, device=dev))
    key = (H, W, 1, 1, 1, 1, 1, 1, False, 0)
    line = f"{C:3d}->{C:<3d} {H}x{W}:"
    for npb in (1, 2, 4):
        for mb in (1, 2):
            for ck in (16, 32):
                pc.tuned[key] = (npb, 4, ck, mb, 1)
                try:
                    t = timeit(lambda: ops.conv2d(x, pc, pad=1, act='relu'))
                    line += f" n{npb}m{mb}k{ck}:{t:5.1f}"
                except Exception:
                    line += f" n{npb}m{mb}k{ck}:  -  "
    print(line, flush=True)
