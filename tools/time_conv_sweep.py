import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
dev="cuda"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
cin, cout, k = 128, 256, 3
wt = torch.randn(cout, cin, k, k, device=dev) / 34.0
pc = ops.PackedConv(wt, torch.zeros(cout, device=dev))
for (H, W) in [(72, 120), (72, 112), (64, 112), (72, 64), (36, 64), (144, 240)]:
    x = torch.randn(1, cin, H, W, device=dev)
    t = timeit(lambda: ops.conv2d(x, pc, pad=1, act="relu"))
    nb = -(-H//4) * -(-W//16) * 4
    print(f"{H}x{W}: blocks {nb:5d}  {t:7.1f} us  {2.0*cin*cout*9*H*W/t/1e6:6.1f} TF")
