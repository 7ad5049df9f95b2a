import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
cin, cout, k, H, W, dil = [int(v) for v in sys.argv[1:7]]
x = torch.randn(1, cin, H, W, device="cuda"); wt = torch.randn(cout, cin, k, k, device="cuda") / (cin*k*k)**0.5
pc = ops.PackedConv(wt, torch.zeros(cout, device="cuda"))
for _ in range(5):
    ops.conv2d(x, pc, pad=dil*(k//2), dil=dil, act="relu")
torch.cuda.synchronize()
