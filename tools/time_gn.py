"""dev: time codd_se3_gn_step (prep + builder + solve) at the update loop's shape (72x120, radius 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
dev = "cuda:0"
B, h, w = 1, 72, 120
g = torch.Generator().manual_seed(0)
T = torch.zeros(B, h, w, 7); T[..., 6] = 1; T[..., :3] = torch.randn(B, h, w, 3, generator=g) * 0.01
d1 = torch.rand(B, h, w, generator=g) * 30 + 3
K8 = [131.25, 131.25, 60.0, 33.75]
ae = torch.randn(B, 32, h, w, generator=g) * 2
xyz = torch.rand(B, h, w, 3, generator=g) * 50
delta = torch.randn(B, 3, h, w, generator=g) * 0.1
weight = torch.sigmoid(torch.randn(B, 3, h, w, generator=g))
args = [t.to(dev).contiguous() for t in (ae, xyz, delta, weight, d1)]
Tg = T.to(dev)
for _ in range(5):
    ops.se3_gn_step(Tg.clone(), *args, K8, radius=32)
torch.cuda.synchronize()
Ts = [Tg.clone() for _ in range(40)]
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for t in Ts:
    ops.se3_gn_step(t, *args, K8, radius=32)
e.record(); torch.cuda.synchronize()
print(f"{s.elapsed_time(e) / 40 * 1e3:.1f} us per GN step (prep + build + solve)")
