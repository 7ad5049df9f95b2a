import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from codd_amd import ops
f1 = torch.randn(1, 128, 72, 120, device="cuda"); f2 = torch.randn(1, 128, 72, 120, device="cuda")
for _ in range(3): p = ops.allpairs_corr(f1, f2)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): p = ops.allpairs_corr(f1, f2)
e.record(); torch.cuda.synchronize()
print(os.environ.get("CODD_CORR_NPB"), os.environ.get("CODD_CORR_NW"), "allpairs %.1f us" % (s.elapsed_time(e) * 100))
