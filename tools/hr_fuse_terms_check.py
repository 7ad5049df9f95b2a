"""dev: HRNet with CODD_HR_FUSE_TERMS on / off -- first HRModule whose outputs differ, and by how much."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_headline_parity as T
from codd_amd import synth, hrnet, ops
est = T._build(False, 16)[0].to("cuda:0")
img, _, _ = synth.stereo_sequence(512, 640, 1)
x = img[:, 0].to("cuda:0")
r3 = est.motion.raft3d
FLAG = os.environ.get("FLAG", "FUSE_TERMS")
logs = {False: [], True: []}
orig = hrnet.HRModule.run if hasattr(hrnet.HRModule, "run") else hrnet.HRModule.forward
name = "run" if hasattr(hrnet.HRModule, "run") else "forward"
def hooked(self, *a, **k):
    out = orig(self, *a, **k)
    logs[getattr(hrnet, FLAG)].append([o.clone() for o in out])
    return out
setattr(hrnet.HRModule, name, hooked)
outs = []
for flag in (False, True):
    setattr(hrnet, FLAG, flag)
    with ops.stage("context"):
        outs.append(r3.context(x).clone())
for m, (a, b) in enumerate(zip(logs[False], logs[True])):
    for i, (p, q) in enumerate(zip(a, b)):
        print(f"module {m} branch {i} {tuple(p.shape)}: equal {torch.equal(p, q)} max |d| {(p - q).abs().max().item():.3e} scale {p.abs().max().item():.2e}")
print("network output bit-identical:", torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item())
