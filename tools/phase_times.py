"""dev: per-phase GPU time of one steady-state frame (eager, CUDA events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, synth, ops
from codd_amd.registry import build_estimator
H, W = 576, 960
est = build_estimator(configs.codd()).eval(); synth.load_synthetic_weights(est, 1.4); est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, 6); img, r_img = img.cuda(), r_img.cuda()
metas = synth.default_metas(H, W, img_shape=(540, 960, 3))[0]
state = {}
ev = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, e))
r3 = est.motion.raft3d
orig_fnet, orig_ctx, orig_ub, orig_gn, orig_ap = r3.fnet.forward, r3.context, r3.update_block.run, ops.se3_gn_step, ops.allpairs_corr
acc = {}
def timed(name, fn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = fn(*a, **k); e.record(); acc.setdefault(name, []).append((s, e)); return out
    return w
r3.fnet.forward = timed("fnet", orig_fnet); r3.context = timed("cnet", orig_ctx)
r3.update_block.run = timed("update_block", orig_ub); ops.se3_gn_step = timed("gn", orig_gn); ops.allpairs_corr = timed("allpairs", orig_ap)
ops.corr_lookup = timed("lookup", ops.corr_lookup); ops.splat = timed("splat", ops.splat); ops.cvx_upsample = timed("cvx", ops.cvx_upsample)
est.stereo.stereo_matching = timed("stereo", est.stereo.stereo_matching)
est.fusion.memory_query = timed("fusion", est.fusion.memory_query)
for i in range(4):
    acc.clear()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    est.consistent_online_depth_estimation(img[:, i].contiguous(), r_img[:, i].contiguous(), metas, state)
    e.record(); torch.cuda.synchronize()
    tot = s.elapsed_time(e)
print("frame total ms", round(tot, 2))
for k, v in acc.items():
    print(f"  {k:14s} calls {len(v):3d}  {sum(a.elapsed_time(b) for a, b in v):7.2f} ms")
