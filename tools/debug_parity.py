import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import configs, synth, ops
from codd_amd.registry import build_estimator
from oracle import codd as oc
H, W, iters, MF = 128, 384, 4, 4
est = build_estimator(configs.codd(iters=iters)).eval(); synth.load_synthetic_weights(est, 1.4)
sd = {k: v.clone() for k, v in est.state_dict().items()}
est = est.cuda()
img, r_img, _ = synth.stereo_sequence(H, W, MF)
intr = (240.0, 240.0, 190.0, 62.0)
metas = synth.default_metas(H, W, intrinsics=intr)[0]
so, sg = {}, {}
def st(a, b, name):
    a = a.detach().cpu().float(); b = b.detach().cpu().float()
    d = (a - b).abs()
    print(f"    {name:16s} shape {tuple(a.shape)} mean|d| {d.mean().item():.3e} max {d.max().item():.3e} frac>1e-2 {(d > 1e-2).float().mean().item():.3e} |ref| {b.abs().mean().item():.3g}")
with torch.no_grad():
    for f in range(MF):
        print("frame", f)
        trace = []
        oo = oc.frame(sd, img[:, f], r_img[:, f], so, intr, iters=iters, trace=trace)
        og = est.consistent_online_depth_estimation(img[:, f].cuda().contiguous(), r_img[:, f].cuda().contiguous(), metas, sg)
        for k in ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp", "left_feat", "weight", "Ts"):
            if k in oo and k in og:
                st(og[k], oo[k], k)
        if trace:
            print("    oracle T after iters: |tau| mean", [round(t["T"][..., :3].abs().mean().item(), 4) for t in trace])
