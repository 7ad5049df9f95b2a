"""dev (GPU box): ONE frame, product against oracle, both started from the ORACLE's saved state behind frame F - 1 (teacher
forcing: tests/_oracle_cache/snap/oracle_state_f<F-1>.pt), with the motion stage's outputs captured in front of Fusion:
which tensor first carries a difference inside the region of interest?
    F=19 ROI=0,64,400,560 PRECISION=split|fp32 python tools/frame_bisect.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from oracle import codd as oc, fusion as ofusion

DEV = "cuda:0"
LONG, F = os.environ.get("LONG", "cfg3_50"), int(os.environ.get("F", "19"))
y0, y1, x0, x1 = [int(v) for v in os.environ.get("ROI", "0,64,400,560").split(",")]
case = T.LONG_CASES[LONG]
H, W, intr, img_shape, _, _ = T.CASES[case[0]]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
snap = os.path.join(root, "tests", "_oracle_cache", "snap", f"oracle_state_f{F - 1}.pt")
img, r_img, _ = synth.stereo_sequence(H, W, F + 1, **({"flow": case[3]} if len(case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
NAMES = ("img_warp", "feat_warp", "conf_warp", "disp_warp", "flow_warp")

# ---- oracle
torch.set_num_threads(32)
est, sd = T._build(False, case[1])
cap_o = {}
orig = ofusion.memory_query
def hook_o(sd_, outputs, state, *a, **k):
    cap_o.update({n: t.detach().float().clone() for n, t in zip(NAMES, state["memory"])})
    cap_o["Ts"] = outputs["Ts"].detach().float().clone()
    return orig(sd_, outputs, state, *a, **k)
ofusion.memory_query = hook_o
oc.ofusion = ofusion
st = torch.load(snap, map_location="cpu")["state"]
t0 = time.time()
with torch.no_grad():
    o = oc.frame(sd, img[:, F], r_img[:, F], st, intr, iters=case[1])
print(f"oracle frame {F}: {time.time() - t0:.0f} s", flush=True)
for k in ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp"):
    cap_o[k] = o[k].detach().float()

# ---- product, teacher-forced, per precision
for prec in os.environ.get("PRECISION", "split,fp32").split(","):
    prev = ops.set_conv_precision(prec)
    ops.enable_autotune(True, shipped=True)
    e = T._build(False, case[1])[0].to(DEV)
    o_st = torch.load(snap, map_location="cpu")["state"]
    state = dict(raft_feat=o_st["raft_feat"].to(DEV), raft_netinp=o_st["raft_netinp"].to(DEV), memory=[m.to(DEV) for m in o_st["memory"]])
    cap_p = {}
    fq = e.fusion.memory_query
    def hook_p(outputs, state_, *a, **k):
        cap_p.update({n: t.detach().float().cpu().clone() for n, t in zip(NAMES, state_["memory"])})
        cap_p["Ts"] = outputs["Ts"].detach().float().cpu().clone()
        return fq(outputs, state_, *a, **k)
    e.fusion.memory_query = hook_p
    out = e.consistent_online_depth_estimation(img[:, F].to(DEV).contiguous(), r_img[:, F].to(DEV).contiguous(), metas[0], state)
    for k in ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp"):
        cap_p[k] = out[k].detach().float().cpu()
    for k in ("Ts",) + NAMES + ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp"):
        a, b = cap_p[k], cap_o[k]
        a = a.reshape(b.shape)
        d = (a - b).abs()
        if k == "Ts":
            d = d.reshape(H, W, -1).amax(-1)
        else:
            d = d.reshape(-1, d.shape[-2], d.shape[-1]).amax(0)
        s = H // d.shape[-2]
        r = d[y0 // s:y1 // s, x0 // s:x1 // s]
        thr = 0.25 if k in ("pred_curr", "pred_warp", "pred_disp", "disp_warp") else 1e-2
        print(f"[{prec}, teacher-forced frame {F}] {k:15s} mean |d| {d.mean():.2e} max {d.max():.2e}  > {thr}: {int((d > thr).sum())} px | ROI: mean {r.mean():.2e} max {r.max():.2e}  > {thr}: {int((r > thr).sum())} px", flush=True)
    ops.set_conv_precision(prev)
