"""dev (GPU box): the product's recurrent state behind every frame against the oracle's saved state (compact snapshots under
tests/_oracle_cache/snap/oracle_state_f<t>.pt): per tensor, mean / max |delta| over the whole map and inside a region of
interest.   LONG=cfg3_50 ROI=0,64,400,560 python tools/product_state_vs_oracle.py"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth

DEV = "cuda:0"
LONG = os.environ.get("LONG", "cfg3_50")
y0, y1, x0, x1 = [int(v) for v in os.environ.get("ROI", "0,64,400,560").split(",")]
case = T.LONG_CASES[LONG]
H, W, intr, img_shape, _, _ = T.CASES[case[0]]
snaps = {int(os.path.basename(p)[len("oracle_state_f"):-3]): p for p in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_oracle_cache", "snap", "oracle_state_f*.pt"))}
F = max(snaps)
img, r_img, _ = synth.stereo_sequence(H, W, F + 1, **({"flow": case[3]} if len(case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
prec = os.environ.get("PRECISION", "split")
prev = ops.set_conv_precision(prec)
est = T._build(False, case[1])[0].to(DEV)
ops.enable_autotune(True, shipped=True)
state = {}
TEACHER = os.environ.get("TEACHER", "")  # frame index: replace the product's state by the oracle's behind that frame
for f in range(F + 1):
    est.consistent_online_depth_estimation(img[:, f].to(DEV).contiguous(), r_img[:, f].to(DEV).contiguous(), metas[0], state)
    if f in snaps:
        o = torch.load(snaps[f], map_location="cpu")["state"]
        items = [("raft_feat", state["raft_feat"], o["raft_feat"], 8), ("raft_netinp", state["raft_netinp"], o["raft_netinp"], 8),
                 ("memory.feat", state["memory"][1], o["memory"][1], 4), ("memory.disp", state["memory"][2], o["memory"][2], 1)]
        for name, a, b, s in items:
            a = a.detach().float().cpu().reshape(b.shape)
            d = (a - b).abs()
            r = d[..., y0 // s:y1 // s, x0 // s:x1 // s]
            print(f"[{prec}] behind frame {f:2d} {name:12s} mean |d| {d.mean():.2e} max {d.max():.2e} | ROI rows {y0}..{y1} cols {x0}..{x1}: mean {r.mean():.2e} max {r.max():.2e}  (scale {b.abs().mean():.2e})", flush=True)
        if TEACHER and int(TEACHER) == f:
            keys = os.environ.get("TEACHER_KEYS", "raft_feat,raft_netinp,memory").split(",")
            if "raft_feat" in keys: state["raft_feat"] = o["raft_feat"].to(DEV)
            if "raft_netinp" in keys: state["raft_netinp"] = o["raft_netinp"].to(DEV)
            if "memory" in keys: state["memory"] = [m.to(DEV) for m in o["memory"]]
            for i_ in range(3):
                if f"memory{i_}" in keys: state["memory"][i_] = o["memory"][i_].to(DEV)
            print(f"  -> product state {keys} REPLACED by the oracle's behind frame {f}")
ops.set_conv_precision(prev)
