"""dev (GPU box): full-resolution intermediates of frames F0..F of the product path (eager) -> gpurun_out/<OUT>.npz (fp32),
for an offline comparison with the oracle evaluated from its saved state (tools/oracle_frame_compare.py).
    LONG=cfg3_50 F0=19 F=20 OUT=prod_f20 python tools/product_frame_dump.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth

DEV = "cuda:0"
LONG, F0, F = os.environ.get("LONG", "cfg3_50"), int(os.environ.get("F0", "19")), int(os.environ.get("F", "20"))
case = T.LONG_CASES[LONG]
H, W, intr, img_shape, _, _ = T.CASES[case[0]]
img, r_img, _ = synth.stereo_sequence(H, W, F + 1, **({"flow": case[3]} if len(case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
KEYS = ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp", "weight")
est = T._build(False, case[1])[0].to(DEV)
ops.enable_autotune(True, shipped=True)
state, arrays = {}, {}
for f in range(F + 1):
    out = est.consistent_online_depth_estimation(img[:, f].to(DEV).contiguous(), r_img[:, f].to(DEV).contiguous(), metas[0], state)
    if f >= F0:
        for k in KEYS:
            if k in out and torch.is_tensor(out[k]):
                arrays[f"{k}_f{f}"] = out[k].detach().float().cpu().numpy().reshape(-1, H, W).astype(np.float32)
        arrays[f"Ts_f{f}"] = out["Ts"].detach().float().cpu().numpy().reshape(H, W, 7) if "Ts" in out else np.zeros(1, np.float32)
path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("OUT", f"prod_f{F}") + ".npz")
os.makedirs(os.path.dirname(path), exist_ok=True)
np.savez_compressed(path, **arrays)
print("wrote", path, os.path.getsize(path))
