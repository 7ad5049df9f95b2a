"""dev (GPU box): which intermediate of frame F carries a product-vs-oracle deviation?

Runs the product path (eager, default precision) over frames 0..F of a LONG case's video, once on the exact inputs and
once with 1e-7 relative input noise, and the CPU oracle on the exact inputs (host cores of the box), and prints for frame
F, per intermediate (Ts = the up-sampled SE3 field, pred_curr = stereo output, pred_warp = the warped previous fused
disparity, fusion / reset weights, pred_disp): mean |delta|, fraction > 0.25, bounding box of the deviating pixels.

    LONG=cfg3_50 F=1 python tools/frame_event_vs_oracle.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from oracle import codd as oc

DEV = "cuda:0"
LONG = os.environ.get("LONG", "cfg3_50")
F = int(os.environ.get("F", "1"))
case = T.LONG_CASES[LONG]
H, W, intr, img_shape, _, _ = T.CASES[case[0]]
img, r_img, _ = synth.stereo_sequence(H, W, F + 1, **({"flow": case[3]} if len(case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
KEYS = ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp", "Ts", "weight")


def product(noise=0.0, precision="split"):
    est = T._build(False, case[1])[0].to(DEV)
    g = torch.Generator().manual_seed(1)
    prev = ops.set_conv_precision(precision)
    ops.enable_autotune(True, shipped=True)
    try:
        state, out = {}, None
        for f in range(F + 1):
            l, r = img[:, f].clone(), r_img[:, f].clone()
            if noise:
                l = l * (1 + noise * torch.randn(l.shape, generator=g))
                r = r * (1 + noise * torch.randn(r.shape, generator=g))
            out = est.consistent_online_depth_estimation(l.to(DEV).contiguous(), r.to(DEV).contiguous(), metas[0], state)
        res = {k: out[k].detach().float().cpu() for k in KEYS if k in out and torch.is_tensor(out[k])}
        res["memory_disp_warp"] = state["memory"][3].detach().float().cpu() if len(state["memory"]) > 3 else None
        return res
    finally:
        ops.enable_autotune(False)
        ops.set_conv_precision(prev)


def oracle():
    sd = T._build(False, case[1])[1]
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    state, o = {}, None
    t0 = time.time()
    with torch.no_grad():
        for f in range(F + 1):
            o = oc.frame(sd, img[:, f], r_img[:, f], state, intr, iters=case[1])
    print(f"oracle: {F + 1} frames in {time.time() - t0:.0f} s on {torch.get_num_threads()} threads", flush=True)
    return {k: o[k].detach().float() for k in KEYS if k in o and torch.is_tensor(o[k])}


def report(tag, a, b):
    for k in KEYS:
        if k not in a or k not in b or a[k] is None or b[k] is None:
            continue
        x, y = a[k].reshape(-1, *a[k].shape[-2:]) if k != "Ts" else a[k].reshape(H, W, -1).permute(2, 0, 1), None
        y = b[k].reshape(-1, *b[k].shape[-2:]) if k != "Ts" else b[k].reshape(H, W, -1).permute(2, 0, 1)
        if x.shape != y.shape:
            print(f"  {tag} {k}: shapes {tuple(x.shape)} vs {tuple(y.shape)}")
            continue
        d = (x - y).abs().amax(0)
        thr = 0.25 if k.startswith("pred") else 1e-3
        bad = d > thr
        ys, xs = torch.nonzero(bad, as_tuple=True)
        box = f"rows {ys.min().item()}..{ys.max().item()} cols {xs.min().item()}..{xs.max().item()}" if len(ys) else "-"
        print(f"  {tag} {k:15s} mean |d| {d.mean().item():.3e}  max {d.max().item():.3e}  > {thr}: {bad.float().mean().item():.3e} ({int(bad.sum())} px)  {box}")


p0 = product()
p1 = product(1e-7)
pf = product(precision="fp32")
o = oracle()
print(f"{LONG} frame {F}:")
report("product        vs oracle", p0, o)
report("product+1e-7   vs oracle", p1, o)
report("product(fp32)  vs oracle", pf, o)
report("product vs product+1e-7 ", p0, p1)
