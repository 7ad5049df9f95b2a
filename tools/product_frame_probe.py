"""dev (GPU box): conditioning of ONE frame measured on the PRODUCT path: frames 0..F-1 of a LONG case's video on the exact
inputs, then frame F from the same recurrent state on the exact images and on K copies with NOISE relative noise.
Prints, per intermediate, how far the noisy evaluations move from the exact one, and where the exact one stands against
the tracked golden (sub-grid).     LONG=cfg3_50 F=20 K=3 NOISE=1e-7 python tools/product_frame_probe.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth

DEV = "cuda:0"
LONG, F, K, NOISE = os.environ.get("LONG", "cfg3_50"), int(os.environ.get("F", "20")), int(os.environ.get("K", "3")), float(os.environ.get("NOISE", "1e-7"))
case = T.LONG_CASES[LONG]
H, W, intr, img_shape, _, _ = T.CASES[case[0]]
z, sub = T._long_golden(LONG)
img, r_img, _ = synth.stereo_sequence(H, W, F + 1, **({"flow": case[3]} if len(case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
KEYS = ("pred_curr", "pred_warp", "fusion_weights", "reset_weights", "pred_disp")


def clone_state(st):
    def c(v):
        if torch.is_tensor(v): return v.clone()
        if isinstance(v, (list, tuple)): return type(v)(c(x) for x in v)
        if isinstance(v, dict): return {k: c(x) for k, x in v.items()}
        return copy.copy(v)
    return c(st)


est = T._build(False, case[1])[0].to(DEV)
ops.enable_autotune(True, shipped=True)
state = {}
for f in range(F):
    est.consistent_online_depth_estimation(img[:, f].to(DEV).contiguous(), r_img[:, f].to(DEV).contiguous(), metas[0], state)
torch.cuda.synchronize()


def frame(l, r):
    st = clone_state(state)
    out = est.consistent_online_depth_estimation(l.to(DEV).contiguous(), r.to(DEV).contiguous(), metas[0], st)
    return {k: out[k].detach().float().cpu().reshape(H, W) for k in KEYS if k in out}


base = frame(img[:, F], r_img[:, F])
g = torch.from_numpy(z[f"{LONG}_f{F}"])
d = (base["pred_disp"][::sub, ::sub] - g).abs()
ys, xs = torch.nonzero(d > 0.25, as_tuple=True)
print(f"{LONG} frame {F}: product vs tracked oracle (sub-grid): mean {d.mean():.2e}  flipped {(d > 0.25).float().mean():.2e} "
      f"({len(ys)} px: " + ", ".join(f"({int(y) * sub},{int(x) * sub}) {d[y, x]:.1f}px" for y, x in list(zip(ys, xs))[:12]) + ")")
for k in range(K):
    gen = torch.Generator().manual_seed(1000 * F + k)
    l = img[:, F] * (1 + NOISE * torch.randn(img[:, F].shape, generator=gen))
    r = r_img[:, F] * (1 + NOISE * torch.randn(img[:, F].shape, generator=gen))
    o = frame(l, r)
    for key in KEYS:
        if key in o:
            dd = (o[key] - base[key]).abs()
            thr = 0.25 if key.startswith("pred") else 1e-3
            print(f"  seed {k} {key:15s} product under {NOISE:g} noise moves: mean {dd.mean():.2e}  > {thr}: {(dd > thr).float().mean():.2e} ({int((dd > thr).sum())} px)  max {dd.max():.2e}")
    dn = (o["pred_disp"][::sub, ::sub] - g).abs()
    print(f"  seed {k} noisy product vs tracked oracle (sub-grid): mean {dn.mean():.2e}  flipped {(dn > 0.25).float().mean():.2e}")
