"""dev: how far do two runs of the SAME recurrence drift apart when they differ by fp32 rounding only?

Runs the benchmarked configuration (full CODD 960x576, iters 16) for N frames under several arithmetic variants that
are all "exact to fp32 rounding" of one another, and prints the per-frame mean |delta| / flipped fraction (> 0.25 px)
of every variant against the default product path, and of every variant against the tracked CPU-oracle frames
(tests/golden/headline_oracle_long_sub4.npz).  If product-vs-product drifts like product-vs-oracle, the drift is a
property of the recurrence (discontinuous selections fed back through state["memory"]), not of an implementation.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner

DEV = "cuda:0"
N = int(os.environ.get("N", "16"))
name = os.environ.get("CASE", "cfg3_codd_960x576")
# LONG=<case of T.LONG_CASES> (e.g. cfg3_50: the second synthetic video + its own golden file) overrides CASE
PREFIX = os.environ.get("LONG") or {"cfg3_codd_960x576": "cfg3_long", "cfg5_tartanair_640x512": "cfg5_long"}[name]
_case = T.LONG_CASES[PREFIX]
name = _case[0]
H, W, intr, img_shape, _, _ = T.CASES[name]
z, sub = T._long_golden(PREFIX)
N = min(N, T.n_frames(z, PREFIX))
img, r_img, _ = synth.stereo_sequence(H, W, N, **({"flow": _case[3]} if len(_case) > 3 else {}))
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)


def _fp32(fn):
    def w(*a, **k):
        with ops.stage("stereo"):  # the stage name whose policy is exact fp32
            return fn(*a, **k)
    return w


def run(tag, roll=True, precision="split", noise=0.0, graph=True, exact=()):
    est = T._build(False, 16)[0].to(DEV)
    r3 = est.motion.raft3d
    for what in exact:  # stages forced onto the exact-fp32 kernels
        if what == "fnet": r3.fnet.forward = _fp32(r3.fnet.forward)
        elif what == "update": r3.update_block.run = _fp32(r3.update_block.run)
        elif what == "fusion": est.fusion.memory_query = _fp32(est.fusion.memory_query)
        elif what == "motion": est.motion.forward = _fp32(est.motion.forward)
    prev_roll, prev_p, prev_ap = ops.USE_ROLL, ops.set_conv_precision(precision), ops.ALLPAIRS_SPLIT
    ops.USE_ROLL = roll
    ops.ALLPAIRS_SPLIT = prev_ap and "allpairs" not in exact
    ops.enable_autotune(True, shipped=True)
    out = []
    try:
        runner = FrameRunner(est, metas[0], use_graph=graph)
        g = torch.Generator().manual_seed(1)
        for f in range(N):
            l, r = img[:, f].clone(), r_img[:, f].clone()
            if noise:
                l = l * (1 + noise * torch.randn(l.shape, generator=g))
                r = r * (1 + noise * torch.randn(r.shape, generator=g))
            out.append(runner.step(l.to(DEV).contiguous(), r.to(DEV).contiguous()).cpu()[0, 0])
    finally:
        ops.enable_autotune(False)
        ops.USE_ROLL = prev_roll
        ops.ALLPAIRS_SPLIT = prev_ap
        ops.set_conv_precision(prev_p)
    return out


def stats(a, b):
    d = (a - b).abs()
    return d.mean().item(), (d > 0.25).float().mean().item()


def line(tag, xs, ys):
    ms, fs = zip(*[stats(x, y) for x, y in zip(xs, ys)])
    print(f"{tag:34s} mean  " + " ".join(f"{m:.1e}" for m in ms))
    print(f"{'':34s} flip  " + " ".join(f"{f:.1e}" for f in fs), flush=True)


ALL = dict(default=dict(), no_roll=dict(roll=False), fp32_convs=dict(precision="fp32"), split16=dict(precision="split16"),
           input_noise_1e_7=dict(noise=1e-7), eager=dict(graph=False), fnet_exact=dict(exact=("fnet",)),
           update_exact=dict(exact=("update",)), fusion_exact=dict(exact=("fusion",)), motion_exact=dict(exact=("motion",)),
           fnet_fusion_exact=dict(exact=("fnet", "fusion")), allpairs_exact=dict(exact=("allpairs",)),
           fnet_allpairs_exact=dict(exact=("fnet", "allpairs")))
want = [a for a in sys.argv[1:] if a in ALL] or ["default", "no_roll", "fp32_convs", "input_noise_1e_7", "eager"]
if "default" not in want:
    want = ["default"] + want
variants = {k: ALL[k] for k in want}
res = {k: run(k, **v) for k, v in variants.items()}
gold = [torch.from_numpy(z[f"{PREFIX}_f{f}"]) for f in range(N)]
print(f"frames 0..{N - 1}; sub-grid 1/{sub} for the oracle rows, every pixel for the product-vs-product rows")
for k in res:
    line(f"{k} vs ORACLE (tracked)", [x[::sub, ::sub] for x in res[k]], gold)
for k in res:
    if k != "default":
        line(f"{k} vs default (product)", res[k], res["default"])
