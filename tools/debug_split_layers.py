"""Dev tool: run one stereo (or full CODD) frame with every conv executed by BOTH the exact-fp32 kernels and the
split-bf16 kernel on identical inputs; report the layers whose outputs differ by more than the split-bf16 bound."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import configs, ops, synth  # noqa: E402
from codd_amd.registry import build_estimator  # noqa: E402

DEV = "cuda:0"
orig = ops.conv2d
worst = []


def both(x, pc, x2=None, stride=1, pad=0, dil=1, act="none", res1=None, res2=None, post=None, out=None, pad_tl=None,
         out_hw=None):
    def clone(t):
        if t is None:
            return None
        if isinstance(t, ops.Slice):
            return ops.Slice(t.buf.clone(), t.coff, t.c)
        return t.clone()
    # fp32 run on cloned operands (in-place residual aliasing safe)
    prev = ops.set_conv_precision("fp32")
    o32 = orig(clone(x), pc, x2=clone(x2), stride=stride, pad=pad, dil=dil, act=act, res1=clone(res1), res2=clone(res2),
               post=clone(post), out=None, pad_tl=pad_tl, out_hw=out_hw)
    ops.set_conv_precision("split")
    o = orig(x, pc, x2=x2, stride=stride, pad=pad, dil=dil, act=act, res1=res1, res2=res2, post=post, out=out,
             pad_tl=pad_tl, out_hw=out_hw)
    ops.set_conv_precision(prev)
    got = o.tensor() if isinstance(o, ops.Slice) else o
    d = (got - o32).abs().max().item()
    sc = o32.abs().max().item() + 1e-9
    xs = ops._as_slice(x)
    worst.append((d / sc, "%dx%d %d->%d s%s p%s d%s act=%s in %s B%d x2=%s res=%s out=%s" % (
        pc.kh, pc.kw, pc.cin, pc.cout, stride, pad if pad_tl is None else pad_tl, dil, act, tuple(xs.shape[2:]), xs.shape[0],
        x2 is not None, res1 is not None, "slice" if isinstance(out, ops.Slice) else type(out).__name__), d, sc))
    return o


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 384)
    full = len(sys.argv) > 3
    est = build_estimator(configs.codd(iters=2) if full else configs.stereo_only()).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    est = est.to(DEV)
    img, r_img, _ = synth.stereo_sequence(H, W, 2)
    metas = synth.default_metas(H, W, intrinsics=(240.0, 240.0, W / 2.0, H / 2.0))
    ops.conv2d = both
    import codd_amd.stereo as st, codd_amd.motion as mo, codd_amd.fusion as fu, codd_amd.hrnet as hr
    state = {}
    with torch.no_grad():
        for f in range(2 if full else 1):
            est.consistent_online_depth_estimation(img[:, f].to(DEV).contiguous(), r_img[:, f].to(DEV).contiguous(), metas[0], state)
    worst.sort(key=lambda r: -r[0])
    print("layers:", len(worst))
    for r in worst[:25]:
        print("rel %.2e  abs %.2e scale %.2e  %s" % (r[0], r[2], r[3], r[1]))


if __name__ == "__main__":
    main()
