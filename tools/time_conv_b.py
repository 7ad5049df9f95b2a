"""Dev tool: correctness + timing of the split-bf16 conv kernel on the heavy layer shapes, every candidate
configuration the library accepts, beside the tuned fp32 kernels.  python tools/time_conv_b.py [terms]"""
import os
import sys
import ctypes as C

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import _abi, ops  # noqa: E402

DEV = "cuda:0"
LAYERS = [  # cin, cout, k, stride, pad, dil, H, W
    (256, 256, 3, 1, 1, 1, 72, 120), (128, 256, 3, 1, 1, 1, 72, 120), (128, 256, 3, 1, 4, 4, 72, 120),
    (196, 256, 3, 1, 1, 1, 72, 120), (128, 768, 3, 1, 1, 1, 72, 120), (128, 128, 3, 1, 1, 1, 72, 120),
    (256, 384, 1, 1, 0, 1, 72, 120), (9, 128, 7, 1, 3, 1, 72, 120),
    (64, 64, 3, 1, 1, 1, 288, 480), (96, 96, 3, 1, 1, 1, 144, 240), (3, 64, 7, 2, 3, 1, 576, 960),
    (16, 16, 3, 1, 1, 1, 576, 960), (32, 32, 3, 1, 1, 1, 288, 480), (32, 32, 3, 1, 1, 1, 144, 240),
    (64, 30, 7, 1, 3, 1, 144, 240), (24, 24, 3, 1, 1, 1, 144, 240), (32, 32, 3, 1, 1, 1, 36, 60),
]


if os.environ.get("LAYER"):  # dev: one extra layer "cin,cout,k,stride,pad,dil,H,W" (becomes index 0)
    LAYERS.insert(0, tuple(int(v) for v in os.environ["LAYER"].split(",")))


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best * 1e3


def main():
    terms = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    lib = _abi.load()
    for li, (cin, cout, k, s, p, d, H, W) in enumerate(LAYERS):
        if only >= 0 and li != only:
            continue
        g = torch.Generator().manual_seed(li)
        x = torch.randn(1, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
        scale = F.conv2d(x.abs().double(), w.abs().double(), None, s, p, d).max().item()
        xd = x.to(DEV)
        pc = ops.PackedConv(w.to(DEV), b.to(DEV))
        gflop = 2.0 * cin * cout * k * k * ref.shape[2] * ref.shape[3] / 1e9
        ops.set_conv_precision("fp32")
        ops.enable_autotune(True, shipped=True)
        o32 = ops.conv2d(xd, pc, stride=s, pad=p, dil=d)
        t32 = timeit(lambda: ops.conv2d(xd, pc, stride=s, pad=p, dil=d))
        ops.enable_autotune(False)
        e32 = (o32.cpu().double() - ref).abs().max().item() / scale
        print(f"[{li}] {cin}->{cout} k{k} s{s} d{d} {H}x{W}: {gflop:.2f} GFLOP | fp32 tuned {t32:7.1f} us {gflop / t32 * 1e3:6.1f} TF err {e32:.1e}")
        ops.set_conv_precision("split" if terms == 3 else "bf16")
        Ho, Wo = ref.shape[2:]
        cands = ops._bf16_candidates(pc, Ho, Wo, 1, k * k, terms)
        out = torch.empty(1, cout, Ho, Wo, device=DEV)
        res = []
        for c in cands:
            pc.tuned.clear()
            key = (Ho, Wo, 1, s, s, d, d, p, False, terms)
            pp = _abi.ConvParams()
            pp.C0, pp.C1, pp.B, pp.Hin, pp.Win = cin, 0, 1, H, W
            pp.Cout, pp.Hout, pp.Wout = cout, Ho, Wo
            pp.kh, pp.kw, pp.sy, pp.sx, pp.pad_t, pp.pad_l, pp.dil_y, pp.dil_x = k, k, s, s, p, p, d, d
            pp.terms = terms
            if not ops._cfg_ok(lib, pp, c):
                continue
            pc.tuned[key] = c
            out.fill_(float("nan"))
            ops.conv2d(xd, pc, stride=s, pad=p, dil=d, out=out)
            err = (out.cpu().double() - ref).abs().max().item() / scale
            t = timeit(lambda: ops.conv2d(xd, pc, stride=s, pad=p, dil=d, out=out))
            res.append((t, c, err))
        res.sort(key=lambda r: r[0])
        for t, c, err in res[:6] + [r for r in res[6:] if r[1][5] == 8]:
            print(f"      {t:7.1f} us {gflop / t * 1e3:6.1f} TF  cfg(xb,th,ck,mb,_,pgw,cgw,terms,ks)={c} err {err:.1e}")
        bad = [r for r in res if not r[2] < (1e-4 if terms == 3 else 2e-2)]
        if bad:
            print("      !!! WRONG:", [(c[:7], e) for _, c, e in bad][:5])
        pc._packs.clear()


if __name__ == "__main__":
    main()
