"""Dev tool: the update-block layers of the split-bf16 kernel, input pre-split (as on the hot path), every candidate
configuration incl. the eight-consumer-wave ones; prints the best of each family.  python tools/time_conv_ks.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import _abi, ops  # noqa: E402

DEV = "cuda:0"
LAYERS = [  # cin, cout, k, pad, dil, H, W, xs_out
    (128, 256, 3, 1, 1, 72, 120, False), (128, 256, 3, 4, 4, 72, 120, False), (128, 128, 3, 1, 1, 72, 120, False),
    (256, 256, 3, 1, 1, 72, 120, True), (196, 256, 3, 1, 1, 72, 120, True), (128, 768, 3, 1, 1, 72, 120, True),
    (256, 384, 1, 0, 1, 72, 120, False), (128, 384, 1, 0, 1, 72, 120, False), (9, 128, 7, 3, 1, 72, 120, True),
    (64, 64, 3, 1, 1, 288, 480, False), (96, 96, 3, 1, 1, 144, 240, False), (128, 128, 3, 1, 1, 72, 120, False),
    (384, 384, 1, 0, 1, 72, 120, False),  # [12] the merged encoder heads
]


def timeit(fn, n=10):
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
    only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    lib = _abi.load()
    ops.set_conv_precision("split")
    for li, (cin, cout, k, p, d, H, W, xso) in enumerate(LAYERS):
        if only >= 0 and li != only:
            continue
        g = torch.Generator().manual_seed(li)
        x = torch.randn(1, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(x, w, b, 1, p, d)
        scale = F.conv2d(x.abs(), w.abs(), None, 1, p, d).max().item()
        pc = ops.PackedConv(w.to(DEV), b.to(DEV))
        gflop = 2.0 * cin * cout * k * k * H * W / 1e9
        xs = ops.split_input(x.to(DEV), border=p)
        so = ops.split_buffer(("ks", li), 1, cout, H, W, 1, DEV) if xso else None
        key = (H, W, 1, 1, 1, d, d, p, False, 3, "split")
        res = []
        pp = _abi.ConvParams()
        pp.C0, pp.C1, pp.B, pp.Hin, pp.Win = cin, 0, 1, H, W
        pp.Cout, pp.Hout, pp.Wout = cout, H, W
        pp.kh, pp.kw, pp.sy, pp.sx, pp.pad_t, pp.pad_l, pp.dil_y, pp.dil_x = k, k, 1, 1, p, p, d, d
        pp.terms = 3
        for c in ops._bf16_candidates(pc, H, W, 1, k * k, 3):
            if not ops._cfg_ok(lib, pp, c):
                continue
            pc.tuned[key] = c
            try:
                y = ops.conv2d(None, pc, pad=p, dil=d, xs=xs)
            except Exception as ex:  # the shared split tensor does not fit this tile
                continue
            err = (y.cpu() - ref).abs().max().item() / scale
            if xso:
                t = timeit(lambda: ops.conv2d(None, pc, pad=p, dil=d, xs=xs, xs_out=so))
            else:
                t = timeit(lambda: ops.conv2d(None, pc, pad=p, dil=d, xs=xs, out=y))
            res.append((t, c, err))
            pc._packs.clear()
        print(f"[{li}] {cin}->{cout} k{k} d{d} {H}x{W} {'records' if xso else 'fp32'} out: {gflop:.2f} GFLOP")
        fam = {}
        for t, c, err in sorted(res, key=lambda r: r[0]):
            f = "ks2" if c[8] == 2 else ("8w" if c[5] * c[6] == 8 else "4w")
            if fam.setdefault(f, 0) < (int(os.environ.get('TOPN', 3))):
                fam[f] += 1
                print(f"   {f:4s} {t:7.1f} us {gflop / t * 1e3:6.1f} TF  (xb,th,ck,mb,_,pgw,cgw,terms,ks)={c} err {err:.1e}")
        bad = [r for r in res if not r[2] < 1e-4]
        if bad:
            print("      !!! WRONG:", [(c, e) for _, c, e in bad][:8])


if __name__ == "__main__":
    main()
