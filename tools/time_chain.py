"""Dev tool: LDS-resident chains (codd_conv_chain) against the same layers as separate launches, at the shapes of the
stereo network / HRNet.  python tools/time_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import ops  # noqa: E402

DEV = "cuda:0"


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


def mk(cout, cin, k):
    return (torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5, torch.randn(cout, device=DEV) * 0.1)


def case(name, B, H, W, spec, tiles=None):
    """spec: list of (cin, cout, k, dil, src, dst, res)."""
    layers, pcs = [], []
    for cin, cout, k, dil, src, dst, res in spec:
        w, b = mk(cout, cin, k)
        layers.append(dict(w=w, b=b, dil=dil, act="lrelu", src=src, dst=dst, res=res))
        pcs.append((ops.PackedConv(w, b), k, dil))
    pch = ops.PackedChain(layers, stage=0)
    x = torch.randn(B, spec[0][0], H, W, device=DEV)
    ops.set_conv_precision("fp32")
    ops.enable_autotune(True, shipped=True)

    def separate():
        t = x
        for pc, k, dil in pcs:
            t = ops.conv2d(t, pc, pad=dil * (k // 2), dil=dil, act="lrelu")
        return t

    separate()
    ops.enable_autotune(False)
    t_sep = timeit(separate)
    gflop = sum(2.0 * ci * co * k * k for ci, co, k, *_ in spec) * H * W * B / 1e9
    print(f"{name}: B{B} {H}x{W} {len(spec)} layers {gflop:.2f} GFLOP | separate {t_sep:7.1f} us ({gflop / t_sep * 1e3:5.1f} TF)")
    for tile in tiles or ops.PackedChain.TILES:
        try:
            t = timeit(lambda: ops.conv_chain(x, pch, tile=tile))
        except Exception:
            continue
        halo = sum(d * (k // 2) for _, _, k, d, *_ in spec)
        m, per = 0, 0
        for ci, co, k, d, *_ in spec:
            m += d * (k // 2)
            npx = (tile[0] + 2 * (halo - m)) * (tile[1] + 2 * (halo - m))
            per += -(-npx // 32) * 2 * k * k * 2 * -(-ci // 8) * -(-co // 16)
        wgs = -(-H // tile[0]) * -(-W // tile[1]) * B
        ideal = wgs * per * 32 / 1024 / 2.4e3  # us if every SIMD issued one MFMA per 32 cycles
        print(f"    chain tile {tile}: {t:7.1f} us ({gflop / t * 1e3:5.1f} TF)  {wgs} workgroups, MFMA floor {ideal:6.1f} us")


def rb(c, a, b, first=False, last=False, dil=1):
    return [(c, c, 3, dil, -1 if first else a, b, -1), (c, c, 3, dil, b, -1 if last else a, a)]


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    cases = [
        ("pair32@288x480", 1, 288, 480, rb(32, 0, 1, True, True)),
        ("pair16@576x960", 1, 576, 960, rb(16, 0, 1, True, True)),
        ("merge16@576x960 B2", 2, 576, 960, [(32, 16, 1, 1, -1, 0, -1), (16, 16, 3, 1, 0, 1, -1), (16, 16, 3, 1, 1, -1, -1)]),
        ("tileupdate32@144x240", 1, 144, 240, [(64, 32, 1, 1, -1, 0, -1)] + rb(32, 0, 1) + rb(32, 0, 1) + [(32, 34, 3, 1, 0, -1, -1)]),
        ("tileupdate32@36x60", 1, 36, 60, [(64, 32, 1, 1, -1, 0, -1)] + rb(32, 0, 1) + rb(32, 0, 1) + [(32, 34, 3, 1, 0, -1, -1)]),
        ("hrnet18@144x240", 1, 144, 240, rb(18, 0, 1, True) + rb(18, 0, 1, False, True)),
        ("hrnet36@72x120", 1, 72, 120, rb(36, 0, 1, True) + rb(36, 0, 1, False, True)),
        ("single32@288x480", 1, 288, 480, [(32, 32, 3, 1, -1, -1, -1)]),
    ]
    tu = [(64, 32, 1, 1, -1, 0, -1)] + rb(32, 0, 1) + rb(32, 0, 1) + [(32, 34, 3, 1, 0, -1, -1)]
    small = ((4, 4), (4, 8), (8, 8), (2, 8), (4, 16))
    cases += [("tiny tileupdate32@9x15", 1, 9, 15, tu, small), ("tiny tileupdate32@18x30", 1, 18, 30, tu, small),
              ("tiny tileupdate32@36x60", 1, 36, 60, tu, small), ("tiny tileupdate32@72x120", 1, 72, 120, tu, small),
              ("tiny merge24@72x120 B2", 2, 72, 120, [(48, 24, 1, 1, -1, 0, -1), (24, 24, 3, 1, 0, 1, -1), (24, 24, 3, 1, 1, -1, -1)], small),
              ("tiny down4tail32@36x60 B2", 2, 36, 60, [(32, 32, 3, 1, -1, 1, -1), (32, 32, 3, 1, 1, 0, -1), (32, 32, 3, 1, 0, -1, -1)], small)]
    for c in cases:
        if only in c[0]:
            case(*c)
