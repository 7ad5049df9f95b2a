"""Dev tool: the persistent quad kernel (layout 3) against the tuned per-tile kernels on HITNet's big 16- / 32-channel
layers.  python tools/time_persist.py"""
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


ops.set_conv_precision("fp32")
for (cin, cout, k, H, W, B) in [(32, 32, 3, 288, 480, 1), (16, 16, 3, 576, 960, 1), (16, 16, 3, 576, 960, 2), (32, 32, 3, 144, 240, 1),
                                (32, 16, 1, 576, 960, 1), (16, 16, 3, 288, 480, 2), (24, 24, 3, 144, 240, 2)]:
    x = torch.randn(B, cin, H, W, device=DEV)
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    b = torch.randn(cout, device=DEV)
    pc = ops.PackedConv(w, b)
    key = (H, W, B, 1, 1, 1, 1, k // 2, False, 0)
    out = torch.empty(B, cout, H, W, device=DEV)
    gflop = 2.0 * cin * cout * k * k * H * W * B / 1e9
    res = []
    ck = 16 if cin <= 16 else 32
    cands = [(npb, 4, ck, mb, 3) for mb in (1, 2) if cout <= 16 * mb for npb in (1, 2, 4)]
    cands += [(npb, nw, ck, mb, 1) for mb in (1, 2) if cout <= 16 * mb or mb == 1 for npb, nw in ((1, 4), (2, 4), (4, 4), (1, 9))]
    cands += [(npb, 4, c, mb, 0) for mb in (1, 2) for npb in (1, 2, 4) for c in (16, 32) if c <= ck]
    import warnings
    for c in cands:
        pc.tuned[key] = c
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                t = timeit(lambda: ops.conv2d(x, pc, pad=k // 2, act="lrelu", out=out))
        except Exception:
            continue
        res.append((t, c))
    print(f"{cin}->{cout} k{k} {H}x{W} B{B}: {gflop:.2f} GFLOP")
    for t, c in sorted(res)[:8]:
        print(f"    {t:7.1f} us {gflop / t * 1e3:6.1f} TF  (npb,nw,ck,mb,layout)={c}")
