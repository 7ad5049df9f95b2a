"""dev: correctness (vs torch CPU fp32) and timing (vs the tuned tile kernels) of codd_conv_roll at HITNet's shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from codd_amd import ops

dev = "cuda:0"


def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g)


def lrelu(x):
    return F.leaky_relu(x, 0.2)


def ref_run(mode, x, wa, ba, wb, bb, residual):
    if mode == 0:
        return lrelu(F.conv2d(x, wa, ba, 1, 1))
    if mode == 1:
        t = lrelu(F.conv2d(x, wa, ba, 1, 1))
        y = F.conv2d(t, wb, bb, 1, 1)
        return lrelu(y + x if residual else y)
    t = lrelu(F.conv2d(x, wa, ba))
    return lrelu(F.conv2d(t, wb, bb, 1, 1))


def make(mode, C, cin, seed=0):
    k0 = 1 if mode == 2 else 3
    wa = rnd(C, cin, k0, k0, seed=seed + 1) / (cin * k0 * k0) ** 0.5
    ba = rnd(C, seed=seed + 2) * 0.1
    wb = rnd(C, C, 3, 3, seed=seed + 3) / (C * 9) ** 0.5
    bb = rnd(C, seed=seed + 4) * 0.1
    return wa, ba, wb, bb


def packed(mode, wa, ba, wb, bb, residual):
    st = [dict(w=wa.to(dev), b=ba.to(dev), act="lrelu")]
    if mode != 0:
        st.append(dict(w=wb.to(dev), b=bb.to(dev), act="lrelu"))
    return ops.PackedRoll(st, residual=residual)


def check():
    worst = 0.0
    for mode, C, cin, B, H, W, res, rh in [(0, 16, 16, 1, 37, 130, False, 8), (1, 16, 16, 2, 29, 75, True, 7),
                                           (1, 16, 16, 1, 64, 200, False, 64), (2, 16, 32, 1, 33, 190, False, 5),
                                           (0, 32, 32, 1, 21, 64, False, 4), (1, 32, 32, 1, 40, 125, True, 9),
                                           (2, 32, 40, 2, 19, 61, False, 19), (2, 32, 64, 1, 17, 70, False, 6),
                                           (1, 16, 16, 1, 5, 9, True, 3), (2, 16, 48, 1, 12, 63, False, 4)]:
        wa, ba, wb, bb = make(mode, C, cin)
        x = rnd(B, cin, H, W, seed=9)
        ref = ref_run(mode, x, wa, ba, wb, bb, res)
        pr = packed(mode, wa, ba, wb, bb, res)
        xg = x.to(dev)
        if mode == 2 and cin > 16:  # two sources
            c0 = 16 if cin != 40 else 24
            got = ops.conv_roll(xg[:, :c0].contiguous(), pr, x2=xg[:, c0:].contiguous(), rh=rh)
        else:
            got = ops.conv_roll(xg, pr, rh=rh)
        err = (got.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        worst = max(worst, err)
        print(f"mode {mode} C {C} cin {cin} B {B} {H}x{W} res {res} rh {rh}: max rel err {err:.2e}", flush=True)
        assert err < 2e-5, err
    print("roll check OK, worst", worst)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def bench():
    ops.enable_autotune(True, shipped=True)
    for mode, C, cin, B, H, W, res in [(1, 16, 16, 1, 576, 960, True), (1, 16, 16, 2, 576, 960, False),
                                       (0, 16, 16, 1, 576, 960, False), (2, 16, 32, 1, 576, 960, False),
                                       (2, 16, 32, 2, 576, 960, False), (1, 16, 16, 2, 288, 480, False),
                                       (1, 32, 32, 1, 288, 480, True), (0, 32, 32, 1, 288, 480, False),
                                       (2, 32, 32, 1, 288, 480, False), (1, 32, 32, 1, 144, 240, True),
                                       (2, 32, 40, 1, 144, 240, False)]:
        wa, ba, wb, bb = make(mode, C, cin)
        x = rnd(B, cin, H, W, seed=9).to(dev)
        pr = packed(mode, wa, ba, wb, bb, res)
        k0 = 1 if mode == 2 else 3
        pca = ops.PackedConv(wa.to(dev), ba.to(dev))
        pcb = ops.PackedConv(wb.to(dev), bb.to(dev))
        out = torch.empty(B, C, H, W, device=dev)
        tmp = torch.empty(B, C, H, W, device=dev)

        def old():
            if mode == 0:
                ops.conv2d(x, pca, pad=1, act="lrelu", out=out)
            else:
                ops.conv2d(x, pca, pad=k0 // 2, act="lrelu", out=tmp)
                ops.conv2d(tmp, pcb, pad=1, act="lrelu", res1=x if res else None, out=out)
        old()
        t_old = timeit(old)
        ref = out.clone()
        line = f"mode {mode} C {C} cin {cin} B {B} {H}x{W}: tile kernels {t_old:7.1f} us | roll"
        best = None
        for rh in sorted({ops._roll_rh(B, H, W, mode), 8, 12, 16, 24, 36, 48, 72}):
            if rh > H:
                continue
            t = timeit(lambda: ops.conv_roll(x, pr, out=out, rh=rh))
            line += f"  rh{rh}:{t:6.1f}"
            best = t if best is None else min(best, t)
        err = (out - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        flops = 2.0 * B * H * W * C * (cin * k0 * k0 + (C * 9 if mode else 0))
        print(line + f"  | best {best:.1f} us = {flops / best / 1e6:.1f} TF  (x{t_old / best:.2f})  err vs tile {err:.1e}", flush=True)


if __name__ == "__main__":
    check()
    if "--bench" in sys.argv:
        bench()
