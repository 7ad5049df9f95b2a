"""dev: where do the ~30 us of an HRNet multi-job launch go?  Times every branch convolution alone in the multi-job class
(quad layout, 4 x 16 tiles, 16 channels per workgroup), tuned stand-alone, and the 2- / 3- / 4-job launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
dev = "cuda:0"
ops.set_conv_precision("fp32")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


g = torch.Generator().manual_seed(0)
shapes = [(18, 144, 240), (36, 72, 120), (72, 36, 60), (144, 18, 30)]
jobs = []
for C, H, W in shapes:
    x = torch.randn(1, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev)
    pc = ops.PackedConv(w, torch.zeros(C, device=dev))
    jobs.append(dict(x=x, pc=pc, pad=1, act="relu"))
    key = (H, W, 1, 1, 1, 1, 1, 1, False, 0)
    line = f"{C:3d}->{C:<3d} {H}x{W}:"
    for ck in (16, 32):
        if ck == 16 and C <= 16:
            continue
        pc.tuned[key] = (1, 4, ck if C > 16 else 16, 1, 1)
        try:
            line += f"  multi-class ck{ck}: {timeit(lambda: ops.conv2d(x, pc, pad=1, act='relu')):6.1f} us"
        except Exception as e:
            line += f"  ck{ck}: {e!r}"
    pc.tuned.clear()
    ops.enable_autotune(True, shipped=True)
    ops.conv2d(x, pc, pad=1, act="relu")
    line += f"  | tuned {pc.tuned.get(key)}: {timeit(lambda: ops.conv2d(x, pc, pad=1, act='relu')):6.1f} us"
    ops.enable_autotune(False)
    print(line, flush=True)
for n in (2, 3, 4):
    print(f"multi launch of the first {n} jobs: {timeit(lambda: ops.conv2d_multi(jobs[:n])):6.1f} us", flush=True)
print(f"multi launch of jobs 2+3 (36, 72 ch): {timeit(lambda: ops.conv2d_multi(jobs[1:3])):6.1f} us")
print(f"multi launch of jobs 3+4 (72, 144 ch): {timeit(lambda: ops.conv2d_multi(jobs[2:4])):6.1f} us")
