"""dev: micro-benchmarks of individual kernels at the 960x576 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
dev = "cuda:0"

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

which = sys.argv[1:] or ["gn", "conv"]
h, w = 72, 120
if any(a.startswith("gn") for a in which):
    T = ops.se3_identity(1, h, w, dev)
    ae = torch.randn(1, 32, h, w, device=dev)
    xyz = torch.rand(1, h, w, 3, device=dev) * 50
    delta = torch.randn(1, 3, h, w, device=dev)
    wgt = torch.rand(1, 3, h, w, device=dev)
    d1 = torch.rand(1, h, w, device=dev) * 50 + 1
    K8 = [131.0, 131.0, 60.0, 36.0]
    for r in ([int(a[2:]) for a in which if a.startswith('gn') and a[2:]] or (32, 8)):
        print(f"gn radius {r}: {timeit(lambda: ops.se3_gn_step(T, ae, xyz, delta, wgt, d1, K8, radius=r)):.1f} us")
if "conv" in which:
    def conv_case(cin, cout, k, H, W, dil=1, stride=1, n=10):
        x = torch.randn(1, cin, H, W, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        pc = ops.PackedConv(wt, torch.zeros(cout, device=dev))
        pad = dil * (k // 2)
        t = timeit(lambda: ops.conv2d(x, pc, pad=pad, dil=dil, stride=stride, act="relu"), n)
        Ho, Wo = H // stride, W // stride
        fl = 2.0 * cin * cout * k * k * Ho * Wo
        print(f"conv {cin:4d}->{cout:4d} k{k} d{dil} s{stride} {H}x{W}: {t:8.1f} us  {fl / t / 1e6:7.2f} TFLOP/s")
    conv_case(128, 256, 3, h, w); conv_case(128, 256, 3, h, w, dil=4); conv_case(128, 1024, 3, h, w)
    conv_case(196, 256, 3, h, w); conv_case(256, 256, 3, h, w); conv_case(128, 128, 3, h, w)
    conv_case(256, 384, 1, h, w); conv_case(9, 128, 7, h, w)
    conv_case(64, 64, 3, 288, 480); conv_case(96, 96, 3, 144, 240); conv_case(16, 16, 3, 576, 960)
    conv_case(32, 32, 3, 288, 480); conv_case(32, 32, 3, 144, 240); conv_case(64, 30, 7, 144, 240)
    conv_case(3, 64, 7, 576, 960, stride=2); conv_case(32, 16, 1, 576, 960)
