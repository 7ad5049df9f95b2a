"""Dev aid: time the GRU-loop conv shapes (72x120 maps) for every workgroup height nw."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codd_amd import ops
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


H, W = 72, 120
shapes = [(128, 768, 3, 1), (128, 256, 3, 1), (128, 256, 3, 4), (256, 256, 3, 1), (128, 128, 3, 1), (196, 256, 3, 1),
          (256, 384, 1, 1), (128, 384, 1, 1), (128, 1024, 3, 1)]
for cin, cout, k, dil in shapes:
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    x = torch.randn(1, cin, H, W, device=dev)
    ref = None
    line = f"{cin:4d}->{cout:4d} k{k} d{dil}:"
    for mb in ((2, 4) if cout <= 128 else (4,)):
        os.environ["CODD_MB"] = str(mb)
        ops._FORCE_MB = mb if hasattr(ops, "_FORCE_MB") else None
        pc = ops.PackedConv(wt, torch.zeros(cout, device=dev))
        for nw in (4, 9):
            ops._FORCE_NW = nw
            try:
                y = ops.conv2d(x, pc, pad=dil * (k // 2), dil=dil, act="relu")
                if ref is None: ref = y.clone()
                err = (y - ref).abs().max().item()
                t = timeit(lambda: ops.conv2d(x, pc, pad=dil * (k // 2), dil=dil, act="relu"))
                line += f"  mb{pc.mb}/nw{nw} {t:6.1f}us {2.0*cin*cout*k*k*H*W/t/1e6:5.1f}TF" + ("" if err == 0 else f" ERR{err:.1e}")
            except Exception as ex:
                line += f"  mb{mb}/nw{nw} FAIL({str(ex)[:30]})"
    print(line, flush=True)
