"""Dev tool: every split-bf16 configuration the library accepts for the level-0 all-pairs GEMM (1x1, 8640 <- 128 at
72x120), timed.  python tools/time_allpairs_split.py"""
import os
import sys
import ctypes as C

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codd_amd import _abi, ops  # noqa: E402

lib = _abi.load()
D, h, w = 128, 72, 120
N = h * w
f1 = torch.randn(1, D, h, w, device="cuda")
f2 = torch.randn(1, D, h, w, device="cuda")
out = torch.empty(N, N, device="cuda")
ref = torch.einsum("dn,dm->nm", f1[0].reshape(D, N).double(), f2[0].reshape(D, N).double()) / 16
res = []
for pgw, cgw, a, b, ks in ops._B_INST:
    mb = b * cgw
    for th, xb in ops._B_TILES[pgw * a]:
        for ck in (32, 64, 128, 16):
            c = (xb, th, ck, mb, 2, pgw, cgw, 3, ks)
            p = _abi.ConvParams()
            p.C0, p.C1, p.B, p.Hin, p.Win, p.Cout, p.Hout, p.Wout = D, 0, 1, h, w, N, h, w
            p.kh = p.kw = p.sy = p.sx = p.dil_y = p.dil_x = 1
            p.terms, p.out_ctot = 3, N
            if not ops._cfg_ok(lib, p, c):
                continue
            xs = ops.split_input_as(f2, "split", [c], p)
            nb = lib.codd_conv2d_packed_bytes_bf16(N, D, 1, 1, mb, ck, 3)
            wp = torch.empty(nb, device="cuda", dtype=torch.uint8)
            lib.codd_conv2d_pack_weights_bf16(f1.data_ptr(), wp.data_ptr(), N, D, 1, 1, mb, ck, 3, 1, N, 1.0 / 16.0, None)
            p.wpacked, p.xs, p.out = wp.data_ptr(), xs.buf.data_ptr(), out.data_ptr()
            p.xs_c8, p.xs_hp, p.xs_wp, p.xs_bt, p.xs_bl, p.xs_o8 = xs.c8, xs.hp, xs.wp, 0, 0, 0
            p.npb, p.nw, p.ck, p.mb, p.layout, p.pgw, p.cgw = c[:7]
            p.ksplit = ks
            if lib.codd_conv2d(C.byref(p), None) != 0:
                continue
            torch.cuda.synchronize()
            err = (out.double() - ref).abs().max().item()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                lib.codd_conv2d(C.byref(p), None)
            e.record()
            e.synchronize()
            res.append((s.elapsed_time(e) / 5 * 1e3, c, err))
for t, c, err in sorted(res)[:12]:
    print(f"{t:7.1f} us  {N * N * 4 / t / 1e6:5.2f} TB/s written  cfg {c}  max err {err:.1e}")
