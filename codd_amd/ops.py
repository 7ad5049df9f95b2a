"""Tensor-level wrappers over the C ABI (include/codd_hip.h).

PyTorch is used for device memory and streams only; every computation below is a call into
libcodd_hip.so.  There is no CPU / eager fallback: tensors must live on a ROCm device.
"""
import ctypes as C

import torch

from . import _abi
from ._abi import ACT, ConvParams, View


class Slice:
    """Channels [coff, coff + c) of a contiguous NCHW buffer."""
    __slots__ = ("buf", "coff", "c")

    def __init__(self, buf, coff=0, c=None):
        assert buf.is_contiguous() and buf.dtype == torch.float32
        self.buf, self.coff = buf, coff
        self.c = buf.shape[1] - coff if c is None else c

    @property
    def shape(self):
        return (self.buf.shape[0], self.c) + tuple(self.buf.shape[2:])

    def tensor(self):
        return self.buf[:, self.coff:self.coff + self.c]


def _as_slice(x):
    return x if isinstance(x, Slice) else Slice(x)


def _view(x):
    if x is None:
        return View(None, 0, 0)
    s = _as_slice(x)
    return View(s.buf.data_ptr(), s.buf.shape[1], s.coff)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t):
    if not t.is_cuda:
        raise _abi.CoddHipError("codd_amd ops run on a ROCm device only (no CPU fallback in the product path)")


# ----------------------------------------------------------------------------------------- conv
class PackedConv:
    """Weights of one Conv2d / ConvTranspose2d(k2,s2) re-laid-out for the MFMA kernel."""

    def __init__(self, weight, bias, deconv=False, cout_keep=None):
        _require_gpu(weight)
        w = weight.detach().float()
        b = None if bias is None else bias.detach().float().contiguous()
        if deconv:  # [Cin, Cout, 2, 2] -> 1x1 conv with 4*Cout outputs, co' = (a*2+b)*Cout + co
            cin, cout = w.shape[:2]
            w = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)
            self.cout = cout
        else:
            if cout_keep is not None:
                w = w[:cout_keep]
                b = None if b is None else b[:cout_keep].contiguous()
            self.cout = w.shape[0]
        w = w.contiguous()
        self.deconv = deconv
        self.cout_eff, self.cin, self.kh, self.kw = w.shape
        self.bias = b
        self.mb = 1 if self.cout_eff <= 16 else (2 if self.cout_eff <= 32 else 4)
        self._w = w
        self._packs = {}

    def packed(self, ck):
        if ck not in self._packs:
            lib = _abi.load()
            n = lib.codd_conv2d_packed_size(self.cout_eff, self.cin, self.kh, self.kw, self.mb, ck)
            wp = torch.empty(n, device=self._w.device, dtype=torch.float32)
            _abi.check(lib.codd_conv2d_pack_weights(self._w.data_ptr(), wp.data_ptr(), self.cout_eff, self.cin,
                                                    self.kh, self.kw, self.mb, ck, _stream()), "pack_weights")
            self._packs[ck] = wp
        return self._packs[ck]


def _wrow(mb):
    return 16 * mb + (0 if mb & 1 else 16)


def _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx):
    """(npb, ck): tile shape per wave and LDS chunk depth."""
    ncog = -(-pc.cout_eff // (16 * pc.mb))
    nblk4 = (-(-Hout // 8)) * (-(-Wout // 32)) * ncog * B
    nblk2 = (-(-Hout // 4)) * (-(-Wout // 32)) * ncog * B
    npb = 4 if nblk4 >= 512 else (2 if nblk2 >= 384 else 1)
    xb = 2 if npb >= 2 else 1
    th, tw = 4 * (npb // xb), 16 * xb
    thi = (th - 1) * sy + (pc.kh - 1) * dy + 1
    twi = (tw - 1) * sx + (pc.kw - 1) * dx + 1
    per = thi * twi
    chs = ((per + 15) // 32) * 32 + 16 if sx == 1 else per | 1
    cin_pad = -(-pc.cin // 4) * 4
    ck = min(cin_pad, 32)
    budget = 64 * 1024
    while ck > 4 and (pc.kh * pc.kw * ck * _wrow(pc.mb) + ck * chs) * 4 > budget:
        ck -= 4
    return npb, ck


def conv2d(x, pc, x2=None, stride=1, pad=0, dil=1, act="none", res1=None, res2=None, post=None,
           out=None, pad_tl=None, out_hw=None):
    """act(conv(cat[x, x2]) + bias + res1 + res2) + post  ->  out (tensor or Slice)."""
    lib = _abi.load()
    xs = _as_slice(x)
    _require_gpu(xs.buf)
    B, C0, Hin, Win = xs.shape
    C1 = 0
    if x2 is not None:
        x2s = _as_slice(x2)
        C1 = x2s.c
        assert x2s.shape[2:] == (Hin, Win)
    assert C0 + C1 == pc.cin, (C0, C1, pc.cin)
    sy, sx = (stride, stride) if isinstance(stride, int) else stride
    dy, dx = (dil, dil) if isinstance(dil, int) else dil
    if pad_tl is None:
        pt, pl = (pad, pad) if isinstance(pad, int) else pad
        pb_, pr_ = pt, pl
    else:
        pt, pl, pb_, pr_ = pad_tl
    if pc.deconv:
        Hout, Wout = Hin, Win
    elif out_hw is not None:
        Hout, Wout = out_hw
    else:
        Hout = (Hin + pt + pb_ - dy * (pc.kh - 1) - 1) // sy + 1
        Wout = (Win + pl + pr_ - dx * (pc.kw - 1) - 1) // sx + 1
    up = 2 if pc.deconv else 1
    if out is None:
        out = torch.empty(B, pc.cout, Hout * up, Wout * up, device=xs.buf.device, dtype=torch.float32)
    os_ = _as_slice(out)
    assert os_.shape == (B, pc.cout, Hout * up, Wout * up), (os_.shape, (B, pc.cout, Hout * up, Wout * up))
    npb, ck = _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx)
    p = ConvParams()
    p.in0 = _view(xs)
    p.in1 = _view(x2)
    p.C0, p.C1, p.B, p.Hin, p.Win = C0, C1, B, Hin, Win
    p.wpacked = pc.packed(ck).data_ptr()
    p.bias = None if pc.bias is None else pc.bias.data_ptr()
    p.res1, p.res2, p.post = _view(res1), _view(res2), _view(post)
    p.out, p.out_ctot, p.out_coff = os_.buf.data_ptr(), os_.buf.shape[1], os_.coff
    p.Cout, p.Hout, p.Wout = pc.cout, Hout, Wout
    p.kh, p.kw, p.sy, p.sx, p.pad_t, p.pad_l, p.dil_y, p.dil_x = pc.kh, pc.kw, sy, sx, pt, pl, dy, dx
    p.act = ACT[act]
    p.store_mode = 1 if pc.deconv else 0
    p.mb, p.npb, p.ck = pc.mb, npb, ck
    _abi.check(lib.codd_conv2d(C.byref(p), _stream()), "codd_conv2d")
    return out


# ----------------------------------------------------------------------------------------- stereo
def tile_costvol_argmin(tl, tr, D, cost, hyp):
    """cost: Slice (1 ch) receiving the min cost; hyp: Slice whose channels 0..2 receive (d, 0, 0)."""
    lib = _abi.load()
    _require_gpu(tl)
    B, Cc, Ht, Wt = tl.shape
    cs, hs = _as_slice(cost), _as_slice(hyp)
    _abi.check(lib.codd_tile_costvol_argmin(tl.data_ptr(), tr.data_ptr(), B, Cc, Ht, Wt, tr.shape[3], D,
                                            cs.buf.data_ptr(), cs.buf.shape[1], cs.coff,
                                            hs.buf.data_ptr(), hs.buf.shape[1], hs.coff, 1, _stream()),
               "tile_costvol_argmin")


def tile_warp_cost(fl, fr, hyp0, hyp1=None):
    """-> one or two [B,64,Ht,Wt] tensors ([fea 16 | cv 48])."""
    lib = _abi.load()
    _require_gpu(fl)
    B, Cc, H, W = fl.shape
    Ht, Wt = H // 4, W // 4
    out0 = torch.empty(B, 64, Ht, Wt, device=fl.device, dtype=torch.float32)
    out1 = torch.empty_like(out0) if hyp1 is not None else None
    _abi.check(lib.codd_tile_warp_cost(fl.data_ptr(), fr.data_ptr(), B, Cc, Ht, Wt, _view(hyp0), _view(hyp1),
                                       2 if hyp1 is not None else 1, out0.data_ptr(),
                                       None if out1 is None else out1.data_ptr(), _stream()), "tile_warp_cost")
    return out0, out1


def hyp_upsample(h, scale, out):
    lib = _abi.load()
    hs, os_ = _as_slice(h), _as_slice(out)
    B, _, hh, ww = hs.shape
    _abi.check(lib.codd_hyp_upsample(_view(hs), B, hh, ww, float(scale), os_.buf.data_ptr(), os_.buf.shape[1],
                                     os_.coff, _stream()), "hyp_upsample")
    return out


def hyp_select(upd, cur, prev, out):
    lib = _abi.load()
    B, _, hh, ww = upd.shape
    os_ = _as_slice(out)
    _abi.check(lib.codd_hyp_select(upd.data_ptr(), _view(cur), _view(prev), B, hh, ww, os_.buf.data_ptr(),
                                   os_.buf.shape[1], os_.coff, _stream()), "hyp_select")
    return out
