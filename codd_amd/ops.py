"""Tensor-level wrappers over the C ABI (include/codd_hip.h).

PyTorch is used for device memory and streams only; every computation below is a call into
libcodd_hip.so.  There is no CPU / eager fallback: tensors must live on a ROCm device.
"""
import ctypes as C

import torch

from . import _abi
from ._abi import ACT, ConvParams, View


class RuntimeState:
    """Mixin of the plug-in modules that keep launch-time objects (side streams, fork helpers, pending prefetches,
    captured frame graphs) in their ``__dict__``: those are neither picklable nor copyable and are rebuilt on demand,
    so they are left out of the module's pickled / deep-copied state."""
    _RUNTIME_KEYS = ("_side", "_pending", "_nowait", "_fk", "_xs", "_runners", "_kside", "_h4c", "_ctx4c", "_fused_now", "_pipe")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._RUNTIME_KEYS:
            state.pop(k, None)
        return state


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
        # 16-channel blocks per workgroup: 64-channel groups only for wide layers that fill them
        # exactly (measured: 96->96 3x3 87 us at mb=2 vs 116 us at mb=4)
        self.mb = 1 if self.cout_eff <= 16 else (4 if (self.cout_eff >= 256 and self.cout_eff % 64 == 0) else 2)
        if _FORCE_MB and self.cout_eff > 32:
            self.mb = _FORCE_MB
        self._w = w
        self._packs = {}
        self.tuned = {}  # launch shape -> (npb, nw, ck)

    def packed(self, ck, mb=None, layout=0):
        """layout 0 / 1: fp32 kernels; 20 + terms (21 | 23): split-bf16 kernel with 1 | 3 product terms."""
        mb = self.mb if mb is None else mb
        if layout >= 20 and (mb, ck, layout) not in self._packs:
            lib = _abi.load()
            terms = layout - 20
            n = lib.codd_conv2d_packed_bytes_bf16(self.cout_eff, self.cin, self.kh, self.kw, mb, ck, terms)
            if n <= 0:
                raise _abi.CoddHipError("no split-bf16 packed layout for ck=%d mb=%d terms=%d" % (ck, mb, terms))
            wp = torch.empty(n, device=self._w.device, dtype=torch.uint8)
            _abi.check(lib.codd_conv2d_pack_weights_bf16(self._w.data_ptr(), wp.data_ptr(), self.cout_eff, self.cin,
                                                         self.kh, self.kw, mb, ck, terms, self.cin * self.kh * self.kw,
                                                         self.kh * self.kw, 1.0, _stream()), "pack_weights_bf16")
            self._packs[(mb, ck, layout)] = wp
        if (mb, ck, layout) not in self._packs:
            lib = _abi.load()
            size, pack = ((lib.codd_conv2d_packed_size_quad, lib.codd_conv2d_pack_weights_quad) if layout == 1 else
                          (lib.codd_conv2d_packed_size, lib.codd_conv2d_pack_weights))
            n = size(self.cout_eff, self.cin, self.kh, self.kw, mb, ck)
            if n <= 0:
                raise _abi.CoddHipError("no packed layout %d for ck=%d mb=%d" % (layout, ck, mb))
            wp = torch.empty(n, device=self._w.device, dtype=torch.float32)
            _abi.check(pack(self._w.data_ptr(), wp.data_ptr(), self.cout_eff, self.cin, self.kh, self.kw, mb, ck,
                            _stream()), "pack_weights")
            self._packs[(mb, ck, layout)] = wp
        return self._packs[(mb, ck, layout)]

    def _pack_key(self, c):
        lay = c[4] if len(c) > 4 else 0
        return (c[3] if len(c) > 3 else self.mb, c[2], 20 + c[7] if lay == 2 else (1 if lay == 3 else lay))

    def drop_unused_packs(self):
        """Free the packed-weight variants no tuned configuration refers to (after autotuning)."""
        used = {self._pack_key(c) for c in self.tuned.values()}
        for k in [k for k in self._packs if k not in used]:
            del self._packs[k]


def _launch_conv(lib, p, stream):
    """Single choke point of every conv launch (bench.py wraps it with HIP events)."""
    return lib.codd_conv2d(C.byref(p), stream)


import os as _os
_FORCE_NPB = 0
_FORCE_MB = 0
_FORCE_CK = 0
_FORCE_NW = 0


def _wrow(mb):
    return 16 * mb + (0 if mb & 1 else 16)


def _pick_nw(pc, npb, Hout, Wout, B, ncog):
    """Waves (= 16-pixel tile rows) per workgroup: 4, or 9 where that removes a nearly empty last round.
    The 256 CUs take workgroups round-robin, so a launch lasts about ceil(grid / 256) rounds of nw rows;
    the 72-row GRU maps with 4 channel groups give 576 blocks = 3 rounds x 4 rows with nw = 4 and 256 blocks
    = 1 round x 9 rows with nw = 9 (measured 88 -> 82, 165 -> 153, 133 -> 125 us on the 128/256/196 -> 256
    3x3 layers; tools/time_conv_sweep.py).  Only for the plain 4x16 tile with >= 32 channels per group."""
    if _FORCE_NW:
        return _FORCE_NW if npb == 1 else 4
    if npb != 1 or pc.mb < 2 or Hout * Wout > 16384:
        return 4
    tx = -(-Wout // 16)
    cost = {nw: -(-(-(-Hout // nw) * tx * ncog * B) // 256) * nw for nw in (4, 9)}
    return 9 if cost[9] * 1.2 <= cost[4] else 4


def _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx, pl):
    """(npb, nw, ck): tile shape per wave, waves per workgroup and LDS chunk depth.

    Measured on MI355X (tools/time_ops.py): the small 4x16-pixel tile (npb = 1) wins on every
    layer with more than 16 output channels -- the kernel is latency / barrier bound, so more
    resident workgroups beat operand re-use; 16-channel layers prefer 4x32.  The LDS chunk is sized
    so that the whole grid is resident at once when possible (a 576-block grid at 2 blocks/CU runs
    a second, nearly empty round: 112 us vs 67 us for 504 blocks).  Limits of the kernel's register
    staging: <= 16 float4 of weights and <= 8 float4 of input per thread and chunk."""
    ncog = -(-pc.cout_eff // (16 * pc.mb))
    npb = 2 if pc.cout_eff <= 16 else 1
    if _FORCE_NPB:
        npb = _FORCE_NPB
    nw = _pick_nw(pc, npb, Hout, Wout, B, ncog)
    nt = 64 * nw
    xb = 2 if npb >= 2 else 1
    th, tw = nw * (npb // xb), 16 * xb
    grid = (-(-Hout // th)) * (-(-Wout // tw)) * ncog * B
    per_cu = min(max(-(-grid // 256), 2 if nw == 4 else 1), 4)
    budget = min(64 * 1024, (160 * 1024) // per_cu - 512)
    thi = (th - 1) * sy + (pc.kh - 1) * dy + 1
    twi = (tw - 1) * sx + (pc.kw - 1) * dx + 1
    xoff = (4 - pl % 4) % 4
    twp = -(-(xoff + twi) // 4) * 4
    per = thi * twp
    chs = ((per + 15) // 32) * 32 + 16 if sx == 1 else per + 4
    cin_pad = -(-pc.cin // 4) * 4
    ck = min(cin_pad, 32)
    taps = pc.kh * pc.kw
    while ck > 4 and ((taps * ck * _wrow(pc.mb) + ck * chs) * 4 > budget or taps * ck * _wrow(pc.mb) > 64 * nt
                      or ck * thi * (twp // 4) > (8 if nw == 4 else 4) * nt):
        ck -= 4
    if _FORCE_CK:
        ck = min(ck, _FORCE_CK)
    return npb, nw, ck


# Arithmetic of the convolution family (include/codd_hip.h, codd_conv_params.layout / terms):
#   "split" (default)  split-bf16 operands, three bf16 MFMAs per product, fp32 accumulate (fp32-grade: the parity
#                      tests at the benchmarked configurations bound its effect on the disparities)
#   "fp32"             exact-fp32 MFMA kernels (v_mfma_f32_16x16x4_f32)
#   "bf16"             bf16 operands, fp32 accumulate (BASELINE.json configs[4]; reference auto_fp16 hook)
#   "split16"          split-FP16 operands (hi | lo IEEE fp16 planes, three v_mfma_f32_16x16x32_f16 per product): the split
#                      scheme with 22-bit operands, products to ~2^-22 -- fp32's own grade -- at the same MFMA rate as
#                      "split" (16-bit operands, 2^-17); fp16's range applies (|x| <= 65504)
#   "fp16"             IEEE fp16 operands (v_mfma_f32_16x16x32_f16), fp32 accumulate: the reference's own reduced
#                      precision (auto_fp16 IS .half(), model/codd.py:37,128): 11 mantissa bits at bf16's MFMA rate
CONV_PRECISION = "split"  # set through set_conv_precision() / bench.py --precision, never through the environment
ALLPAIRS_SPLIT = True  # (A/B switch of allpairs_corr)
_TERMS = dict(split=3, bf16=1, fp16=16, split16=48)  # codd_conv_params.terms (CODD_TERMS_*)


# "bf16" / "fp16" with BF16_STAGE_POLICY on ("bf16mix" / "fp16mix" in bench.py / set_conv_precision): the stages of
# _STAGE_PRECISION keep their exact-fp32 kernels (HITNet, whose output IS the disparity, the context network, Fusion) and
# only RAFT3D's feature encoder and its 16 update iterations -- 85 % of the frame's convolution FLOPs -- run on plain
# bf16 / fp16 operands.
BF16_STAGE_POLICY = False
_HALF_MODES = ("bf16", "fp16")  # one 16-bit operand plane, one MFMA per product
_MODES = ("split", "split16", "fp32", "bf16", "bf16mix", "fp16", "fp16mix")
_SPLIT_MODES = ("split", "split16")  # three MFMAs per product on hi | lo operand planes: the fp32-grade modes


def _split_terms():
    """terms of the all-pairs GEMMs / forced-split re-layouts under the current mode (3 | 48)"""
    return _TERMS["split16"] if CONV_PRECISION == "split16" else 3


def set_conv_precision(mode):
    """-> the previous mode (pass it back to restore).  "bf16mix" / "fp16mix" = "bf16" / "fp16" + the stage policy.
    Range caveat of the fp16 formats ("fp16", "split16", "fp16mix"): IEEE fp16 records saturate to +-inf above 65504 and lo parts
    below ~6e-8 flush to zero; plain "fp16" / "split16" apply to EVERY stage including HITNet (whose channels carry raw
    disparities), "fp16mix" only to RAFT3D's feature encoder and update block (O(1) activations) -- the supported use.  Nothing
    guards against overflow at run time except ``FrameRunner(check_finite=True)`` (debug) and the parity tests."""
    global CONV_PRECISION, BF16_STAGE_POLICY
    if mode not in _MODES:
        raise ValueError("conv precision must be one of %s" % (_MODES,))
    prev = CONV_PRECISION + "mix" if CONV_PRECISION in _HALF_MODES and BF16_STAGE_POLICY else CONV_PRECISION
    CONV_PRECISION, BF16_STAGE_POLICY = (mode[:-3], True) if mode.endswith("mix") else (mode, False)
    return prev


# Stage policy under the default "split" mode.  Measured on MI355X (tools/debug_split_layers.py, headline parity tests):
# a split-bf16 conv is accurate to ~1e-5 of its OUTPUT SCALE.  HITNet's propagation layers carry raw disparities
# (~100 px) in their input channels and emit the disparity itself, so 1e-5 relative is ~1e-3 px absolute -- the whole
# north-star budget -- and its arg-min / arg-max selections flip for ~1.5 % of the pixels; the stereo network
# (7 % of the frame's conv FLOPs) therefore stays on the exact-fp32 kernels, everything else (RAFT3D encoders and the
# 16 update iterations: ~90 % of the FLOPs, O(1) feature scales, smooth outputs) runs split-bf16.
# Round 4: Fusion as well.  Over the configured sequence length (tests/test_gpu_headline_parity.py::
# test_recurrent_sequence_matches_oracle, 16 frames) the split-bf16 error of Fusion's weight path is multiplied by
# |pred_warp - pred_curr| (up to 250 px with the synthetic weights, whose weight-head logits saturate the sigmoid): from
# frame 6 on isolated 4x4 blocks take the other side of a 0 | 1 fusion weight and the all-pixel mean leaves the 1e-3 px
# budget (6.6e-3 ... 1.5e-2 px); with Fusion's ~10 quarter-resolution layers on the exact-fp32 kernels frames 0-13 stay
# at <= 1e-4 px (profiles/r04_sequence_divergence.log).  (`_STAGE_PRECISION["fusion"] = "split"` restores the round-3 policy.)
_STAGE_PRECISION = dict(stereo="fp32", context="fp32", fusion="fp32")


class stage:
    """``with ops.stage("stereo"): ...`` -- conv precision of a pipeline stage (only the "split" default is refined;
    explicit "fp32" / "bf16" modes apply to every stage)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = None
        if self.name in _STAGE_PRECISION and (CONV_PRECISION in _SPLIT_MODES or (CONV_PRECISION in _HALF_MODES and BF16_STAGE_POLICY)):
            to = _STAGE_PRECISION[self.name]
            if to != "split" or CONV_PRECISION == "split":  # (a "split" stage under bf16mix stays bf16)
                self.prev = set_conv_precision(to)

    def __exit__(self, *exc):
        if self.prev is not None:
            set_conv_precision(self.prev)
        return False


# Launch-configuration hint for convolutions that run BESIDE other work on side streams (``with ops.coresident():``):
# only 8-wave workgroups with <= ~106 registers per wave are tried, which leave half of a CU's wave slots and
# registers to the partner kernel (a stand-alone timing would pick 12-wave workgroups that own the CU).
_CORESIDENT = False


class coresident:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global _CORESIDENT
        self.prev, _CORESIDENT = _CORESIDENT, self.on
        return self

    def __exit__(self, *exc):
        global _CORESIDENT
        _CORESIDENT = self.prev
        return False


def _small_footprint(cands):
    small = [c for c in cands if c[5] * c[6] * (c[8] if len(c) > 8 else 1) == 4 and c[3] // c[6] <= 2]
    return small or cands


# split-bf16 kernel instantiations (conv_bf16_kernel.h): (pgw, cgw, A, B) and the tiles (rows, units per row) tried
_B_INST = ((2, 2, 5, 2, 1), (4, 1, 4, 4, 1), (4, 1, 4, 2, 1), (4, 1, 4, 1, 1), (4, 1, 8, 1, 1), (4, 1, 2, 2, 1),
           (4, 1, 2, 1, 1), (4, 1, 3, 4, 1), (4, 1, 3, 2, 1),
           # 8 consumer waves (two per SIMD): k-split pairs (ks = 2) and 4x2 wave grids that split the tile
           (2, 2, 5, 2, 2), (4, 1, 4, 2, 2), (4, 1, 3, 2, 2), (4, 1, 3, 4, 2), (4, 2, 4, 2, 1), (4, 2, 3, 2, 1),
           (8, 1, 4, 4, 1))
# tiles (rows, 16-pixel units per row) by pixel units of a workgroup (= pgw * a)
_B_TILES = {10: ((9, 1), (10, 1), (5, 2)), 16: ((8, 2), (16, 1)), 32: ((16, 2),), 8: ((4, 2), (8, 1)), 12: ((12, 1), (6, 2))}
# (round 6 added 96-channel groups (4,2,3,3) / (2,2,5,3) and 20-unit tiles (4,2,5,2) so that the 384-channel gate-input convolution
# -- 288 workgroups = 1.12 dispatch rounds on its 64-channel x 12-unit grid -- could run as ONE round: every single-round
# configuration is SLOWER (33.8 us at 192 / 216 / 240 workgroups against 29.8 us at 288: tools/sweep_update_block.py,
# profiles/r06_sweep_update_block.log), the shipped picks are the optimum of the extended space on all seven update-block
# layers, and the instantiations were removed again.)


def _bf16_candidates(pc, Hout, Wout, B, taps, terms):
    """Launch configurations (xb, th, ck, mb, 2, pgw, cgw, terms, ksplit) of the split-bf16 kernel for this layer, best guess first:
    fewest dispatch rounds over the 256 CUs times work per workgroup, larger channel groups and deeper chunks first."""
    nblk = -(-pc.cout_eff // 16)
    cands = []
    for (pgw, cgw, a, b, ks) in _B_INST:
        mb = b * cgw
        if mb > 1 and mb // 2 >= nblk:  # a channel group at least twice as wide as the layer
            continue
        for (th, xb) in _B_TILES[pgw * a]:
            grid = -(-Hout // th) * -(-Wout // (16 * xb)) * -(-pc.cout_eff // (16 * mb)) * B
            rounds = -(-grid // 256)
            cost = rounds * (pgw * a) * mb
            cands.append((cost, ks if pgw * cgw == 4 else 2, -mb, -th * xb, (xb, th, mb, pgw, cgw, ks)))
    cands.sort()
    cin8 = -(-pc.cin // 8) * 8
    cks = [c for c in ((128, 64, 32, 16, 8) if taps == 1 else (32, 16, 8)) if c <= max(8, cin8)]
    if not cks or cks[0] < min(cin8, 32):
        cks.insert(0, min(cin8, 32))
    out = []
    for _, _, _, _, (xb, th, mb, pgw, cgw, ks) in cands:
        for ck in cks:
            out.append((xb, th, ck, mb, 2, pgw, cgw, terms, ks))
    return out


class SplitTensor:
    """An activation in the split-bf16 conv-input form (include/codd_hip.h, codd_split_bf16): made once by
    ``split_input`` and consumed by any number of convolutions of that activation (same spatial size, padding up to
    the border, channel slices at multiples of 8)."""
    __slots__ = ("buf", "B", "C", "H", "W", "bt", "bl", "hp", "wp", "c8", "terms")

    def __init__(self, buf, B, C, H, W, bt, bl, hp, wp, c8, terms):
        self.buf, self.B, self.C, self.H, self.W = buf, B, C, H, W
        self.bt, self.bl, self.hp, self.wp, self.c8, self.terms = bt, bl, hp, wp, c8, terms


def _split_rows(H):
    """Image rows of a shared split tensor: enough for the row overhang of EVERY instantiated tile height
    (4, 5, 6, 8, 9, 10, 12, 16 rows: ceil(H / th) * th <= H + th - 1)."""
    return max(-(-H // th) * th for th in (4, 5, 6, 8, 9, 10, 12, 16))


def split_input(x, x2=None, border=0, c8=None, hp=None, wp=None, out=None):
    """codd_split_bf16 of (x | x2) with a zero border of ``border`` pixels (int or (top, left)); hp / wp default to
    the size that serves every instantiated tile (4 ... 16 rows, 32-pixel columns) of a stride-1 'same' convolution
    whose padding does not exceed the border.  Returns None outside the split / bf16 precision modes."""
    terms = _TERMS.get(CONV_PRECISION, 0)
    if not terms:
        return None
    lib = _abi.load()
    xs = _as_slice(x)
    _require_gpu(xs.buf)
    B, C0, H, W = xs.shape
    C1 = 0 if x2 is None else _as_slice(x2).c
    if out is not None:  # re-layout into an existing (persistent) tensor of the same geometry
        _abi.check(lib.codd_split_bf16(_view(xs), C0, _view(x2), C1, B, H, W, out.bt, out.bl, out.c8, out.hp, out.wp,
                                       out.terms, out.buf.data_ptr(), _stream()), "codd_split_bf16")
        return out
    bt, bl = (border, border) if isinstance(border, int) else border
    c8 = -(-(C0 + C1) // 32) * 4 if c8 is None else c8  # whole 32-channel chunks
    hp = 2 * bt + _split_rows(H) if hp is None else hp
    wp = 2 * bl + -(-W // 32) * 32 if wp is None else wp
    buf = torch.empty(lib.codd_split_bf16_bytes(B, c8, hp, wp, terms), device=xs.buf.device, dtype=torch.uint8)
    _abi.check(lib.codd_split_bf16(_view(xs), C0, _view(x2), C1, B, H, W, bt, bl, c8, hp, wp, terms, buf.data_ptr(),
                                   _stream()), "codd_split_bf16")
    return SplitTensor(buf, B, C0 + C1, H, W, bt, bl, hp, wp, c8, terms)


_SPLIT_BUFFERS = {}


def split_buffer(key, B, C, H, W, border=0, device=None):
    """A PERSISTENT, zero-initialised SplitTensor for activations that a convolution writes directly in split-bf16
    form (conv2d(..., xs_out=...)): the producer only writes the image interior, so the zero border (and the zero
    channel padding) made here once stays valid for every later frame.  One buffer per call site (``key``) and shape;
    returns None outside the split / bf16 precision modes."""
    terms = _TERMS.get(CONV_PRECISION, 0)
    if not terms:
        return None
    lib = _abi.load()
    bt, bl = (border, border) if isinstance(border, int) else border
    k = (key, B, C, H, W, bt, bl, terms, str(device))
    st = _SPLIT_BUFFERS.get(k)
    if st is None:
        c8 = -(-C // 32) * 4
        hp, wp = 2 * bt + _split_rows(H), 2 * bl + -(-W // 32) * 32
        buf = torch.zeros(lib.codd_split_bf16_bytes(B, c8, hp, wp, terms), device=device, dtype=torch.uint8)
        st = _SPLIT_BUFFERS[k] = SplitTensor(buf, B, C, H, W, bt, bl, hp, wp, c8, terms)
    return st


def conv2d(x, pc, x2=None, stride=1, pad=0, dil=1, act="none", res1=None, res2=None, post=None,
           out=None, pad_tl=None, out_hw=None, xs=None, xs_coff=0, xs_out=None, xs_out_coff=0):
    """act(conv(cat[x, x2]) + bias + res1 + res2) + post  ->  out (tensor or Slice).
    ``xs``: a SplitTensor of the input made by split_input (channels [xs_coff, xs_coff + Cin) of it): used instead of
    a private re-layout when this layer runs on the split-bf16 kernel and the tensor fits its tiles.  ``x`` may be
    None when the input only exists in split form (it was written by a producer's ``xs_out``).
    ``xs_out``: write the result as split-bf16 records into this (persistent, zero-bordered: split_buffer) tensor at
    channel offset ``xs_out_coff`` instead of an fp32 tensor, and return it -- the next convolution then needs no
    re-layout pass.  Both options pin the layer to the split-bf16 kernel."""
    lib = _abi.load()
    force_split = (x is None) or (xs_out is not None)
    if x is None:
        assert xs is not None and x2 is None and xs_coff % 8 == 0
        _require_gpu(xs.buf)
        B, C0, Hin, Win = xs.B, pc.cin, xs.H, xs.W
        xsl = None
        dev_ = xs.buf.device
    else:
        xsl = _as_slice(x)
        _require_gpu(xsl.buf)
        B, C0, Hin, Win = xsl.shape
        dev_ = xsl.buf.device
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
    terms = _TERMS.get(CONV_PRECISION, 0)
    if force_split and not terms:
        raise _abi.CoddHipError("split-form convolution inputs / outputs need the 'split' or 'bf16' conv precision")
    if xs_out is None:
        if out is None:
            out = torch.empty(B, pc.cout, Hout * up, Wout * up, device=dev_, dtype=torch.float32)
        os_ = _as_slice(out)
        assert os_.shape == (B, pc.cout, Hout * up, Wout * up), (os_.shape, (B, pc.cout, Hout * up, Wout * up))
    else:
        assert (xs_out.B, xs_out.H, xs_out.W) == (B, Hout, Wout) and xs_out.terms == terms and xs_out_coff % 8 == 0
        assert xs_out_coff + pc.cout <= 8 * xs_out.c8 and not pc.deconv and res1 is None and res2 is None and post is None
        os_ = None
    key = (Hout, Wout, B, sy, sx, dy, dx, pl, C1 > 0, terms) + (("split",) if force_split else ()) + (
        ("co",) if _CORESIDENT and terms else ())
    p = ConvParams()
    p.in0 = _view(xsl)
    p.in1 = _view(x2)
    p.C0, p.C1, p.B, p.Hin, p.Win = C0, C1, B, Hin, Win
    p.bias = None if pc.bias is None else pc.bias.data_ptr()
    p.res1, p.res2, p.post = _view(res1), _view(res2), _view(post)
    if os_ is not None:
        p.out, p.out_ctot, p.out_coff = os_.buf.data_ptr(), os_.buf.shape[1], os_.coff
    p.Cout, p.Hout, p.Wout = pc.cout, Hout, Wout
    p.kh, p.kw, p.sy, p.sx, p.pad_t, p.pad_l, p.dil_y, p.dil_x = pc.kh, pc.kw, sy, sx, pt, pl, dy, dx
    p.act = ACT[act]
    p.store_mode = 1 if pc.deconv else 0
    p.terms = terms

    cfg = pc.tuned.get(key)
    if cfg is None:
        sig = ("b%d|" % terms if terms else "") + "%d,%d,%d,%d,%d,%d|" % (
            pc.cout_eff, pc.cin, pc.kh, pc.kw, pc.mb, int(pc.deconv)) + ",".join(str(int(v)) for v in key[:9]) + (
                "|split" if force_split else "") + ("|co" if _CORESIDENT and terms else "")
        co = _small_footprint if _CORESIDENT else (lambda c: c)
        capturing = torch.cuda.is_current_stream_capturing()
        if _AUTOTUNE and sig in TUNE_DB and _db_cfg_ok(lib, p, TUNE_DB[sig], sig):
            cfg = pc.tuned[key] = tuple(TUNE_DB[sig])  # same layer signature already timed (this process or a loaded file)
        elif force_split:
            # pinned to the split-bf16 kernel: the configuration this layer was tuned to in its plain form if that is
            # a split one, else the first candidate the library accepts
            cands = co([c for c in _bf16_candidates(pc, Hout, Wout, B, pc.kh * pc.kw, terms) if _cfg_ok(lib, p, c)])
            if not cands:
                raise _abi.CoddHipError("no split-bf16 launch configuration for conv %dx%d %d->%d" % (
                    pc.kh, pc.kw, pc.cin, pc.cout))
            if _AUTOTUNE and not capturing:  # time the candidates on the real split input / output tensors
                if x is None:
                    p.xs, p.xs_c8, p.xs_hp, p.xs_wp = xs.buf.data_ptr(), xs.c8, xs.hp, xs.wp
                    p.xs_bt, p.xs_bl, p.xs_o8 = xs.bt, xs.bl, xs_coff // 8
                else:
                    keep = _make_split(lib, p, xsl, x2, cands)  # noqa: F841
                if xs_out is not None:
                    p.xso, p.xso_c8, p.xso_hp, p.xso_wp = xs_out.buf.data_ptr(), xs_out.c8, xs_out.hp, xs_out.wp
                    p.xso_bt, p.xso_bl, p.xso_o8, p.xso_terms = xs_out.bt, xs_out.bl, xs_out_coff // 8, xs_out.terms
                cfg, _ = _autotune_b(lib, p, pc, cands, None, None)
                TUNE_DB[sig] = cfg
            else:
                plain = TUNE_DB.get(sig[:-6]) if _AUTOTUNE and not _CORESIDENT else None
                cfg = tuple(plain) if plain is not None and len(plain) > 4 and plain[4] == 2 else cands[0]
            pc.tuned[key] = cfg
        elif terms:
            cands = co([c for c in _bf16_candidates(pc, Hout, Wout, B, pc.kh * pc.kw, terms) if _cfg_ok(lib, p, c)])
            if _AUTOTUNE and not capturing:
                # measure: best split-bf16 configuration (incl. its re-layout pass) against the best exact-fp32 one --
                # small or large-map layers can be faster (and are more exact) on the fp32 kernels
                h32 = _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx, pl)
                f32cfg, t32 = _autotune(lib, p, pc, h32 + (pc.mb, 0), with_time=True)
                p.terms = terms
                bcfg, tb = (None, float("inf"))
                if cands:
                    keep = _make_split(lib, p, xsl, x2, cands)  # noqa: F841 one split input serving every candidate
                    bcfg, tb = _autotune_b(lib, p, pc, cands, xsl, x2)
                cfg = bcfg if tb < t32 else f32cfg
                AUTOTUNE_LOG.append(("choice %dx%d %d->%d out %dx%d" % (pc.kh, pc.kw, pc.cin, pc.cout, Hout, Wout),
                                     f32cfg, t32 * 1e3, cfg, min(tb, t32) * 1e3))
                pc.tuned[key] = TUNE_DB[sig] = cfg
            else:
                cfg = cands[0] if cands else _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx, pl) + (pc.mb, 0)
                if not _AUTOTUNE:
                    pc.tuned[key] = cfg  # (capturing with autotune on: heuristic for this launch, tune later)
        else:
            cfg = _conv_cfg(pc, Hout, Wout, B, sy, sx, dy, dx, pl)
            if _AUTOTUNE and not capturing:
                cfg = TUNE_DB[sig] = _autotune(lib, p, pc, cfg + (pc.mb, 0))
            if not _AUTOTUNE or not capturing:
                pc.tuned[key] = cfg

    if len(cfg) > 4 and cfg[4] == 2:  # split-bf16 / bf16 kernel
        keep = None
        if xs_out is not None:
            p.xso, p.xso_c8, p.xso_hp, p.xso_wp = xs_out.buf.data_ptr(), xs_out.c8, xs_out.hp, xs_out.wp
            p.xso_bt, p.xso_bl, p.xso_o8, p.xso_terms = xs_out.bt, xs_out.bl, xs_out_coff // 8, xs_out.terms
            out = xs_out
        if (xs is not None and xs.terms == terms and (xs.B, xs.H, xs.W) == (B, Hin, Win) and xs_coff % 8 == 0 and
                xs_coff + pc.cin <= max(xs.C, 8 * xs.c8 if x is None else 0)):
            p.xs, p.xs_c8, p.xs_hp, p.xs_wp = xs.buf.data_ptr(), xs.c8, xs.hp, xs.wp
            p.xs_bt, p.xs_bl, p.xs_o8 = xs.bt, xs.bl, xs_coff // 8
            _set_cfg(p, pc, cfg)
            rc = _launch_conv(lib, p, _stream())
            if rc == 0:
                return out
            if rc != -1 or x is None:  # -1: the shared tensor does not fit this configuration's tiles -> private re-layout below
                _abi.check(rc, "codd_conv2d")
        keep = _make_split(lib, p, xsl, x2, [cfg])  # noqa: F841 (alive until the launch below is enqueued)
        _set_cfg(p, pc, cfg)
        _abi.check(_launch_conv(lib, p, _stream()), "codd_conv2d")
        return out
    npb, nw, ck = cfg[:3]
    mb = cfg[3] if len(cfg) > 3 else pc.mb
    layout = cfg[4] if len(cfg) > 4 else 0
    p.terms = 0
    p.wpacked = pc.packed(ck, mb, layout).data_ptr()
    p.mb, p.npb, p.nw, p.ck, p.layout = mb, npb, nw, ck, layout
    heur = (key, (Hout, Wout, B, sy, sx, dy, dx, pl))
    if _DEFERRED is not None and layout == 1 and npb == 1 and nw == 4 and mb == 1:
        # inside ``with deferred_convs():`` -- recorded, launched on exit together with its independent neighbours
        _DEFERRED.append((ConvParams.from_buffer_copy(p), (x, x2, res1, res2, post, out), pc, heur))
        return out
    _launch_fp32(lib, p, pc, heur)
    return out


def _launch_fp32(lib, p, pc, heur):
    """Launch one exact-fp32 convolution; a tuned / loaded configuration this build does not support (rc -2, e.g. a tune
    db from another version) falls back to the heuristic configuration for this launch shape, loudly.  Shared by conv2d
    and by deferred_convs' single-launch path."""
    rc = _launch_conv(lib, p, _stream())
    if rc == -2:
        import warnings
        key, geom = heur
        warnings.warn("codd_amd: launch configuration %s rejected for conv %dx%d %d->%d, using the heuristic" % (
            (p.npb, p.nw, p.ck, p.mb, p.layout), pc.kh, pc.kw, pc.cin, pc.cout))
        npb, nw, ck = pc.tuned[key] = _conv_cfg(pc, *geom)
        p.wpacked = pc.packed(ck, pc.mb, 0).data_ptr()
        p.mb, p.npb, p.nw, p.ck, p.layout = pc.mb, npb, nw, ck, 0
        rc = _launch_conv(lib, p, _stream())
    _abi.check(rc, "codd_conv2d")


def _db_cfg_ok(lib, p, c, sig):
    """A launch configuration loaded from a tune db must be one THIS build accepts (a db written by another version
    may name tiles that no longer exist): split-bf16 entries are probed with codd_conv2d_check (fp32 entries are
    validated at launch, rc -2 -> heuristic); a rejected entry is dropped with a warning and the layer is tuned /
    given the heuristic as if the db had no entry."""
    if len(c) > 4 and c[4] == 2:
        if not p.terms or (len(c) > 7 and c[7] != p.terms) or not _cfg_ok(lib, p, c):
            import warnings
            warnings.warn("codd_amd: tune-db entry %s = %s is not a valid split-bf16 configuration of this build; ignored" % (
                sig, tuple(c)))
            del TUNE_DB[sig]
            return False
    return True


MULTI_CONV = True  # (A/B switch of conv2d_multi)
MULTI_DEEP_FIRST = True  # (A/B: job order inside a multi-job launch)
MULTI_MB = 1  # 16-channel blocks per workgroup of a multi-job launch (1 | 2)

_DEFERRED = None


class deferred_convs:
    """``with ops.deferred_convs(): ...`` -- exact-fp32 convolutions issued inside the block whose launch configuration is
    the multi-job class (quad layout, 4 x 16 tiles, 16 channels per workgroup: csrc/conv_quad_kernel.h) are RECORDED with
    the parameters the single launch would have used and launched on exit, up to four per codd_conv2d_multi launch: the
    same kernel body on the same parameters, i.e. bit-identical results, and one kernel node instead of up to four (every
    node of the frame graph costs ~3.8 us of wall clock, DESIGN.md finding 43).  Convolutions of other classes launch at
    once.  The caller guarantees that the convolutions inside one block do not read one another's results."""

    def __enter__(self):
        global _DEFERRED
        self.prev, _DEFERRED = _DEFERRED, []
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        items, _DEFERRED = _DEFERRED, self.prev
        if exc[0] is not None or not items:
            return False
        lib = _abi.load()
        for k0 in range(0, len(items), 4):
            chunk = items[k0:k0 + 4]
            if MULTI_DEEP_FIRST:
                chunk = sorted(chunk, key=lambda c: -((c[0].C0 + c[0].C1) * c[0].kh * c[0].kw))
            rc = -2
            if len(chunk) > 1 and MULTI_CONV:
                arr = (ConvParams * len(chunk))(*[c[0] for c in chunk])
                rc = _launch_conv_multi(lib, arr, len(chunk), _stream())
                if rc not in (0, -2):
                    _abi.check(rc, "codd_conv2d_multi (deferred)")
            if rc != 0:  # one job, or a multi-job launch this build rejects: single launches with conv2d's own fallback
                for pp, _, pc, heur in chunk:
                    _launch_fp32(lib, pp, pc, heur)
        return False


def _launch_conv_multi(lib, params, n, stream):
    """Single choke point of the multi-job conv launches (bench.py wraps it with HIP events)."""
    return lib.codd_conv2d_multi(params, n, stream)


def conv2d_multi(jobs):
    """Independent stride-1/2 convolutions -> list of outputs.  ``jobs``: dicts(x, pc[, stride, pad, act, res1]).  Under the
    exact-fp32 precision the jobs whose rows are 16-byte aligned are launched up to four at a time by ONE
    codd_conv2d_multi (quad layout, 4 x 16 tiles, 16 output channels per workgroup: the launch-bound small layers of
    HRNet's branches); everything else goes through conv2d one by one."""
    lib = _abi.load()
    outs = [None] * len(jobs)
    group = []
    if MULTI_CONV and CONV_PRECISION == "fp32" and len(jobs) > 1:
        for idx, j in enumerate(jobs):
            x, pc = j["x"], j["pc"]
            if (isinstance(x, torch.Tensor) and x.shape[3] % 4 == 0 and x.data_ptr() % 16 == 0 and not pc.deconv
                    and pc.cin >= 16 and j.get("stride", 1) in (1, 2)):
                group.append(idx)
    for k0 in range(0, len(group), 4):
        ids = group[k0:k0 + 4]
        if len(ids) < 2:
            break
        if MULTI_DEEP_FIRST:
            # workgroups are dispatched in job order: the job with the longest per-workgroup chain (most chunks x taps)
            # goes first, so that its chain runs beside the wide, shallow jobs instead of after them
            ids = sorted(ids, key=lambda i_: -(jobs[i_]["pc"].cin * jobs[i_]["pc"].kh * jobs[i_]["pc"].kw))
        arr = (ConvParams * len(ids))()
        keep = []
        for slot, idx in enumerate(ids):
            j = jobs[idx]
            x, pc = j["x"], j["pc"]
            B, C0, Hin, Win = x.shape
            st, pad = j.get("stride", 1), j.get("pad", 0)
            Hout = (Hin + 2 * pad - (pc.kh - 1) - 1) // st + 1
            Wout = (Win + 2 * pad - (pc.kw - 1) - 1) // st + 1
            out = torch.empty(B, pc.cout, Hout, Wout, device=x.device, dtype=torch.float32)
            ck = 32 if pc.cin > 16 else 16
            p = arr[slot]
            p.in0, p.C0, p.C1, p.B, p.Hin, p.Win = _view(x), C0, 0, B, Hin, Win
            p.bias = None if pc.bias is None else pc.bias.data_ptr()
            p.res1 = _view(j.get("res1"))
            p.out, p.out_ctot, p.out_coff = out.data_ptr(), pc.cout, 0
            p.Cout, p.Hout, p.Wout = pc.cout, Hout, Wout
            p.kh, p.kw, p.sy, p.sx, p.pad_t, p.pad_l, p.dil_y, p.dil_x = pc.kh, pc.kw, st, st, pad, pad, 1, 1
            p.act = ACT[j.get("act", "none")]
            p.mb, p.npb, p.nw, p.ck, p.layout = MULTI_MB, 1, 4, ck, 1
            p.wpacked = pc.packed(ck, MULTI_MB, 1).data_ptr()
            keep.append(out)
        rc = _launch_conv_multi(lib, arr, len(ids), _stream())
        if rc == 0:
            for idx, out in zip(ids, keep):
                outs[idx] = out
        elif rc != -2:
            _abi.check(rc, "codd_conv2d_multi")
    for idx, j in enumerate(jobs):
        if outs[idx] is None:
            outs[idx] = conv2d(j["x"], j["pc"], stride=j.get("stride", 1), pad=j.get("pad", 0), act=j.get("act", "none"),
                               res1=j.get("res1"))
    return outs


class C4Tensor:
    """A channel-quad fp32 activation [B][C/4][H*W][4] (include/codd_hip.h, codd_conv_params.gate): the private layout of
    the ConvGRU's gate operands -- a lane of the split-bf16 kernel's record-form accumulator holds 4 consecutive
    channels of one pixel, i.e. one 16-byte access here."""
    __slots__ = ("buf", "B", "C", "H", "W")

    def __init__(self, buf, B, C, H, W):
        self.buf, self.B, self.C, self.H, self.W = buf, B, C, H, W

    def view(self, coff=0):
        return _abi.View(self.buf.data_ptr(), self.C, coff)

    def nchw(self):
        """-> a new fp32 NCHW tensor (tests / the hand-over to code outside the update block)."""
        return self.buf.view(self.B, self.C // 4, self.H * self.W, 4).permute(0, 1, 3, 2).reshape(
            self.B, self.C, self.H, self.W).contiguous()


_C4_BUFFERS = {}


def c4_buffer(key, B, C, H, W, device):
    """A persistent C4Tensor per call site (``key``) and shape."""
    assert C % 4 == 0
    k = (key, B, C, H, W, str(device))
    t = _C4_BUFFERS.get(k)
    if t is None:
        t = _C4_BUFFERS[k] = C4Tensor(torch.zeros(B * C * H * W, device=device, dtype=torch.float32), B, C, H, W)
    return t


def to_c4(x, out):
    """fp32 NCHW tensor -> the C4Tensor ``out`` (one strided copy)."""
    B, C, H, W = x.shape
    assert (out.B, out.C, out.H, out.W) == (B, C, H, W)
    out.buf.view(B, C // 4, H * W, 4).copy_(x.reshape(B, C // 4, 4, H * W).permute(0, 1, 3, 2))
    return out


def conv_gate(pc, xs, gate, pad=0, dil=1, dil2=0, out=None, out_coff=0, res1=None, res2=None, post=None, xs_out=None,
              xs_coff=0):
    """A stride-1 'same' convolution of the split tensor ``xs`` on the split-bf16 kernel with a ConvGRU gate epilogue
    (include/codd_hip.h, codd_conv_params.gate / dil2): ``gate`` 1 = plain result into the C4Tensor ``out``;
    2 = z | r*h | q-input of the merged gate-input convolution; 3 = the hidden-state update.  ``dil2``: the weights
    hold two tap sets ([cout, cin, 2*k, k]: dilation dil2 rows first, then dilation dil), both over one input tile.
    out / res1 / res2 / post are C4Tensors, xs_out a SplitTensor."""
    lib = _abi.load()
    terms = _TERMS.get(CONV_PRECISION, 0)
    if not terms:
        raise _abi.CoddHipError("gate-epilogue convolutions need the 'split' or 'bf16' conv precision")
    _require_gpu(xs.buf)
    B, H, W = xs.B, xs.H, xs.W
    p = ConvParams()
    p.C0, p.C1, p.B, p.Hin, p.Win = pc.cin, 0, B, H, W
    p.bias = None if pc.bias is None else pc.bias.data_ptr()
    p.Cout, p.Hout, p.Wout = pc.cout, H, W
    p.kh, p.kw, p.sy, p.sx, p.pad_t, p.pad_l, p.dil_y, p.dil_x = pc.kh, pc.kw, 1, 1, pad, pad, dil, dil
    p.dil2, p.gate, p.terms, p.layout = dil2, gate, terms, 2
    p.out, p.out_ctot, p.out_coff = out.buf.data_ptr(), out.C, out_coff
    for name, t in (("res1", res1), ("res2", res2), ("post", post)):
        if t is not None:
            setattr(p, name, t.view())
    p.xs, p.xs_c8, p.xs_hp, p.xs_wp = xs.buf.data_ptr(), xs.c8, xs.hp, xs.wp
    p.xs_bt, p.xs_bl, p.xs_o8 = xs.bt, xs.bl, xs_coff // 8
    if xs_out is not None:
        p.xso, p.xso_c8, p.xso_hp, p.xso_wp = xs_out.buf.data_ptr(), xs_out.c8, xs_out.hp, xs_out.wp
        p.xso_bt, p.xso_bl, p.xso_o8, p.xso_terms = xs_out.bt, xs_out.bl, 0, xs_out.terms
    key = ("gate", gate, H, W, B, pad, dil, dil2, terms)
    cfg = pc.tuned.get(key)
    if cfg is None:
        sig = "g%d,b%d|%d,%d,%d,%d|%d,%d,%d,%d,%d,%d" % (gate, terms, pc.cout, pc.cin, pc.kh, pc.kw, H, W, B, pad, dil, dil2)
        capturing = torch.cuda.is_current_stream_capturing()
        if _AUTOTUNE and sig in TUNE_DB and _db_cfg_ok(lib, p, TUNE_DB[sig], sig):
            cfg = tuple(TUNE_DB[sig])
        else:
            cands = [c for c in _bf16_candidates(pc, H, W, B, pc.kh * pc.kw, terms) if _cfg_ok(lib, p, c)]
            if gate == 1:
                # the z|r convolution runs BESIDE the VALU-only Gauss-Newton builder (BasicUpdateBlock.zr_convs): a
                # stand-alone timing would pick 12-wave workgroups, whose ~450 registers per SIMD leave the builder
                # no room on the same CU; 8-wave workgroups (2 waves per SIMD, ~106 registers each) co-reside with
                # two builder waves per SIMD.  Measured in the frame: 93.2-93.5 against 91.7-92.1 frames/s
                # (tools/zr_cfg_sweep.sh); the isolated timing then only chooses among those.
                cands = _small_footprint(cands)
            if not cands:
                raise _abi.CoddHipError("no split-bf16 launch configuration for gate conv %dx%d %d->%d" % (
                    pc.kh, pc.kw, pc.cin, pc.cout))
            if _AUTOTUNE and not capturing:
                cfg, _ = _autotune_b(lib, p, pc, cands, None, None)  # (times into a scratch ``out``: in-place gates are safe)
                TUNE_DB[sig] = cfg
            else:
                cfg = cands[0]
        if not capturing or not _AUTOTUNE:
            pc.tuned[key] = cfg
    _set_cfg(p, pc, cfg)
    _abi.check(_launch_conv(lib, p, _stream()), "codd_conv2d (gate %d)" % gate)
    return out


def _cfg_ok(lib, p, c):
    """Does the library accept split-bf16 configuration ``c`` for the layer described by ``p``? (nothing is launched)"""
    p.npb, p.nw, p.ck, p.mb, p.layout, p.pgw, p.cgw = c[:7]
    p.ksplit = c[8] if len(c) > 8 else 1
    return lib.codd_conv2d_check(C.byref(p)) == 0


def _set_cfg(p, pc, c):
    """Fill the launch-configuration fields of ``p`` from a split-bf16 configuration tuple."""
    xb, th, ck, mb, _, pgw, cgw = c[:7]
    p.wpacked = pc.packed(ck, mb, 20 + p.terms).data_ptr()
    p.mb, p.npb, p.nw, p.ck, p.layout, p.pgw, p.cgw = mb, xb, th, ck, 2, pgw, cgw
    p.ksplit = c[8] if len(c) > 8 else 1


def _split_dims(p, cands):
    """(c8, hp, wp) of a split-bf16 input that serves every configuration in ``cands`` for the conv described by
    ``p`` (include/codd_hip.h, codd_split_bf16): borders (pad_t, pad_l) + the largest tile overhang."""
    cin = p.C0 + p.C1
    c8 = hp = wp = 0
    for c in cands:
        xb, th, ck = c[0], c[1], c[2]
        c8 = max(c8, -(-cin // ck) * (ck // 8))
        hp = max(hp, p.pad_t + p.Hin, (-(-p.Hout // th) * th - 1) * p.sy + (p.kh - 1) * p.dil_y + 1)
        wp = max(wp, p.pad_l + p.Win, (-(-p.Wout // (16 * xb)) * 16 * xb - 1) * p.sx + (p.kw - 1) * p.dil_x + 1)
    return c8, hp, wp


def _make_split(lib, p, xs, x2, cands):
    """Run codd_split_bf16 for the conv input (x | x2) and attach the result to ``p``; returns the buffer (the
    caller keeps it alive until the conv is enqueued -- the caching allocator is stream-ordered)."""
    c8, hp, wp = _split_dims(p, cands)
    n = lib.codd_split_bf16_bytes(p.B, c8, hp, wp, p.terms)
    buf = torch.empty(n, device=xs.buf.device, dtype=torch.uint8)
    _abi.check(lib.codd_split_bf16(_view(xs), p.C0, _view(x2), p.C1, p.B, p.Hin, p.Win, p.pad_t, p.pad_l, c8, hp, wp,
                                   p.terms, buf.data_ptr(), _stream()), "codd_split_bf16")
    p.xs, p.xs_c8, p.xs_hp, p.xs_wp = buf.data_ptr(), c8, hp, wp
    p.xs_bt, p.xs_bl, p.xs_o8 = p.pad_t, p.pad_l, 0
    return buf


def _autotune_b(lib, p, pc, cands, xsl, x2):
    """Time every split-bf16 candidate the library accepts (scratch output, see _autotune) and keep the fastest;
    returns (configuration, ms) where the time includes the layer's own re-layout pass (codd_split_bf16)."""
    stream = _stream()
    real_out = p.out
    scratch_out = torch.empty(p.B * p.out_ctot * (4 if p.store_mode else 1) * p.Hout * p.Wout, device=pc._w.device,
                              dtype=torch.float32)
    p.out = scratch_out.data_ptr()
    torch.cuda.synchronize()
    best, best_t, first = None, float("inf"), None
    for c in cands[:1] + cands:  # first candidate twice: the first pass warms clocks / caches
        _set_cfg(p, pc, c)
        if _launch_conv(lib, p, stream) != 0:
            continue
        t = float("inf")
        for _rep in range(2):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                _launch_conv(lib, p, stream)
            e.record()
            e.synchronize()
            t = min(t, s.elapsed_time(e) / 3.0)
        if first is None:
            first = (c, t)
            continue
        if AUTOTUNE_TRACE is not None:
            xb_, th_, mb_ = c[0], c[1], c[3]
            AUTOTUNE_TRACE.append(("b%d g%d k%dx%d d%d/%d %d->%d out %dx%d" % (p.terms, p.gate, pc.kh, pc.kw, p.dil_y, p.dil2, pc.cin, pc.cout, p.Hout, p.Wout), c,
                                   -(-p.Hout // th_) * -(-p.Wout // (16 * xb_)) * -(-pc.cout_eff // (16 * mb_)) * p.B, t * 1e3))
        if t < best_t * 0.97 or best is None:
            best, best_t = c, t
    p.out = real_out
    if best is None:
        raise _abi.CoddHipError("no split-bf16 launch configuration for conv %dx%d %d->%d" % (pc.kh, pc.kw, pc.cin, pc.cout))
    used = {pc._pack_key(c) for c in pc.tuned.values()} | {pc._pack_key(best)}
    for k in [k for k in pc._packs if k not in used]:
        del pc._packs[k]
    # the re-layout pass of the winner (private tensor of exactly its size); none when the input exists in split form
    t_split = 0.0
    if xsl is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _make_split(lib, p, xsl, x2, [best])
        s.record()
        for _ in range(3):
            keep = _make_split(lib, p, xsl, x2, [best])  # noqa: F841
        e.record()
        e.synchronize()
        t_split = s.elapsed_time(e) / 3.0
    AUTOTUNE_LOG.append(("b%d %dx%d k%dx%d %d->%d out %dx%d" % (p.terms, p.sy, p.sx, pc.kh, pc.kw, pc.cin, pc.cout, p.Hout,
                                                                p.Wout), first[0], first[1] * 1e3, best, best_t * 1e3))
    return best, best_t + t_split


_AUTOTUNE = False  # enable_autotune()
AUTOTUNE_LOG = []  # (layer description, heuristic cfg, us, chosen cfg, us) of every tuned launch shape
AUTOTUNE_TRACE = None  # dev (tools/sweep_update_block.py): a list collects (layer description, cfg, grid, us) of EVERY candidate timed


TUNE_DB = {}  # layer signature "cout,cin,kh,kw,mb,deconv|launch shape" -> (npb, nw, ck)


def save_tune_db(path):
    import json
    with open(path, "w") as f:
        json.dump({k: list(v) for k, v in TUNE_DB.items()}, f, indent=0, sort_keys=True)


def load_tune_db(path=None):
    """path = None: the db shipped with the library (codd_amd/tuned/mi355x.json, tuned on an MI355X)."""
    import json
    if path is None:
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuned", "mi355x.json")
    TUNE_DB.update({k: tuple(v) for k, v in json.load(open(path)).items()})


def enable_autotune(flag=True, shipped=True):
    """Measure-don't-guess launch configuration: the first (eager, un-captured) launch of every
    (layer, shape) times the heuristic (npb, nw, ck, mb, layout) against every other configuration the kernels
    are instantiated for and keeps the fastest for the rest of the process (bench.py and the CLI turn this on;
    the tests run the deterministic heuristics).  ``shipped``: start from codd_amd/tuned/mi355x.json (the
    configurations found on an MI355X for the 960x576 workload), so that known layer signatures are not timed
    again.  Different chunk depths / layouts change the fp32 summation order, nothing else."""
    global _AUTOTUNE
    _AUTOTUNE = bool(flag)
    if _AUTOTUNE and shipped and not TUNE_DB:
        try:
            load_tune_db()
        except (OSError, ValueError):
            pass


def _autotune(lib, p, pc, default, with_time=False):
    p.terms = 0
    cin_pad = -(-pc.cin // 4) * 4
    cks = sorted({c for c in (8, 12, 16, 24, 32) if c <= cin_pad} | {min(cin_pad, 32)})
    mbs = [m for m in (1, 2, 4) if m == pc.mb or (16 * m <= max(16, -(-pc.cout_eff // 16) * 16) and pc.cout_eff > 16)]
    cands = [default]
    for mb in mbs:
        for npb in (1, 2, 4):
            for nw in ((4, 9, 2, 8) if npb == 1 else (4,)):
                for ck in cks:
                    if (npb, nw, ck, mb, 0) not in cands:
                        cands.append((npb, nw, ck, mb, 0))
    if p.sx <= 2 and cin_pad >= 16:  # quad layout (ds_read_b128 operands): ck 16 / 32
        for mb in mbs:
            for npb, nw in ((1, 4), (2, 4), (4, 4), (1, 9)):
                for ck in (16, 32):
                    if ck <= max(16, cin_pad):
                        cands.append((npb, nw, ck, mb, 1))
    stream = _stream()
    # the timed launches write into a scratch copy of the output buffer: the real one may alias an operand
    # (in-place accumulation "out = conv(x) + out"), which repeated launches would accumulate over and over
    real_out = p.out
    scratch_out = torch.empty(p.B * p.out_ctot * (4 if p.store_mode else 1) * p.Hout * p.Wout, device=pc._w.device,
                              dtype=torch.float32)
    p.out = scratch_out.data_ptr()
    torch.cuda.synchronize()  # nothing else on the device while the candidates are timed
    best, best_t, t_default = default, float("inf"), None
    for (npb, nw, ck, mb, layout) in [default] + cands:  # the heuristic is timed twice (first = warm-up of clocks / caches)
        try:
            p.wpacked = pc.packed(ck, mb, layout).data_ptr()
        except Exception:
            continue
        p.mb, p.npb, p.nw, p.ck, p.layout = mb, npb, nw, ck, layout
        if _launch_conv(lib, p, stream) != 0:  # not instantiated / LDS or staging limits: skip
            continue
        t = float("inf")
        for _rep in range(2):  # best of two bursts of three launches
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                _launch_conv(lib, p, stream)
            e.record()
            e.synchronize()
            t = min(t, s.elapsed_time(e) / 3.0)
        if (npb, nw, ck, mb, layout) == default:
            if t_default is None:
                t_default = t
                continue  # warm-up pass
            t_default = t
        if t < best_t * 0.97 or best_t == float("inf"):  # 3 % hysteresis: earlier (heuristic-first) candidates win ties
            best, best_t = (npb, nw, ck, mb, layout), t
    used = {pc._pack_key(c) for c in pc.tuned.values()} | {pc._pack_key(best)}
    for k in [k for k in pc._packs if k not in used]:
        del pc._packs[k]  # packed-weight variants of the losing candidates
    p.out = real_out
    AUTOTUNE_LOG.append(("%dx%d k%dx%d %d->%d out %dx%d" % (p.sy, p.sx, pc.kh, pc.kw, pc.cin, pc.cout, p.Hout, p.Wout),
                         default, None if t_default is None else t_default * 1e3, best, best_t * 1e3))
    return (best, best_t) if with_time else best


# ---------------------------------------------------------------------------- rolling-window convolutions
# (csrc/conv_roll.hip, include/codd_hip.h codd_conv_roll): one 3x3, a pair of 3x3 (BasicBlock / merge tail) or a
# 1x1 -> 3x3 pair per launch for HITNet's large 16- / 32-channel maps.  `USE_ROLL` (default on), maps of at
# least ROLL_MIN_PIXELS pixels (below that a 64-column strip grid cannot fill the chip and the tile kernels win).
USE_ROLL = True
ROLL_MIN_PIXELS = 2 * 288 * 480
ROLL_RH = 0  # dev override of the rows per workgroup


ROLL_C32 = False  # (dev: the 32-channel instantiations lose to the tile kernels)


def use_roll(C, B, Cin_unused, H, W):
    """Rolling-window launch for a C-channel stride-1 layer (pair) on a [B, *, H, W] map?  Measured on MI355X
    (tools/time_roll.py): 16 channels at >= 288x480x2 pixels 1.4-1.8x the tile kernels; the 32-channel class has too
    few 64-column strips on HITNet's half-resolution maps to fill 256 CUs and stays on the tile kernels.  The kernel is
    exact fp32: under the plain bf16 mode (bf16 operands EVERYWHERE, BASELINE.json configs[4]) the layers stay on the bf16
    tile kernels so that the mode's dtype description holds."""
    return (USE_ROLL and CONV_PRECISION in ("split", "split16", "fp32") and (C == 16 or (C == 32 and ROLL_C32))
            and B * H * W >= ROLL_MIN_PIXELS)


class PackedRoll:
    """Weights of one codd_conv_roll launch.  ``stages``: one or two dicts(w [cout,cin,k,k], b | None, act); k = 3
    for every stage except that the FIRST of two may be 1x1 (then cin <= 64).  C = the output channels of every stage
    (16 | 32); ``residual``: add the chain input to the last stage (pair of 3x3 only)."""

    def __init__(self, stages, residual=False):
        lib = _abi.load()
        assert len(stages) in (1, 2)
        w0 = stages[0]["w"]
        _require_gpu(w0)
        ks = [int(st["w"].shape[2]) for st in stages]
        self.C = int(stages[-1]["w"].shape[1]) if len(stages) == 2 else int(w0.shape[1])
        self.cin = int(w0.shape[1])
        self.cout = int(stages[-1]["w"].shape[0])
        if len(stages) == 1:
            self.mode = 0
        else:
            self.mode = 1 if ks[0] == 3 else 2
        assert ks[-1] == 3 and (ks[0] in (1, 3)) and self.C in (16, 32), (ks, self.C)
        assert not residual or self.mode == 1
        self.residual = bool(residual)
        self.acts = [ACT[st.get("act", "none")] for st in stages]
        self.w, self.b = [], []
        for st, k in zip(stages, ks):
            w = st["w"].detach().float().contiguous()
            cout, cin = int(w.shape[0]), int(w.shape[1])
            assert cout <= self.C and (k == 1 or cin == self.C), (cout, cin, self.C)
            n = lib.codd_roll_packed_size(self.C, k, cin)
            if n <= 0:
                raise _abi.CoddHipError("conv_roll: unsupported stage %dx%d %d->%d" % (k, k, cin, cout))
            buf = torch.empty(n, device=w.device, dtype=torch.float32)
            _abi.check(lib.codd_roll_pack_weights(w.data_ptr(), buf.data_ptr(), self.C, cout, cin, k, _stream()), "roll_pack")
            self.w.append(buf)
            b = st.get("b")
            if b is not None:
                bb = torch.zeros(self.C, device=w.device, dtype=torch.float32)
                bb[:cout] = b.detach().float()
                b = bb
            self.b.append(b)


def _roll_rh(B, H, W, mode):
    """Output rows per workgroup: the largest row block that still gives every CU ~3 workgroups (the strips of a
    launch are independent; a row block re-reads `lag` rows of its neighbour and idles for the pipeline fill)."""
    if ROLL_RH:
        return ROLL_RH
    stride = 60 if mode == 1 else 62
    strips = -(-W // stride) * B
    want = 3 * 256
    nrb = max(1, -(-want // strips))
    return max(4, -(-H // nrb))


def conv_roll(x, pr, x2=None, out=None, cout_store=None, rh=None):
    """Run the packed rolling-window launch on (x | x2) -> out [B, cout_store, H, W] (tensor or Slice)."""
    lib = _abi.load()
    xs = _as_slice(x)
    _require_gpu(xs.buf)
    B, C0, H, W = xs.shape
    C1 = 0 if x2 is None else _as_slice(x2).c
    assert C0 + C1 == pr.cin, (C0, C1, pr.cin)
    cs = pr.cout if cout_store is None else cout_store
    if out is None:
        out = torch.empty(B, cs, H, W, device=xs.buf.device, dtype=torch.float32)
    os_ = _as_slice(out)
    assert os_.shape == (B, cs, H, W), (os_.shape, (B, cs, H, W))
    p = _abi.RollParams()
    p.in0, p.in1, p.C0, p.C1, p.B, p.H, p.W = _view(xs), _view(x2), C0, C1, B, H, W
    p.C, p.mode = pr.C, pr.mode
    p.wA, p.bA = pr.w[0].data_ptr(), None if pr.b[0] is None else pr.b[0].data_ptr()
    p.actA = pr.acts[0]
    if pr.mode != 0:
        p.wB, p.bB = pr.w[1].data_ptr(), None if pr.b[1] is None else pr.b[1].data_ptr()
        p.actB = pr.acts[1]
    p.residual = int(pr.residual)
    p.out, p.out_ctot, p.out_coff, p.cout_store = os_.buf.data_ptr(), os_.buf.shape[1], os_.coff, cs
    p.rh = _roll_rh(B, H, W, pr.mode) if rh is None else rh
    _abi.check(_launch_roll(lib, p, _stream()), "codd_conv_roll")
    return out


def _launch_roll(lib, p, stream):
    """Single choke point of every rolling-window launch (bench.py wraps it with HIP events)."""
    return lib.codd_conv_roll(C.byref(p), stream)


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


# ----------------------------------------------------------------------------------------- motion
def _f32(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def instnorm(x, relu=True, res=None, res_relu=False, xs_out=None, want_fp32=True):
    """InstanceNorm2d(+ReLU).  Plain form: relu(norm(x) + res).  With ``xs_out`` (a split_buffer tensor of the same
    size) the result is written as split-bf16 records -- the next convolution's input -- by the apply pass itself:
    v = norm(x); relu; [+ res; res_relu]; the fp32 tensor is only written (and returned) when ``want_fp32``."""
    lib = _abi.load()
    _require_gpu(x)
    B, Cc, H, W = x.shape
    stats = _f32(128 * B * Cc, like=x)
    if xs_out is not None:
        assert (xs_out.B, xs_out.H, xs_out.W) == (B, H, W) and Cc % 8 == 0 and Cc <= 8 * xs_out.c8
        y = torch.empty_like(x) if want_fp32 else None
        _abi.check(lib.codd_instnorm_xs(x.data_ptr(), B, Cc, H, W, stats.data_ptr(), _ptr(res), int(relu), int(res_relu),
                                        _ptr(y), _xs_view(xs_out), _stream()), "instnorm_xs")
        return y
    assert not res_relu
    y = torch.empty_like(x)
    _abi.check(lib.codd_instnorm(x.data_ptr(), B, Cc, H * W, stats.data_ptr(),
                                 None if res is None else res.data_ptr(), int(relu), y.data_ptr(), _stream()),
               "instnorm")
    return y


# configurations (xb, th, ck, mb, 2, pgw, cgw, 3, ks) tried, in order, for the split-bf16 all-pairs "convolution"
# (1x1, Cout = h*w source pixels, Cin = 128): 64-channel groups, whole 32-channel chunks
_ALLPAIRS_CFGS = ((2, 8, 32, 4, 2, 4, 2, 3, 1), (1, 16, 32, 4, 2, 4, 2, 3, 1), (1, 16, 32, 4, 2, 4, 1, 3, 1), (1, 12, 32, 4, 2, 4, 2, 3, 1),
                  (1, 12, 32, 4, 2, 4, 1, 3, 1), (1, 10, 32, 4, 2, 2, 2, 3, 1), (2, 8, 32, 4, 2, 4, 1, 3, 1),
                  (1, 8, 32, 2, 2, 4, 1, 3, 1), (1, 8, 16, 2, 2, 4, 1, 3, 1))
_ALLPAIRS_PICK = {}  # (D, hh, ww) -> configuration (timed once per shape when autotuning, else the first accepted)


def allpairs_corr_split(f1, f2):
    """The all-pairs pyramid on the split-bf16 convolution kernel (3 bf16 MFMAs per product, fp32 accumulate): level i =
    a 1x1 "convolution" of the i-times pooled f2 (as split records) whose output channels are the h*w source pixels and
    whose weights are f1^T / 16, re-packed every frame.  Level 0 writes 299 MB at 960x576: on the exact-fp32 MFMA kernel
    it was compute-bound at 0.87 TB/s (345 us); here it is write-bound."""
    lib = _abi.load()
    _require_gpu(f1)
    B, D, h, w = f1.shape
    N = h * w
    lv = [_f32(B, N, (h >> i) * (w >> i), like=f1) for i in range(4)]
    srcs = [f2.contiguous()]
    for i in range(1, 4):
        hh, ww = h >> (i - 1), w >> (i - 1)
        nxt = _f32(B, D, hh >> 1, ww >> 1, like=f1)
        _abi.check(lib.codd_avgpool2(srcs[-1].data_ptr(), B * D, hh, ww, nxt.data_ptr(), _stream()), "avgpool2")
        srcs.append(nxt)
    packs = {}
    tune = _AUTOTUNE and not torch.cuda.is_current_stream_capturing()
    for i, src in enumerate(srcs):
        hh, ww = h >> i, w >> i
        p = ConvParams()
        p.C0, p.C1, p.B, p.Hin, p.Win, p.Cout, p.Hout, p.Wout = D, 0, 1, hh, ww, N, hh, ww
        p.kh = p.kw = p.sy = p.sx = p.dil_y = p.dil_x = 1
        p.terms, p.out_ctot = _split_terms(), N
        key = (D, hh, ww)
        cfg = _ALLPAIRS_PICK.get(key)
        if cfg is None:
            ok = [c for c in _ALLPAIRS_CFGS if _cfg_ok(lib, p, c)]
            if not ok:
                raise _abi.CoddHipError("all-pairs: no split-bf16 configuration for a %dx%d map" % (hh, ww))
            # the timing may only choose among configurations that give THE SAME BITS (tile shape and wave grid do not
            # change a sum, the chunk depth does): a pick that depended on the box would make the pyramid -- and with it
            # every selection downstream -- differ from lease to lease
            if any(c[2] == 32 for c in ok):
                ok = [c for c in ok if c[2] == 32]
            cfg = _allpairs_tune(lib, p, ok, f1[0], N, D) if tune else ok[0]
            if tune:
                _ALLPAIRS_PICK[key] = cfg  # (un-tuned picks are not remembered: a later eager call may still time them)
        xs = split_input_as(src, "split", cands=[cfg], p=p)
        for b in range(B):
            pk = (cfg[3], cfg[2])
            if (pk, b) not in packs:  # weights[co = n1][ci = d] = f1[b, d, n1] / 16 for this (mb, ck)
                nbytes = lib.codd_conv2d_packed_bytes_bf16(N, D, 1, 1, cfg[3], cfg[2], p.terms)
                wp = torch.empty(nbytes, device=f1.device, dtype=torch.uint8)
                _abi.check(lib.codd_conv2d_pack_weights_bf16(f1[b].data_ptr(), wp.data_ptr(), N, D, 1, 1, cfg[3], cfg[2], p.terms,
                                                             1, N, 1.0 / 16.0, _stream()), "pack_weights_bf16")
                packs[(pk, b)] = wp
            p.wpacked = packs[(pk, b)].data_ptr()
            p.out = lv[i][b].data_ptr()
            p.xs = xs.buf.data_ptr() + b * (xs.buf.numel() // B)
            p.xs_c8, p.xs_hp, p.xs_wp, p.xs_bt, p.xs_bl, p.xs_o8 = xs.c8, xs.hp, xs.wp, xs.bt, xs.bl, 0
            p.npb, p.nw, p.ck, p.mb, p.layout, p.pgw, p.cgw = cfg[:7]
            p.ksplit = cfg[8]
            _abi.check(_launch_conv(lib, p, _stream()), "allpairs conv")
    return lv


def split_input_as(x, mode, cands, p):
    """codd_split_bf16 of ``x`` (3 terms) sized for the configurations ``cands`` of the conv described by ``p``."""
    lib = _abi.load()
    c8, hp, wp = _split_dims(p, cands)
    B, C0, H, W = x.shape
    t = p.terms if p.terms in (3, 48) else 3
    buf = torch.empty(lib.codd_split_bf16_bytes(B, c8, hp, wp, t), device=x.device, dtype=torch.uint8)
    _abi.check(lib.codd_split_bf16(_view(x), C0, _view(None), 0, B, H, W, 0, 0, c8, hp, wp, t, buf.data_ptr(), _stream()),
               "codd_split_bf16")
    return SplitTensor(buf, B, C0, H, W, 0, 0, hp, wp, c8, t)


def _allpairs_tune(lib, p, cands, f1b, N, D):
    """Time the accepted all-pairs configurations on this level's shape (own split input / weights per candidate)."""
    best, best_t = cands[0], float("inf")
    src = torch.zeros(1, D, p.Hin, p.Win, device=f1b.device)
    for c in cands:
        xs = split_input_as(src, "split", [c], p)
        nbytes = lib.codd_conv2d_packed_bytes_bf16(N, D, 1, 1, c[3], c[2], p.terms)
        wp = torch.empty(nbytes, device=f1b.device, dtype=torch.uint8)
        lib.codd_conv2d_pack_weights_bf16(f1b.data_ptr(), wp.data_ptr(), N, D, 1, 1, c[3], c[2], p.terms, 1, N, 1.0 / 16.0, _stream())
        p.wpacked, p.xs = wp.data_ptr(), xs.buf.data_ptr()
        p.xs_c8, p.xs_hp, p.xs_wp, p.xs_bt, p.xs_bl, p.xs_o8 = xs.c8, xs.hp, xs.wp, xs.bt, xs.bl, 0
        p.npb, p.nw, p.ck, p.mb, p.layout, p.pgw, p.cgw = c[:7]
        p.ksplit = c[8]
        if lib.codd_conv2d(C.byref(p), _stream()) != 0:
            continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            lib.codd_conv2d(C.byref(p), _stream())
        e.record()
        e.synchronize()
        t = s.elapsed_time(e) / 3
        if t < best_t * 0.97:
            best, best_t = c, t
    AUTOTUNE_LOG.append(("allpairs %dx%d" % (p.Hin, p.Win), cands[0], None, best, best_t * 1e3))
    return best


def allpairs_corr(f1, f2, split=None):
    """-> 4 pyramid levels [B, h*w, (h>>i)*(w>>i)] (reference blocks/corr.py:28-45).  ``split`` (default: the "split"
    conv precision is active): the GEMMs run on the split-bf16 kernel (allpairs_corr_split), else on exact-fp32 MFMA."""
    if split is None:
        split = CONV_PRECISION in _SPLIT_MODES and ALLPAIRS_SPLIT
    if split:
        return allpairs_corr_split(f1, f2)
    lib = _abi.load()
    _require_gpu(f1)
    B, D, h, w = f1.shape
    lv = [_f32(B, h * w, (h >> i) * (w >> i), like=f1) for i in range(4)]
    scratch = _f32(lib.codd_allpairs_corr_scratch(B, D, h, w), like=f1)
    _abi.check(lib.codd_allpairs_corr(f1.data_ptr(), f2.data_ptr(), B, D, h, w, lv[0].data_ptr(), lv[1].data_ptr(),
                                      lv[2].data_ptr(), lv[3].data_ptr(), scratch.data_ptr(), _stream()),
               "allpairs_corr")
    return lv


def corr_lookup(pyr, coords, h, w, out=None):
    """coords [B,h,w,>=2] (x,y,...) -> [B,196,h,w]."""
    lib = _abi.load()
    B = coords.shape[0]
    if out is None:
        out = _f32(B, 196, h, w, like=coords)
    _abi.check(lib.codd_corr_lookup(pyr[0].data_ptr(), pyr[1].data_ptr(), pyr[2].data_ptr(), pyr[3].data_ptr(),
                                    coords.data_ptr(), coords.shape[-1], B, h, w, out.data_ptr(), _stream()),
               "corr_lookup")
    return out


def raft_geometry(T, d1, d2, K8):
    lib = _abi.load()
    B, h, w, _ = T.shape
    xyz = _f32(B, h, w, 3, like=T)
    minfo = _f32(B, 9, h, w, like=T)
    _abi.check(lib.codd_raft_geometry(T.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, h, w, *K8, xyz.data_ptr(),
                                      minfo.data_ptr(), _stream()), "raft_geometry")
    return xyz, minfo


def raft_geometry_lookup(T, d1, d2, K8, pyr, minfo_xs=None, corr_xs=None):
    """raft_geometry + corr_lookup in one launch -> (xyz [B,h,w,3], minfo [B,9,h,w], corr [B,196,h,w]).
    With ``minfo_xs`` / ``corr_xs`` (split_buffer tensors of 9 / 196 channels) the two tensor results are written
    directly as split-bf16 records -- the encoder convolutions' input form -- and (xyz, None, None) is returned."""
    lib = _abi.load()
    B, h, w, _ = T.shape
    if corr_xs is not None:
        xyz = _f32(B, h, w, 3, like=T)
        _abi.check(lib.codd_raft_geometry_lookup_xs(T.data_ptr(), d1.data_ptr(), d2.data_ptr(), pyr[0].data_ptr(),
                                                    pyr[1].data_ptr(), pyr[2].data_ptr(), pyr[3].data_ptr(), B, h, w,
                                                    *K8, xyz.data_ptr(), _xs_view(minfo_xs), _xs_view(corr_xs),
                                                    _stream()), "raft_geometry_lookup_xs")
        return xyz, None, None
    xyz, minfo, out = _f32(B, h, w, 3, like=T), _f32(B, 9, h, w, like=T), _f32(B, 196, h, w, like=T)
    _abi.check(lib.codd_raft_geometry_lookup(T.data_ptr(), d1.data_ptr(), d2.data_ptr(), pyr[0].data_ptr(),
                                             pyr[1].data_ptr(), pyr[2].data_ptr(), pyr[3].data_ptr(), B, h, w, *K8,
                                             xyz.data_ptr(), minfo.data_ptr(), out.data_ptr(), _stream()),
               "raft_geometry_lookup")
    return xyz, minfo, out


def se3_gn_step(T, ae, xyz, delta, weight, d1, K8, radius=32, lm=1e-4, ep=10.0):
    """In-place Gauss-Newton update of the SE3 field T [B,h,w,7]."""
    lib = _abi.load()
    B, h, w, _ = T.shape
    scratch = _f32(lib.codd_se3_gn_scratch(B, h, w, radius), like=T)
    _abi.check(lib.codd_se3_gn_step(T.data_ptr(), ae.data_ptr(), ae.shape[1], xyz.data_ptr(), delta.data_ptr(),
                                    weight.data_ptr(), d1.data_ptr(), B, h, w, *K8, radius, lm, ep,
                                    scratch.data_ptr(), _stream()), "se3_gn_step")
    return T


def se3_gn_step_heads(T, hidden_xs, head_w, head_b, xyz, d1, K8, radius=32, lm=1e-4, ep=10.0):
    """se3_gn_step with the ae / delta / weight 1x1 heads computed inside the record-packing kernel from the hidden
    channels in split form (``hidden_xs``: 768 channels); returns the confidence weights [B,3,h,w]."""
    lib = _abi.load()
    B, h, w, _ = T.shape
    scratch = _f32(lib.codd_se3_gn_scratch(B, h, w, radius), like=T)
    wout = _f32(B, 3, h, w, like=T)
    _abi.check(lib.codd_se3_gn_step_heads(T.data_ptr(), _xs_view(hidden_xs), head_w.data_ptr(), head_b.data_ptr(),
                                          xyz.data_ptr(), d1.data_ptr(), B, h, w, *K8, radius, lm, ep, wout.data_ptr(),
                                          scratch.data_ptr(), _stream()), "se3_gn_step_heads")
    return wout


def cvx_upsample(data, mask, mode):
    """mode 0: [B,h,w,D] -> [B,8h,8w,D]; 1: SE3 field [B,h,w,7]; 2: [B,D,h,w] -> [B,D,8h,8w]."""
    lib = _abi.load()
    if mode == 2:
        B, D, h, w = data.shape
        out = _f32(B, D, 8 * h, 8 * w, like=data)
    else:
        B, h, w, D = data.shape
        out = _f32(B, 8 * h, 8 * w, D, like=data)
    _abi.check(lib.codd_cvx_upsample(data.data_ptr(), mask.data_ptr(), B, h, w, D, mode, out.data_ptr(), _stream()),
               "cvx_upsample")
    return out


def cvx_upsample_se3_weight(T, weight, mask):
    """cvx_upsample(T, mask, 1) and cvx_upsample(weight, mask, 2) in one launch (the mask is read once)."""
    lib = _abi.load()
    B, h, w, _ = T.shape
    To, wo = _f32(B, 8 * h, 8 * w, 7, like=T), _f32(B, 3, 8 * h, 8 * w, like=T)
    _abi.check(lib.codd_cvx_upsample_se3_weight(T.data_ptr(), weight.data_ptr(), mask.data_ptr(), B, h, w, To.data_ptr(),
                                                wo.data_ptr(), _stream()), "cvx_upsample_se3_weight")
    return To, wo


def disp_to_depth(disp, bf):
    lib = _abi.load()
    out = torch.empty_like(disp)
    _abi.check(lib.codd_disp_to_depth(disp.data_ptr(), disp.numel(), float(bf), out.data_ptr(), _stream()),
               "disp_to_depth")
    return out


def subsample(x, oy, ox, step):
    """x[:, oy::step, ox::step] of a contiguous [B, H, W] map as a kernel of this library."""
    lib = _abi.load()
    _require_gpu(x)
    B, H, W = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty(B, -(-(H - oy) // step), -(-(W - ox) // step), device=x.device, dtype=torch.float32)
    _abi.check(lib.codd_subsample(x.data_ptr(), B, H, W, oy, ox, step, out.data_ptr(), _stream()), "subsample")
    return out


def batch_pair(a, b):
    """torch.cat([a, b], 0) of two contiguous fp32 tensors of one shape as ONE copy kernel of this library (other dtypes:
    torch.cat, as before round 4)."""
    assert a.shape == b.shape and a.dtype == b.dtype
    if a.dtype != torch.float32:
        return torch.cat([a, b], 0)
    out = torch.empty((2 * a.shape[0],) + tuple(a.shape[1:]), device=a.device, dtype=torch.float32)
    copy_many([(out[:a.shape[0]], a.contiguous()), (out[a.shape[0]:], b.contiguous())])
    return out


def splat(T, depth, featA, featB, with_flow, H, W, oy, ox, ds, K, radius, bf=0.0):
    """T [B,HT,WT,7], depth [B,HT,WT] sampled at (oy+ds*y, ox+ds*x) -> (out [B,C,H,W], z [B,1,H,W])."""
    lib = _abi.load()
    B, HT, WT, _ = T.shape
    CA = 0 if featA is None else featA.shape[1]
    CB = 0 if featB is None else featB.shape[1]
    Cc = CA + (3 if with_flow else 0) + CB
    out = _f32(B, Cc, H, W, like=T)
    z = _f32(B, 1, H, W, like=T)
    scratch = torch.empty(lib.codd_splat_scratch(B, H, W, float(radius)), device=T.device, dtype=torch.int32)
    _abi.check(lib.codd_splat(T.data_ptr(), depth.data_ptr(), HT, WT, oy, ox, ds,
                              None if featA is None else featA.data_ptr(), CA,
                              None if featB is None else featB.data_ptr(), CB, int(with_flow), B, H, W, *K,
                              float(radius), float(bf), out.data_ptr(), z.data_ptr(), scratch.data_ptr(), _stream()),
               "splat")
    return out, z


def induced_flow(T, depth, K):
    lib = _abi.load()
    B, H, W, _ = T.shape
    out = _f32(B, H, W, 3, like=T)
    _abi.check(lib.codd_induced_flow(T.data_ptr(), depth.data_ptr(), B, H, W, *K, out.data_ptr(), _stream()),
               "induced_flow")
    return out


def context_split(x):
    lib = _abi.load()
    B, _, h, w = x.shape
    net, inp = _f32(B, 128, h, w, like=x), _f32(B, 384, h, w, like=x)
    _abi.check(lib.codd_context_split(x.data_ptr(), B, h * w, net.data_ptr(), inp.data_ptr(), _stream()),
               "context_split")
    return net, inp


def se3_identity(B, h, w, device):
    lib = _abi.load()
    T = torch.empty(B, h, w, 7, device=device, dtype=torch.float32)
    _abi.check(lib.codd_se3_identity(T.data_ptr(), B * h * w, _stream()), "se3_identity")
    return T


def resize_bilinear(x, size, align_corners, out=None, accumulate=False, relu=False, extra=None):
    """``extra`` (contiguous [B, C, Ho, Wo]; ``out`` then a whole tensor): out = relu?((out + extra) + blend) with
    ``accumulate``, (extra + blend) without -- an add_relu launch folded in with its rounding."""
    lib = _abi.load()
    B, Cc, Hi, Wi = x.shape
    Ho, Wo = size
    if out is None:
        out = _f32(B, Cc, Ho, Wo, like=x)
    os_ = _as_slice(out)
    if extra is not None:
        assert extra.is_contiguous() and tuple(extra.shape) == (B, Cc, Ho, Wo) and os_.buf.shape[1] == Cc and os_.coff == 0
        _abi.check(lib.codd_resize_bilinear_add(x.data_ptr(), B, Cc, Hi, Wi, Ho, Wo, int(align_corners), os_.buf.data_ptr(),
                                                os_.buf.shape[1], os_.coff, int(accumulate), int(relu), extra.data_ptr(),
                                                _stream()), "resize_bilinear_add")
        return out
    _abi.check(lib.codd_resize_bilinear(x.data_ptr(), B, Cc, Hi, Wi, Ho, Wo, int(align_corners), os_.buf.data_ptr(),
                                        os_.buf.shape[1], os_.coff, int(accumulate), int(relu), _stream()),
               "resize_bilinear")
    return out


def add_relu(a, b=None, relu=True, out=None):
    lib = _abi.load()
    if out is None:
        out = torch.empty_like(a)
    _abi.check(lib.codd_add_relu(a.data_ptr(), None if b is None else b.data_ptr(), a.numel(), int(relu),
                                 out.data_ptr(), _stream()), "add_relu")
    return out


def copy_many(pairs):
    """dst.copy_(src) for up to 8 (dst, src) pairs of contiguous fp32 tensors in one kernel launch (a kernel node under
    graph capture: see runtime.FrameRunner._capture); pairs that do not meet the 16-byte rules fall back to one
    launch each."""
    import ctypes as C
    lib = _abi.load()
    fast = []
    for d, s in pairs:
        assert d.numel() == s.numel() and d.is_contiguous() and s.is_contiguous()
        if s.numel() % 4 == 0 and s.data_ptr() % 16 == 0 and d.data_ptr() % 16 == 0:
            fast.append((d, s))
        else:
            add_relu(s, None, relu=False, out=d)
    for i in range(0, len(fast), 8):
        chunk = fast[i:i + 8]
        n = len(chunk)
        srcs = (C.c_void_p * n)(*[s.data_ptr() for _, s in chunk])
        dsts = (C.c_void_p * n)(*[d.data_ptr() for d, _ in chunk])
        cnt = (C.c_longlong * n)(*[s.numel() for _, s in chunk])
        _abi.check(lib.codd_copy_many(srcs, dsts, cnt, n, _stream()), "copy_many")


def gru_gate_zr(t1, t2, inp, cor, mot, h):
    lib = _abi.load()
    B, _, hh, ww = h.shape
    zr, rh = _f32(B, 256, hh, ww, like=h), torch.empty_like(h)
    _abi.check(lib.codd_gru_gate_zr(t1.data_ptr(), t2.data_ptr(), inp.data_ptr(), _ptr(cor), _ptr(mot),
                                    h.data_ptr(), B, hh * ww, zr.data_ptr(), rh.data_ptr(), _stream()), "gru_gate_zr")
    return zr, rh


def _ptr(t):
    return None if t is None else t.data_ptr()


def _xs_view(st, coff=0):
    return _abi.XsView(st.buf.data_ptr(), st.c8, st.hp, st.wp, st.bt, st.bl, coff // 8, st.terms)


def gru_gate_zr_xs(t1, t2, inp, cor, mot, h, rh_xs):
    """gru_gate_zr writing r*h straight into the q convolutions' split-bf16 input (``rh_xs``: split_buffer) and only
    z as fp32: no fp32 r*h tensor, no re-layout launch.  Returns z [B,128,h,w]."""
    lib = _abi.load()
    B, _, hh, ww = h.shape
    z = torch.empty_like(h)
    _abi.check(lib.codd_gru_gate_zr_xs(t1.data_ptr(), t2.data_ptr(), inp.data_ptr(), _ptr(cor), _ptr(mot),
                                       h.data_ptr(), B, hh, ww, z.data_ptr(), _xs_view(rh_xs), _stream()),
               "gru_gate_zr_xs")
    return z


def gru_gate_q_xs(t1, t2, inp, cor, mot, z, h, h_xs):
    """gru_gate_q with ``z`` as returned by gru_gate_zr_xs; the new hidden state is returned as fp32 AND written as
    split records into ``h_xs`` (the input of the head convolution and of the next update's z|r convolutions)."""
    lib = _abi.load()
    B, _, hh, ww = h.shape
    ho = torch.empty_like(h)
    _abi.check(lib.codd_gru_gate_q_xs(t1.data_ptr(), t2.data_ptr(), inp.data_ptr(), _ptr(cor), _ptr(mot),
                                      z.data_ptr(), h.data_ptr(), B, hh, ww, ho.data_ptr(), _xs_view(h_xs), _stream()),
               "gru_gate_q_xs")
    return ho


def gru_gate_q(t1, t2, inp, cor, mot, zr, h):
    lib = _abi.load()
    B, _, hh, ww = h.shape
    ho = torch.empty_like(h)
    _abi.check(lib.codd_gru_gate_q(t1.data_ptr(), t2.data_ptr(), inp.data_ptr(), _ptr(cor), _ptr(mot),
                                   zr.data_ptr(), h.data_ptr(), B, hh * ww, ho.data_ptr(), _stream()), "gru_gate_q")
    return ho


class Fork:
    """Fork / join of independent launch chains over side HIP streams (parallel branches of the
    captured frame graph).  Discipline: every branch starts by waiting on the caller's stream, only
    the caller's stream consumes branch outputs, and ``join`` makes it wait for every branch used
    since the last join -- so the stream-tagged block re-use of the caching allocator stays safe."""

    serial = False  # debugging / per-launch timing: run every branch on the caller's stream

    def __init__(self, device, n):
        self.dev = device
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.used = []
        self.inline = False  # this fork only: run the branches on the caller's stream (A/B switches of call sites)

    def run(self, i, fn, *a, **k):
        if Fork.serial or self.inline:
            return fn(*a, **k)
        cur = torch.cuda.current_stream(self.dev)
        s = self.streams[i]
        if s not in self.used:
            s.wait_stream(cur)
            self.used.append(s)
        with torch.cuda.stream(s):
            return fn(*a, **k)

    def join(self):
        cur = torch.cuda.current_stream(self.dev)
        for s in self.used:
            cur.wait_stream(s)
        self.used = []

    def prefork(self, origin=None):
        """Bring every branch stream into the caller's ORIGIN stream's dependency graph (default: the current stream)
        before the branches are first used from a stream that is itself a fork.  Under hipGraph capture (ROCm 7.2) a
        stream whose FIRST captured operation is a wait on an already-forked stream crashes hipGraphInstantiate; a
        stream that joined the capture through the origin stream can wait on forked streams freely."""
        if Fork.serial:
            return
        cur = torch.cuda.current_stream(self.dev) if origin is None else origin
        for s in self.streams:
            s.wait_stream(cur)


# ----------------------------------------------------------------------------------------- fusion
def fusion_cues_lr(pred_curr, pred_warp, feat_curr, feat_warp, fea_l, fea_r, dsub, patch=3, ds=4):
    """-> corr_feat [B, 3 patch^2 + 4, H/ds, W/ds]; writes (pc, pw) sub-sampled into the 2-channel Slice ``dsub``."""
    lib = _abi.load()
    B, _, H, W = pred_curr.shape
    corr = _f32(B, 3 * patch * patch + 4, H // ds, W // ds, like=pred_curr)
    ds_ = _as_slice(dsub)
    _abi.check(lib.codd_fusion_cues_lr(pred_curr.data_ptr(), pred_warp.data_ptr(), feat_curr.data_ptr(),
                                       feat_warp.data_ptr(), fea_l.data_ptr(), fea_r.data_ptr(), B, H, W, patch, ds,
                                       feat_curr.shape[1], fea_l.shape[1], corr.data_ptr(), ds_.buf.data_ptr(),
                                       ds_.buf.shape[1], ds_.coff, _stream()), "fusion_cues_lr")
    return corr


def fusion_cues_fr(pred_curr, pred_warp, flow_warp, conf_warp, patch=3):
    lib = _abi.load()
    B, _, H, W = pred_curr.shape
    out = _f32(B, 3 * patch * patch + 5, H, W, like=pred_curr)
    _abi.check(lib.codd_fusion_cues_fr(pred_curr.data_ptr(), pred_warp.data_ptr(), flow_warp.data_ptr(),
                                       conf_warp.data_ptr(), B, H, W, patch, out.data_ptr(), _stream()), "fusion_cues_fr")
    return out


def fusion_forget(pred_curr, pred_warp, flow_warp, conf_warp, weff, patch=3):
    """cues -> forget head -> sigmoid in one launch (codd_fusion_forget); ``weff`` = the merged head
    [W_eff 9 x nc | beta 9 | c0] (fusion.Fusion.forget_matrix).  -> wr [B,1,H,W]."""
    lib = _abi.load()
    B, _, H, W = pred_curr.shape
    assert weff.numel() == 9 * (3 * patch * patch + 5) + 10 and weff.is_contiguous()
    wr = torch.empty_like(pred_curr)
    _abi.check(lib.codd_fusion_forget(pred_curr.data_ptr(), pred_warp.data_ptr(), flow_warp.data_ptr(),
                                      conf_warp.data_ptr(), B, H, W, patch, weff.data_ptr(), wr.data_ptr(), _stream()),
               "fusion_forget")
    return wr


def fusion_blend(pred_curr, pred_warp, wf_lr, wr, ds=4):
    lib = _abi.load()
    B, _, H, W = pred_curr.shape
    fused, wf, wro = torch.empty_like(pred_curr), torch.empty_like(pred_curr), torch.empty_like(pred_curr)
    _abi.check(lib.codd_fusion_blend(pred_curr.data_ptr(), pred_warp.data_ptr(), wf_lr.data_ptr(), wr.data_ptr(),
                                     B, H, W, ds, fused.data_ptr(), wf.data_ptr(), wro.data_ptr(), _stream()),
               "fusion_blend")
    return fused, wf, wro


def timestamp(buf, idx):
    """(diagnostics) buf[idx] (int64, device) = device wall clock (100 MHz) when this launch runs on the current stream."""
    lib = _abi.load()
    _require_gpu(buf)
    _abi.check(lib.codd_timestamp(buf.data_ptr() + 8 * idx, _stream()), "timestamp")


def disp_metrics(pred, gt, crop_hw, lo, hi, thr, meters, scratch=None):
    """Accumulate EPE / threshold-rate of one frame into ``meters`` ([3] fp64 on the device)."""
    lib = _abi.load()
    _require_gpu(pred)
    B, _, H, W = pred.shape
    _same_hw(pred, gt)
    if scratch is None:
        scratch = torch.empty(3 * 128 * B, device=pred.device, dtype=torch.float64)
    _abi.check(lib.codd_disp_metrics(pred.data_ptr(), gt.data_ptr(), B, H, W, crop_hw[0], crop_hw[1], float(lo),
                                     float(hi), float(thr), scratch.data_ptr(), meters.data_ptr(), _stream()),
               "disp_metrics")
    return meters


def tepe_metrics(pred, gt, pred_prev, gt_prev, flow_prev, crop_hw, lo, hi, bf, meters, scratch=None, gt_mask=None,
                 gt2_prev=None):
    """Accumulate the temporal metrics of one frame pair into ``meters`` ([7] fp64 on the device).  ``gt_mask``: the map
    the current frame's validity is computed from (default ``gt``); ``gt2_prev``: second-frame ground-truth disparity in
    the previous frame's coordinates, replacing the flow-warped ground truth (include/codd_hip.h)."""
    lib = _abi.load()
    _require_gpu(pred)
    B, _, H, W = pred.shape
    _same_hw(pred, gt, pred_prev, gt_prev, flow_prev, gt_mask, gt2_prev)
    if scratch is None:
        scratch = torch.empty(6 * 128 * B, device=pred.device, dtype=torch.float64)
    _abi.check(lib.codd_tepe_metrics(pred.data_ptr(), gt.data_ptr(), pred_prev.data_ptr(), gt_prev.data_ptr(),
                                     flow_prev.data_ptr(), None if gt_mask is None else gt_mask.data_ptr(),
                                     None if gt2_prev is None else gt2_prev.data_ptr(),
                                     B, H, W, crop_hw[0], crop_hw[1], float(lo), float(hi),
                                     float(bf), scratch.data_ptr(), meters.data_ptr(), _stream()), "tepe_metrics")
    return meters


def _same_hw(ref, *ts):
    """GT / flow maps handed to the metric kernels must be padded like the prediction: the kernels index them with
    the prediction's strides (a smaller tensor would be read out of bounds)."""
    for t in ts:
        if t is not None and tuple(t.shape[-2:]) != tuple(ref.shape[-2:]):
            raise ValueError("metric inputs must share the prediction's padded size %s, got %s" % (
                tuple(ref.shape[-2:]), tuple(t.shape[-2:])))


def sceneflow_metrics(Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ, crop_hw, lo, hi, bf, K,
                      meters, scratch=None):
    """Accumulate the five scene-flow sums of one frame pair into ``meters`` ([5] fp64 on the device)."""
    lib = _abi.load()
    _require_gpu(pred_prev)
    B, _, H, W = pred_prev.shape
    _same_hw(pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ)
    if tuple(Ts.shape) != (B, H, W, 7):
        raise ValueError("Ts must be [B,H,W,7] at the prediction's padded size")
    occ = None if gt_flow_occ is None else gt_flow_occ.to(torch.uint8).contiguous()
    if scratch is None:
        scratch = torch.empty(5 * 128 * B, device=pred_prev.device, dtype=torch.float64)
    _abi.check(lib.codd_sceneflow_metrics(Ts.contiguous().data_ptr(), pred_prev.data_ptr(), gt_disp_prev.data_ptr(),
                                          gt_flow_prev.data_ptr(), gt_disp_change.data_ptr(),
                                          None if occ is None else occ.data_ptr(), B, H, W, crop_hw[0], crop_hw[1],
                                          float(lo), float(hi), float(bf), *[float(v) for v in K], scratch.data_ptr(),
                                          meters.data_ptr(), _stream()), "sceneflow_metrics")
    return meters


IMAGENET_MEAN = (123.675, 116.28, 103.53)  # reference configs/datasets/custom.py:9
IMAGENET_STD = (58.395, 57.12, 57.375)


def preprocess(img_u8, bgr=True, mean=IMAGENET_MEAN, std=IMAGENET_STD, divisor=64):
    """uint8 [h,w,3] device image -> normalised, reflect-padded fp32 [1,3,H,W] (H, W multiples of 64)."""
    lib = _abi.load()
    _require_gpu(img_u8)
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous()
    h, w = img_u8.shape[:2]
    H, W = -(-h // divisor) * divisor, -(-w // divisor) * divisor
    out = torch.empty(1, 3, H, W, device=img_u8.device, dtype=torch.float32)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _abi.check(lib.codd_preprocess(img_u8.data_ptr(), h, w, int(bgr), m, s, H, W, out.data_ptr(), _stream()),
               "preprocess")
    return out


def fusion_select(mode, cur, warp, gt=None, K=0.5):
    """mode 'kalman' / 'gt' (ablation fusions).  cur, warp [B,1,H,W]; gt [B,1,hg,wg]."""
    lib = _abi.load()
    _require_gpu(cur)
    B, _, H, W = cur.shape
    out = torch.empty_like(cur)
    hg, wg = (gt.shape[-2], gt.shape[-1]) if gt is not None else (0, 0)
    _abi.check(lib.codd_fusion_select(0 if mode == "kalman" else 1, cur.data_ptr(), warp.data_ptr(),
                                      gt.data_ptr() if gt is not None else None, B, H, W, hg, wg, float(K),
                                      out.data_ptr(), _stream()), "fusion_select")
    return out


def gt_motion(img_prev, feat_prev, disp_prev, gt_flow, gt_disp_change, gt_flow_occ):
    """-> [img_warp, feat_warp, conf, disp_warp [B,1,H,W], flow3] (GTMotion ablation)."""
    lib = _abi.load()
    _require_gpu(img_prev)
    B, _, H, W = img_prev.shape
    hg, wg = gt_flow.shape[-2:]
    occ = gt_flow_occ.to(torch.uint8).contiguous()
    img_w, feat_w = torch.empty_like(img_prev), torch.empty_like(feat_prev)
    conf, flow3 = torch.empty_like(img_prev), torch.empty_like(img_prev)
    disp_w = torch.empty(B, 1, H, W, device=img_prev.device, dtype=torch.float32)
    _abi.check(lib.codd_gt_motion(img_prev.data_ptr(), disp_prev.data_ptr(), feat_prev.data_ptr(), feat_prev.shape[1],
                                  gt_flow.data_ptr(), gt_disp_change.data_ptr(), occ.data_ptr(), B, H, W, hg, wg,
                                  img_w.data_ptr(), feat_w.data_ptr(), conf.data_ptr(), disp_w.data_ptr(),
                                  flow3.data_ptr(), _stream()), "gt_motion")
    return [img_w, feat_w, conf, disp_w, flow3]
