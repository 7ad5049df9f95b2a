"""GPU parity of the conv family and the HITNet kernels against the CPU oracle (fp32).

Tolerances: convs are fp32 MFMA fma chains vs MKL-DNN on CPU -> relative 1e-4 of the output
scale; cost/warp kernels 1e-4 absolute on O(1..10) values; arg-min indices exact except where
the two best costs differ by < 1e-5 (reported, bounded)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed + sum(s))
    return torch.randn(*s, generator=g)


CONV_CASES = [
    # Cin, Cout, k, stride, pad, dil, H, W, act
    (3, 16, 3, 1, 1, 1, 64, 128, "lrelu"),
    (16, 16, 3, 1, 1, 1, 40, 72, "lrelu"),
    (16, 24, 4, 2, 1, 1, 64, 96, "lrelu"),
    (24, 24, 3, 1, 1, 1, 9, 15, "none"),
    (32, 32, 3, 1, 3, 3, 36, 60, "relu"),
    (32, 34, 3, 1, 1, 1, 18, 30, "none"),
    (16, 16, 4, 4, 0, 1, 64, 128, "lrelu"),
    (64, 16, 1, 1, 0, 1, 36, 60, "lrelu"),
    (9, 128, 7, 1, 3, 1, 24, 40, "relu"),
    (196, 256, 3, 1, 1, 1, 16, 24, "relu"),
    (128, 128, 3, 1, 4, 4, 24, 40, "sigmoid"),
    (64, 30, 7, 1, 3, 1, 36, 60, "relu"),
    (3, 64, 7, 2, 3, 1, 64, 128, "none"),
    (64, 96, 3, 2, 1, 1, 64, 96, "tanh"),
    (32, 32, 3, 1, 1, 1, 144, 256, "mish"),
    (270, 512, 1, 1, 0, 1, 16, 32, "relu"),
]


def _act(v, act):
    return dict(none=lambda t: t, lrelu=lambda t: F.leaky_relu(t, 0.2), relu=F.relu, sigmoid=torch.sigmoid,
                tanh=torch.tanh, mish=F.mish)[act](v)


# precision mode of the conv family -> tolerance relative to the output scale (fp32: fma chains vs MKL-DNN;
# split: three-term split-bf16 products, ~2^-17 per product; bf16: 2^-9 per operand; fp16: 2^-12 per operand)
PRECISIONS = [("fp32", 2e-4), ("split", 2e-4), ("split16", 2e-4), ("bf16", 3e-2), ("fp16", 4e-3)]


@pytest.fixture(params=PRECISIONS, ids=[p[0] for p in PRECISIONS])
def precision(request):
    from codd_amd import ops
    prev = ops.set_conv_precision(request.param[0])
    yield request.param
    ops.set_conv_precision(prev)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, precision):
    from codd_amd import ops
    cin, cout, k, s, p, d, H, W, act = case
    x, w, b = rnd(2, cin, H, W), rnd(cout, cin, k, k, seed=1) / (cin * k * k) ** 0.5, rnd(cout, seed=2) * 0.1
    ref = _act(F.conv2d(x, w, b, s, p, d), act)
    pc = ops.PackedConv(w.to(dev()), b.to(dev()))
    got = ops.conv2d(x.to(dev()), pc, stride=s, pad=p, dil=d, act=act).cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < precision[1] * max(1.0, ref.abs().max().item()), err


def test_split_bf16_conv_every_launch_configuration():
    """Every (tile, chunk depth, wave grid) the library accepts for a layer gives the same result to the split-bf16
    bound: |err| <= 3 * 2^-18 * sum |w||x| per output (dropped lo*lo term + the two split residuals), checked against
    an fp64 reference; terms = 1 (plain bf16 operands) to 2^-8 * sum |w||x|; terms = 16 (IEEE fp16 operands on
    v_mfma_f32_16x16x32_f16, round 5) to 2^-11 * sum |w||x| (two operands rounded to 11 significant bits each); terms = 48
    (split-fp16: hi + lo fp16 planes, 22-bit operands) to 2^-20 * sum |w||x| -- 16x tighter than split-bf16."""
    from codd_amd import _abi, ops
    lib = _abi.load()
    for (cin, cout, k, s, p, d, H, W) in [(40, 64, 3, 1, 1, 1, 27, 40), (16, 16, 3, 1, 1, 1, 48, 64),
                                         (24, 40, 3, 2, 1, 1, 32, 64), (72, 130, 1, 1, 0, 1, 20, 33)]:
        x, w, b = rnd(1, cin, H, W), rnd(cout, cin, k, k, seed=1) / (cin * k * k) ** 0.5, rnd(cout, seed=2) * 0.1
        ref = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
        mag = F.conv2d(x.abs().double(), w.abs().double(), None, s, p, d)
        Ho, Wo = ref.shape[2:]
        pc = ops.PackedConv(w.to(dev()), b.to(dev()))
        for terms, bound in ((3, 3.5 * 2.0 ** -18), (1, 2.0 ** -8), (16, 2.0 ** -11), (48, 2.0 ** -20)):
            prev = ops.set_conv_precision({3: "split", 1: "bf16", 16: "fp16", 48: "split16"}[terms])
            try:
                n = 0
                for c in ops._bf16_candidates(pc, Ho, Wo, 1, k * k, terms):
                    pp = _abi.ConvParams()
                    pp.C0, pp.C1, pp.B, pp.Hin, pp.Win, pp.Cout, pp.Hout, pp.Wout = cin, 0, 1, H, W, cout, Ho, Wo
                    pp.kh, pp.kw, pp.sy, pp.sx, pp.pad_t, pp.pad_l, pp.dil_y, pp.dil_x = k, k, s, s, p, p, d, d
                    pp.terms = terms
                    if not ops._cfg_ok(lib, pp, c):
                        continue
                    pc.tuned.clear()
                    pc.tuned[(Ho, Wo, 1, s, s, d, d, p, False, terms)] = c
                    out = torch.full((1, cout, Ho, Wo), float("nan"), device=dev())
                    ops.conv2d(x.to(dev()), pc, stride=s, pad=p, dil=d, out=out)
                    err = ((out.cpu().double() - ref).abs() / (mag + 1e-6)).max().item()
                    assert err < bound + 2e-7, (c, terms, err)
                    n += 1
                assert n >= 3, n
            finally:
                ops.set_conv_precision(prev)


def test_conv2d_views_residuals(precision):
    from codd_amd import ops
    from codd_amd.ops import Slice
    xa, xb = rnd(1, 24, 36, 60), rnd(1, 16, 36, 60, seed=3)
    w, b = rnd(32, 40, 3, 3, seed=4) / 19.0, rnd(32, seed=5) * 0.1
    r1, r2, post = rnd(1, 32, 36, 60, seed=6), rnd(1, 32, 36, 60, seed=7), rnd(1, 32, 36, 60, seed=8)
    ref = F.relu(F.conv2d(torch.cat([xa, xb], 1), w, b, 1, 1) + r1 + r2) + post
    big_in = torch.zeros(1, 40, 36, 60)
    big_in[:, 8:32] = xa
    out = torch.full((1, 48, 36, 60), -7.0, device=dev())
    pc = ops.PackedConv(w.to(dev()), b.to(dev()))
    ops.conv2d(Slice(big_in.to(dev()), 8, 24), pc, x2=xb.to(dev()), pad=1, act="relu", res1=r1.to(dev()),
               res2=r2.to(dev()), post=post.to(dev()), out=Slice(out, 10, 32))
    out = out.cpu()
    assert (out[:, 10:42] - ref).abs().max().item() < precision[1] * ref.abs().max().item()
    assert (out[:, :10] == -7).all() and (out[:, 42:] == -7).all()


def test_conv_right_pad_stride41_and_relu_ch0(precision):
    from codd_amd import ops
    x, w, b = rnd(1, 16, 32, 64), rnd(16, 16, 4, 4, seed=1) / 16.0, rnd(16, seed=2) * 0.1
    ref = F.leaky_relu(F.conv2d(F.pad(x, (0, 3, 0, 0)), w, b, (4, 1)), 0.2)
    pc = ops.PackedConv(w.to(dev()), b.to(dev()))
    got = ops.conv2d(x.to(dev()), pc, stride=(4, 1), pad_tl=(0, 0, 0, 3), act="lrelu").cpu()
    assert got.shape == ref.shape == (1, 16, 8, 64)
    assert (got - ref).abs().max().item() < precision[1] * 3
    w3 = rnd(16, 16, 3, 3, seed=3) / 12.0
    res = rnd(1, 16, 32, 64, seed=4)
    ref = F.conv2d(x, w3, b, 1, 1) + res
    ref[:, :1] = F.relu(ref[:, :1])
    got = ops.conv2d(x.to(dev()), ops.PackedConv(w3.to(dev()), b.to(dev())), pad=1, res1=res.to(dev()),
                     act="relu_ch0").cpu()
    assert (got - ref).abs().max().item() < precision[1] * 3


@pytest.mark.parametrize("cin,cout,H,W", [(32, 24, 9, 15), (16, 16, 72, 120), (24, 16, 36, 60)])
def test_deconv2x2(cin, cout, H, W, precision):
    from codd_amd import ops
    x, w, b = rnd(2, cin, H, W), rnd(cin, cout, 2, 2, seed=1) / cin ** 0.5, rnd(cout, seed=2) * 0.1
    ref = F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2), 0.2)
    pc = ops.PackedConv(w.to(dev()), b.to(dev()), deconv=True)
    got = ops.conv2d(x.to(dev()), pc, act="lrelu").cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < precision[1] * 3


@pytest.mark.parametrize("Ht,Wt,D", [(9, 15, 20), (18, 30, 40), (16, 32, 80), (36, 60, 320)])
def test_tile_costvol_argmin(Ht, Wt, D):
    from codd_amd import ops
    from codd_amd.ops import Slice
    from oracle import stereo as ost
    tl, tr = rnd(1, 16, Ht, Wt), rnd(1, 16, Ht, 4 * Wt, seed=1)
    cost_ref, d_ref = ost.tile_cost_volume_min(tl, tr, D)
    cvfull = ost.tile_cost_volume(tl, tr, D)
    cost = torch.empty(1, 1, Ht, Wt, device=dev())
    hyp = torch.full((1, 16, Ht, Wt), 5.0, device=dev())
    ops.tile_costvol_argmin(tl.to(dev()), tr.to(dev()), D, cost, Slice(hyp, 0, 3))
    cost, hyp = cost.cpu(), hyp.cpu()
    assert (cost - cost_ref).abs().max().item() < 1e-4
    assert (hyp[:, 1:3] == 0).all() and (hyp[:, 3:] == 5).all()
    mism = hyp[:, 0:1] != d_ref
    if mism.any():  # only admissible where the two costs tie to rounding
        alt = torch.gather(cvfull, 1, hyp[:, 0:1].long())
        assert ((alt - cost_ref).abs()[mism] < 1e-5).all()
        assert mism.float().mean().item() < 0.01
    # systematic ties in the zero-padded region must resolve to the FIRST index exactly
    x_idx = torch.arange(Wt).view(1, 1, 1, Wt).expand_as(d_ref)
    padded = d_ref > 4 * x_idx
    assert (hyp[:, 0:1][padded] == d_ref[padded]).all()


@pytest.mark.parametrize("C,Ht,Wt,two", [(32, 9, 15, False), (24, 18, 30, True), (16, 36, 64, True)])
def test_tile_warp_cost(C, Ht, Wt, two):
    from codd_amd import ops
    from codd_amd.ops import Slice
    from oracle import stereo as ost
    fl, fr = rnd(1, C, 4 * Ht, 4 * Wt), rnd(1, C, 4 * Ht, 4 * Wt, seed=1)

    def plane(seed):
        h = rnd(1, 16, Ht, Wt, seed=seed)
        h[:, 0] = h[:, 0].abs() * 6
        h[:, 1:3] *= 0.3
        h[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 2.5, 300.0])  # integer / far out-of-range samples
        return h

    h0, h1 = plane(2), plane(3)
    ref0 = torch.cat([ost.unshuffle4(fl.abs().sum(1, keepdim=True)), ost.tile_warping(h0[:, :3], fl, fr)], 1)
    ref1 = torch.cat([ost.unshuffle4(fl.abs().sum(1, keepdim=True)), ost.tile_warping(h1[:, :3], fl, fr)], 1)
    big = torch.zeros(1, 64, Ht, Wt)
    big[:, 32:48] = h1
    bigd = big.to(dev())
    o0, o1 = ops.tile_warp_cost(fl.to(dev()), fr.to(dev()), h0.to(dev()), Slice(bigd, 32, 16) if two else None)
    assert (o0.cpu() - ref0).abs().max().item() < 2e-5 * ref0.abs().max().item()
    if two:
        assert (o1.cpu() - ref1).abs().max().item() < 2e-5 * ref1.abs().max().item()


def test_hyp_upsample_select():
    from codd_amd import ops
    from oracle import stereo as ost
    h = rnd(2, 16, 9, 15)
    for scale in (1.0, 2.0):
        out = torch.empty(2, 16, 18, 30, device=dev())
        ops.hyp_upsample(h.to(dev()), scale, out)
        assert (out.cpu() - ost.upsample_hyp(h, scale, 2)).abs().max().item() < 1e-6
    upd, cur, prev = rnd(2, 34, 9, 15, seed=1), rnd(2, 16, 9, 15, seed=2), rnd(2, 16, 9, 15, seed=3)
    upd[0, 1, 0, :5] = upd[0, 0, 0, :5]  # conf ties -> previous
    sel = upd[:, :2].argmax(1, keepdim=True).float()
    ref = sel * ost._relu_d(cur + upd[:, 18:34]) + (1 - sel) * ost._relu_d(prev + upd[:, 2:18])
    out = torch.empty(2, 16, 9, 15, device=dev())
    ops.hyp_select(upd.to(dev()), cur.to(dev()), prev.to(dev()), out)
    assert (out.cpu() - ref).abs().max().item() < 1e-6


def test_c_abi_error_codes_are_loud():
    """Invalid arguments come back as negative CODD_E* codes (never a silent fallback), surfaced as
    CoddHipError by the Python host (reference behaviour: Python exceptions)."""
    import ctypes as C
    from codd_amd import _abi, ops
    lib = _abi.load()
    x = rnd(1, 32, 16, 32).to("cuda")
    pc = ops.PackedConv(rnd(32, 32, 3, 3, seed=1).to("cuda"), None)
    p = _abi.ConvParams()
    C.memset(C.byref(p), 0, C.sizeof(p))
    assert lib.codd_conv2d(C.byref(p), None) == -1  # CODD_EINVAL: no input / output / weights
    old, prev = ops._FORCE_NW, ops.set_conv_precision("fp32")
    ops._FORCE_NW = 7  # not an instantiated workgroup height
    try:
        with pytest.raises(_abi.CoddHipError, match="-2"):  # CODD_EUNSUPPORTED
            ops.conv2d(x, pc, pad=1)
        ops.set_conv_precision("split")
        pc.tuned[(16, 32, 1, 1, 1, 1, 1, 1, False, 3)] = (2, 8, 16, 3, 2, 4, 1, 3)  # 48-channel groups: not instantiated
        with pytest.raises(_abi.CoddHipError, match="-2"):
            ops.conv2d(x, pc, pad=1)
    finally:
        ops._FORCE_NW = old
        ops.set_conv_precision(prev)
    assert lib.codd_tile_costvol_argmin(None, None, 1, 16, 4, 8, 32, 4, None, 1, 0, None, 1, 0, 4, None) == -1
    with pytest.raises(_abi.CoddHipError):
        ops.conv2d(x.cpu(), pc, pad=1)  # host tensor: the product path has no CPU fallback


def test_autotuned_launch_configuration_keeps_results():
    """ops.enable_autotune: the first launch of a layer shape times the alternative (npb, nw, ck)
    configurations; whichever wins, the output only differs by fp32 summation order."""
    from codd_amd import ops
    x = rnd(1, 64, 72, 120).to("cuda")
    w = (rnd(128, 64, 3, 3, seed=1) / 24.0).to("cuda")
    ref = ops.conv2d(x, ops.PackedConv(w, None), pad=1, act="relu")
    n0 = len(ops.AUTOTUNE_LOG)
    ops.enable_autotune(True, shipped=False)
    try:
        pc = ops.PackedConv(w, None)
        y1 = ops.conv2d(x, pc, pad=1, act="relu")
        y2 = ops.conv2d(x, pc, pad=1, act="relu")  # cached configuration
    finally:
        ops.enable_autotune(False)
    assert len(ops.AUTOTUNE_LOG) > n0 and len(pc.tuned) == 1  # (split mode logs the fp32 pass, the split pass and the choice)
    assert torch.equal(y1, y2)
    assert (y1 - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(20, 40, 3, 1, 1, 1, 37, 70), (64, 96, 3, 1, 4, 4, 24, 40), (33, 16, 1, 1, 0, 1, 19, 50),
                                   (16, 130, 4, 4, 0, 1, 64, 128), (9, 128, 7, 1, 3, 1, 24, 40), (96, 64, 3, 2, 1, 1, 40, 56),
                                   (48, 80, 3, 1, 1, 1, 37, 72), (100, 256, 1, 1, 0, 1, 21, 44), (36, 36, 3, 1, 1, 1, 72, 120)])
def test_every_launch_configuration_the_autotuner_may_pick(shape):
    """All (npb, nw, ck, mb) candidates of ops._autotune -- every kernel instantiation, including the 2-, 8-
    and 9-wave workgroups -- against torch's fp32 convolution on ragged shapes."""
    import warnings
    from codd_amd import ops
    cin, cout, k, s, p, d, H, W = shape
    x, w, b = rnd(1, cin, H, W), rnd(cout, cin, k, k, seed=1) / (cin * k * k) ** 0.5, rnd(cout, seed=2) * 0.1
    ref = F.conv2d(x, w, b, stride=s, padding=p, dilation=d)
    xd = x.to("cuda")
    pc = ops.PackedConv(w.to("cuda"), b.to("cuda"))
    key = (ref.shape[2], ref.shape[3], 1, s, s, d, d, p, False)
    cin_pad = -(-cin // 4) * 4
    cks = sorted({c for c in (8, 12, 16, 24, 32) if c <= cin_pad} | {min(cin_pad, 32)})
    tried = 0
    for mb in (1, 2, 4):
        if 16 * mb > max(16, -(-cout // 16) * 16):
            continue
        for npb in (1, 2, 4):
            for nw in ((4, 9, 2, 8) if npb == 1 else (4,)):
                for ck in cks:
                    pc.tuned[key] = (npb, nw, ck, mb)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        y = ops.conv2d(xd, pc, stride=s, pad=p, dil=d)
                    if tuple(pc.tuned[key]) != (npb, nw, ck, mb):
                        continue  # CODD_EUNSUPPORTED (staging / LDS limits): ops fell back to the heuristic
                    tried += 1
                    err = (y.cpu() - ref).abs().max().item()
                    assert err < 2e-5 * max(1.0, ref.abs().max().item()), ((npb, nw, ck, mb), err)
    # quad layout (ds_read_b128 operands): ck 16 / 32, unit stride
    tq = 0
    if s <= 2 and cin_pad >= 16:
        for mb in (1, 2, 4):
            if 16 * mb > max(16, -(-cout // 16) * 16):
                continue
            for npb, nw in ((1, 4), (2, 4), (4, 4), (1, 9)):
                for ck in (16, 32):
                    pc.tuned[key] = (npb, nw, ck, mb, 1)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        y = ops.conv2d(xd, pc, stride=s, pad=p, dil=d)
                    if tuple(pc.tuned[key]) != (npb, nw, ck, mb, 1):
                        continue
                    tq += 1
                    err = (y.cpu() - ref).abs().max().item()
                    assert err < 2e-5 * max(1.0, ref.abs().max().item()), ((npb, nw, ck, mb, 1), err)
        assert tq >= 2 or W % 4 != 0, tq  # rows that are not 16-byte aligned are refused (CODD_EUNSUPPORTED)
    assert tried >= 3


def test_autotune_leaves_in_place_accumulation_alone():
    """out = conv(x) + out (HRNet's fuse layers write into their own residual operand): the tuner's timing
    launches must not accumulate into the real output."""
    from codd_amd import ops
    x = rnd(1, 32, 36, 60).to("cuda")
    w = (rnd(32, 32, 3, 3, seed=1) / 17.0).to("cuda")
    acc0 = rnd(1, 32, 36, 60, seed=5).to("cuda")
    ref = F.conv2d(x.cpu(), w.cpu(), padding=1) + acc0.cpu()
    ops.enable_autotune(True, shipped=False)
    try:
        pc = ops.PackedConv(w, None)
        acc = acc0.clone()
        ops.conv2d(x, pc, pad=1, res1=acc, out=acc)
    finally:
        ops.enable_autotune(False)
    assert (acc.cpu() - ref).abs().max().item() < 1e-4


def test_tunable_configurations_with_views_and_two_inputs():
    """Quad-layout and classic kernels with a channel-slice view as first input (24 of 40 channels, offset 6), a
    second input (18 channels: the quad of channels 24..27 straddles both), residual + post operands and a slice
    of a wider buffer as output."""
    import warnings
    from codd_amd import ops
    from codd_amd.ops import Slice
    H, W = 36, 64
    big, x2 = rnd(1, 40, H, W), rnd(1, 18, H, W, seed=3)
    w, b = rnd(48, 42, 3, 3, seed=1) / 19.0, rnd(48, seed=2) * 0.1
    res, post = rnd(1, 48, H, W, seed=4), rnd(1, 48, H, W, seed=5)
    ref = F.relu(F.conv2d(torch.cat([big[:, 6:30], x2], 1), w, b, padding=1) + res) + post
    bd, x2d, resd, postd = big.to("cuda"), x2.to("cuda"), res.to("cuda"), post.to("cuda")
    pc = ops.PackedConv(w.to("cuda"), b.to("cuda"))
    key = (H, W, 1, 1, 1, 1, 1, 1, True, 0)
    n = 0
    prev = ops.set_conv_precision("fp32")
    try:
        for cfg in [(1, 4, 12, 2, 0), (2, 4, 8, 1, 0), (1, 9, 16, 2, 0), (1, 2, 8, 2, 0), (1, 8, 16, 1, 0),
                    (1, 4, 16, 2, 1), (2, 4, 16, 2, 1), (1, 9, 16, 2, 1), (1, 4, 32, 1, 1), (4, 4, 16, 1, 1)]:
            pc.tuned[key] = cfg
            outbuf = torch.zeros(1, 64, H, W, device="cuda")
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ops.conv2d(Slice(bd, 6, 24), pc, x2=x2d, pad=1, act="relu", res1=resd, post=postd, out=Slice(outbuf, 8, 48))
            if tuple(pc.tuned[key]) != cfg:
                continue
            n += 1
            assert (outbuf[:, 8:56].cpu() - ref).abs().max().item() < 5e-5, cfg
            assert outbuf[:, :8].abs().max().item() == 0 and outbuf[:, 56:].abs().max().item() == 0, cfg
        assert n >= 6, n
        # split-bf16 kernel: the channel octet 24..31 straddles both inputs; every wave grid / tile / chunk depth
        ops.set_conv_precision("split")
        key = key[:-1] + (3,)
        n = 0
        for cfg in [(2, 8, 16, 4, 2, 4, 1), (1, 9, 16, 4, 2, 2, 2), (2, 5, 8, 4, 2, 2, 2), (2, 8, 16, 2, 2, 4, 1),
                    (1, 16, 8, 1, 2, 4, 1), (2, 16, 16, 1, 2, 4, 1), (2, 4, 16, 2, 2, 4, 1), (1, 8, 32, 1, 2, 4, 1)]:
            pc.tuned[key] = cfg + (3,)
            outbuf = torch.zeros(1, 64, H, W, device="cuda")
            ops.conv2d(Slice(bd, 6, 24), pc, x2=x2d, pad=1, act="relu", res1=resd, post=postd, out=Slice(outbuf, 8, 48))
            n += 1
            assert (outbuf[:, 8:56].cpu() - ref).abs().max().item() < 1e-4, cfg
            assert outbuf[:, :8].abs().max().item() == 0 and outbuf[:, 56:].abs().max().item() == 0, cfg
        # eight consumer waves: k-split pairs (ksplit = 2: partial sums exchanged through LDS, odd and even k-step
        # counts per chunk) and 4 x 2 wave grids
        lib = ops._abi.load()
        n = 0
        for cfg in [(1, 10, 16, 4, 2, 2, 2, 3, 2), (1, 9, 8, 4, 2, 2, 2, 3, 2), (2, 5, 16, 4, 2, 2, 2, 3, 2),
                    (2, 8, 16, 2, 2, 4, 1, 3, 2), (1, 12, 8, 2, 2, 4, 1, 3, 2), (1, 12, 16, 4, 2, 4, 1, 3, 2),
                    (2, 6, 32, 2, 2, 4, 1, 3, 2), (1, 16, 16, 4, 2, 4, 2, 3, 1), (2, 8, 8, 4, 2, 4, 2, 3, 1),
                    (1, 12, 16, 4, 2, 4, 2, 3, 1), (2, 6, 8, 4, 2, 4, 2, 3, 1)]:
            pp = ops._abi.ConvParams()
            pp.C0, pp.C1, pp.B, pp.Hin, pp.Win, pp.Cout, pp.Hout, pp.Wout = 24, 18, 1, H, W, 48, H, W
            pp.kh, pp.kw, pp.sy, pp.sx, pp.pad_t, pp.pad_l, pp.dil_y, pp.dil_x, pp.terms = 3, 3, 1, 1, 1, 1, 1, 1, 3
            if not ops._cfg_ok(lib, pp, cfg):  # over the LDS / DMA-queue budget of this layer
                continue
            n += 1
            pc.tuned[key] = cfg
            outbuf = torch.zeros(1, 64, H, W, device="cuda")
            ops.conv2d(Slice(bd, 6, 24), pc, x2=x2d, pad=1, act="relu", res1=resd, post=postd, out=Slice(outbuf, 8, 48))
            assert (outbuf[:, 8:56].cpu() - ref).abs().max().item() < 1e-4, cfg
            assert outbuf[:, :8].abs().max().item() == 0 and outbuf[:, 56:].abs().max().item() == 0, cfg
        assert n >= 8, n
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("mode,tol", [("split", 2e-4), ("split16", 2e-4), ("bf16", 4e-2), ("fp16", 5e-3)])
def test_conv_chain_through_split_records(mode, tol):
    """conv -> conv with the intermediate written by the first kernel's epilogue as split-bf16 records straight into
    the second one's input tensor (ops.split_buffer / xs_out): no fp32 tensor, no re-layout pass.  Output channels that
    are not a multiple of 8 (zero-padded octet), a 3x3 consumer (border 1), a 1x1 consumer reading a channel slice, and
    a second frame through the same persistent buffers."""
    from codd_amd import ops
    B, H, W = 1, 36, 60
    w0, b0 = rnd(44, 20, 3, 3, seed=1) / 13.0, rnd(44, seed=2) * 0.1
    w1, b1 = rnd(32, 44, 3, 3, seed=3) / 20.0, rnd(32, seed=4) * 0.1
    w2, b2 = rnd(24, 16, 1, 1, seed=5) / 4.0, rnd(24, seed=6) * 0.1
    prev = ops.set_conv_precision(mode)
    try:
        p0, p1, p2 = (ops.PackedConv(w.to(dev()), b.to(dev())) for w, b in ((w0, b0), (w1, b1), (w2, b2)))
        for frame in range(2):
            x = rnd(B, 20, H, W, seed=10 + frame)
            r0 = F.relu(F.conv2d(x, w0, b0, padding=1))
            r1 = F.relu(F.conv2d(r0, w1, b1, padding=1))
            r2 = F.conv2d(r1[:, 8:24], w2, b2)
            s0 = ops.split_buffer("t0", B, 44, H, W, 1, dev())
            s1 = ops.split_buffer("t1", B, 32, H, W, 0, dev())
            assert ops.conv2d(x.to(dev()), p0, pad=1, act="relu", xs_out=s0) is s0
            ops.conv2d(None, p1, pad=1, act="relu", xs=s0, xs_out=s1)
            y1 = ops.conv2d(None, p1, pad=1, act="relu", xs=s0)          # the same layer with an fp32 result
            y2 = ops.conv2d(None, p2, xs=s1, xs_coff=8)
            assert (y1.cpu() - r1).abs().max().item() < tol * r1.abs().max().item(), frame
            assert (y2.cpu() - r2).abs().max().item() < tol * max(1.0, r2.abs().max().item()), frame
        # the record-writing epilogue of the eight-consumer-wave configurations (k-split pairs finish half of the tiles each)
        terms = ops._TERMS[mode]
        k0 = (H, W, B, 1, 1, 1, 1, 1, False, terms, "split")
        for cfg in [(1, 10, 8, 4, 2, 2, 2, terms, 2), (2, 8, 16, 2, 2, 4, 1, terms, 2), (1, 12, 8, 4, 2, 4, 1, terms, 2),
                    (1, 16, 8, 4, 2, 4, 2, terms, 1), (2, 6, 16, 4, 2, 4, 2, terms, 1)]:
            p0.tuned[k0] = cfg
            p1.tuned[k0] = cfg
            s0.buf.zero_()
            s1.buf.zero_()
            ops.conv2d(x.to(dev()), p0, pad=1, act="relu", xs_out=s0)
            assert tuple(p0.tuned[k0]) == cfg
            y1 = ops.conv2d(None, p1, pad=1, act="relu", xs=s0)
            ops.conv2d(None, p1, pad=1, act="relu", xs=s0, xs_out=s1)
            y2 = ops.conv2d(None, p2, xs=s1, xs_coff=8)
            assert (y1.cpu() - r1).abs().max().item() < tol * r1.abs().max().item(), cfg
            assert (y2.cpu() - r2).abs().max().item() < tol * max(1.0, r2.abs().max().item()), cfg
    finally:
        ops.set_conv_precision(prev)


def test_conv2d_multi_equals_separate_launches():
    """codd_conv2d_multi: up to four independent convolutions (different maps, channel counts, strides; residual +
    ReLU epilogues) as ONE launch -- HRNet's resolution branches -- against torch; a job whose rows are not 16-byte
    aligned (18 x 30 map) silently takes the single-launch path."""
    from codd_amd import ops
    shapes = [(18, 48, 80), (36, 24, 40), (72, 12, 20), (144, 18, 30)]
    prev = ops.set_conv_precision("fp32")
    try:
        jobs, refs = [], []
        for n, (c, h, w) in enumerate(shapes):
            x, res = rnd(1, c, h, w, seed=n), rnd(1, c, h, w, seed=10 + n)
            wt, b = rnd(c, c, 3, 3, seed=20 + n) / (3.0 * c ** 0.5), rnd(c, seed=30 + n) * 0.1
            refs.append(F.relu(F.conv2d(x, wt, b, padding=1) + res))
            jobs.append(dict(x=x.to(dev()), pc=ops.PackedConv(wt.to(dev()), b.to(dev())), pad=1, act="relu", res1=res.to(dev())))
        # + a strided job (HRNet transition shape)
        x = rnd(1, 36, 24, 40, seed=50)
        wt, b = rnd(72, 36, 3, 3, seed=51) / 18.0, rnd(72, seed=52) * 0.1
        refs.append(F.conv2d(x, wt, b, stride=2, padding=1))
        jobs.append(dict(x=x.to(dev()), pc=ops.PackedConv(wt.to(dev()), b.to(dev())), stride=2, pad=1))
        calls = []
        orig = ops._launch_conv_multi
        ops._launch_conv_multi = lambda lib, p, n, s: (calls.append(n), orig(lib, p, n, s))[1]
        try:
            outs = ops.conv2d_multi(jobs)
        finally:
            ops._launch_conv_multi = orig
        assert calls == [4], calls  # four aligned jobs in one launch, the 18 x 30 one on its own
        for o, r in zip(outs, refs):
            assert o.shape == r.shape
            assert (o.cpu() - r).abs().max().item() < 5e-5 * max(1.0, r.abs().max().item())
    finally:
        ops.set_conv_precision(prev)


ROLL_CASES = [
    # mode (0: 3x3, 1: 3x3 -> 3x3, 2: 1x1 -> 3x3), C, cin, c0 (first source's channels), B, H, W, residual, rows/workgroup
    (0, 16, 16, 16, 1, 37, 130, False, 8), (1, 16, 16, 16, 2, 29, 75, True, 7), (1, 16, 16, 16, 1, 64, 200, False, 64),
    (2, 16, 32, 16, 1, 33, 190, False, 5), (0, 32, 32, 32, 1, 21, 64, False, 4), (1, 32, 32, 32, 1, 40, 125, True, 9),
    (2, 32, 40, 24, 2, 19, 61, False, 19), (2, 32, 64, 64, 1, 17, 70, False, 6), (1, 16, 16, 16, 1, 5, 9, True, 3),
    (2, 16, 48, 16, 1, 12, 63, False, 4), (1, 16, 16, 8, 1, 130, 62, True, 130), (0, 16, 16, 16, 1, 3, 1, False, 1),
]


@pytest.mark.parametrize("case", ROLL_CASES, ids=lambda c: "m%d_c%d_%dx%d" % (c[0], c[1], c[5], c[6]))
def test_conv_roll_matches_torch(case):
    """Rolling-window launches (csrc/conv_roll.hip: BasicBlock pairs, merge heads / tails of HITNet, reference
    propagation.py:103-121, backbone.py:24-39) against torch fp32 on the CPU: strips that do not divide the width,
    row blocks that do not divide the height, maps smaller than one strip / one row block, two concatenated sources,
    batch 2, output written into a channel slice of a larger buffer (neighbouring channels untouched)."""
    from codd_amd import ops
    from codd_amd.ops import Slice
    mode, C, cin, c0, B, H, W, res, rh = case
    k0 = 1 if mode == 2 else 3
    wa, ba = rnd(C, cin, k0, k0, seed=1) / (cin * k0 * k0) ** 0.5, rnd(C, seed=2) * 0.1
    wb, bb = rnd(C, C, 3, 3, seed=3) / (C * 9) ** 0.5, rnd(C, seed=4) * 0.1
    x = rnd(B, cin, H, W, seed=9)
    t = F.leaky_relu(F.conv2d(x, wa, ba, padding=k0 // 2), 0.2)
    if mode == 0:
        ref = t
    else:
        y = F.conv2d(t, wb, bb, padding=1)
        ref = F.relu(y + x) if res else F.leaky_relu(y, 0.2)  # (relu after the residual: a second activation code)
    st = [dict(w=wa.to(dev()), b=ba.to(dev()), act="lrelu")]
    if mode:
        st.append(dict(w=wb.to(dev()), b=bb.to(dev()), act="relu" if res else "lrelu"))
    pr = ops.PackedRoll(st, residual=res)
    xd = x.to(dev())
    out = torch.full((B, C + 3, H, W), 5.0, device=dev())
    if c0 < cin:
        ops.conv_roll(Slice(xd, 0, c0), pr, x2=Slice(xd, c0, cin - c0), out=Slice(out, 2, C), rh=rh)
    else:
        ops.conv_roll(xd, pr, out=Slice(out, 2, C), rh=rh)
    got = out.cpu()
    assert (got[:, 2:2 + C] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert (got[:, :2] == 5.0).all() and (got[:, 2 + C:] == 5.0).all()
    # the launch granularity must not change a single bit
    out2 = torch.empty(B, C, H, W, device=dev())
    ops.conv_roll(Slice(xd, 0, c0), pr, x2=Slice(xd, c0, cin - c0) if c0 < cin else None, out=out2, rh=max(1, rh // 2 + 1))
    assert torch.equal(out2.cpu(), got[:, 2:2 + C])


def test_roll_launches_equal_tile_kernels_in_hitnet():
    """HITNetMF with the rolling-window launches (default) against the same network on the per-layer tile kernels
    (CODD_ROLL=0 path): same arithmetic per output (k-ordered fp32 fma chains), so the disparities agree to fp32
    rounding of the few layers whose chunking differs."""
    import codd_amd  # noqa: F401
    from codd_amd import configs, ops, synth
    from codd_amd.registry import build_estimator
    est = build_estimator(configs.stereo_only()).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    est = est.to(dev())
    H, W = 576, 960
    img, r_img, _ = synth.stereo_sequence(H, W, 1)
    l, r = img[:, 0].to(dev()), r_img[:, 0].to(dev())
    prev = ops.USE_ROLL
    try:
        ops.USE_ROLL = True
        a = est.stereo.stereo_matching(l, r)["pred_disp"].clone()
        ops.USE_ROLL = False
        b = est.stereo.stereo_matching(l, r)["pred_disp"].clone()
    finally:
        ops.USE_ROLL = prev
    d = (a - b).abs()
    assert d.median().item() < 1e-5 and (d > 0.25).float().mean().item() < 1e-4, (d.median().item(), d.max().item())


def test_conv_roll_random_shapes():
    """40 seeded random (mode, C, sources, batch, map size, rows per workgroup) draws of codd_conv_roll against torch
    fp32 on the CPU -- widths around the 60 / 62-column strip stride, heights around the row-block size, 1-pixel maps."""
    import random
    from codd_amd import ops
    rng = random.Random(1234)
    for it in range(40):
        mode = rng.choice([0, 1, 1, 2])
        C = rng.choice([16, 16, 32])
        cin = C if mode != 2 else rng.choice([16, 24, 32, 40, 48, 64])
        B = rng.choice([1, 1, 2])
        H = rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 47])
        W = rng.choice([1, 2, 15, 16, 17, 59, 60, 61, 62, 63, 64, 65, 119, 120, 121, 124, 125, 187])
        rh = rng.choice([1, 2, 3, 4, 7, 12, 50])
        res = mode == 1 and rng.random() < 0.5
        c0 = cin if (mode != 2 or rng.random() < 0.3) else rng.choice([c for c in (8, 16, 24, 32) if c < cin])
        if mode != 2 and rng.random() < 0.3:
            c0 = rng.choice([4, 8, 12])
        k0 = 1 if mode == 2 else 3
        wa, ba = rnd(C, cin, k0, k0, seed=it) / (cin * k0 * k0) ** 0.5, rnd(C, seed=it + 1) * 0.1
        wb, bb = rnd(C, C, 3, 3, seed=it + 2) / (C * 9) ** 0.5, rnd(C, seed=it + 3) * 0.1
        x = rnd(B, cin, H, W, seed=it + 4)
        t = F.leaky_relu(F.conv2d(x, wa, ba, padding=k0 // 2), 0.2)
        ref = t if mode == 0 else F.leaky_relu(F.conv2d(t, wb, bb, padding=1) + (x if res else 0), 0.2)
        st = [dict(w=wa.to(dev()), b=ba.to(dev()), act="lrelu")] + ([dict(w=wb.to(dev()), b=bb.to(dev()), act="lrelu")] if mode else [])
        pr = ops.PackedRoll(st, residual=res)
        xd = x.to(dev())
        if c0 < cin:
            got = ops.conv_roll(xd[:, :c0].contiguous(), pr, x2=xd[:, c0:].contiguous(), rh=rh)
        else:
            got = ops.conv_roll(xd, pr, rh=rh)
        err = (got.cpu() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (it, mode, C, cin, c0, B, H, W, rh, res, err)
