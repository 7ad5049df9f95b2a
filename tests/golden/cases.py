"""Deterministic inputs / weights shared by the golden generator and the oracle tests (data only:
closed-form textures and seeded torch generators -- no reference code)."""
import torch

from codd_amd import configs, synth

# H >= 128: with a single tile row (H = 64) the reference normalises y by (Ht - 1) = 0 and samples NaN
# coordinates (initialization.py:27), i.e. its own cost volume degenerates to the zero-padded value.
STEREO_SIZES = {"s128": (128, 192), "s192": (192, 256)}
_SD = None


def state_dict():
    """Synthetic weights for every parameter of the full CODD model (reference key names)."""
    global _SD
    if _SD is None:
        import codd_amd  # noqa: F401
        from codd_amd.registry import build_estimator
        est = build_estimator(configs.codd())
        _SD = synth.fill_state_dict(est.state_dict(), gain=1.4)
    return _SD


def stereo_pair(H, W):
    img, r_img, _ = synth.stereo_sequence(H, W, 1, dmax=24.0)
    return img[:, 0], r_img[:, 0]


def image(H, W):
    return stereo_pair(H, W)[0]


def _gen(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda *s: torch.randn(*s, generator=g)


def fusion_case(H=64, W=128):
    R = _gen(100)
    pred = (R(1, 1, H, W) * 5 + 20).abs()
    pw = (R(1, 1, H, W) * 5 + 20).abs()
    pw[:, :, 10:20, 30:50] = 0  # holes: pred_warp == 0
    out = dict(left_feat=R(1, 24, H // 4, W // 4), right_feat=R(1, 24, H // 4, W // 4), pred_disp=pred,
               left_img=R(1, 3, H, W))
    mem = [R(1, 3, H, W), R(1, 32, H // 4, W // 4), torch.sigmoid(R(1, 3, H, W)), pw, R(1, 3, H, W)]
    return out, dict(memory=mem)


def fusion_p5_state_dict():
    """The fusion.* weights for corr_cfg.patch_size = 5: the two layers that read the cue tensors get
    3*25+4 = 79 and 3*25+5 = 80 input channels (reference fusion.py:82-126); everything else as state_dict()."""
    sd = {k: v for k, v in state_dict().items() if k.startswith("fusion.")}
    R = _gen(150)
    sd["fusion.conv_corr.0.weight"] = R(64, 79, 1, 1) * (1.4 / 79 ** 0.5)
    sd["fusion.forget_head.0.weight"] = R(16, 80, 1, 1) * (1.4 / 80 ** 0.5)
    return sd


def update_inputs(h=8, w=16):
    R = _gen(200)
    return R(1, 128, h, w), R(1, 384, h, w), R(1, 196, h, w), R(1, h, w, 2), R(1, h, w, 6), R(1, h, w, 1)


def fmaps(h=16, w=24):
    R = _gen(300)
    return R(1, 128, h, w), R(1, 128, h, w)


def cvx_inputs(h=8, w=16):
    R = _gen(400)
    return R(1, h, w, 6), R(1, 576, h, w)


def proj_inputs(h=8, w=16):
    g = torch.Generator().manual_seed(500)
    depth = torch.rand(1, h, w, generator=g) * 50 + 1
    K = torch.tensor([[100.0, 110.0, 8.0, 4.0]])
    coords = torch.rand(1, h, w, 2, generator=g) * torch.tensor([20.0, 10.0]) - 2
    return depth, K, coords


def warp_inputs(h=16, w=32):
    R = _gen(600)
    return R(1, 24, h, w), R(1, 1, h, w).abs() * 4


def metric_case(H=128, W=192, MF=3, h=120, w=180):
    """Stereo sequence + ground truth for the reference's calc_metric (EPE / TEPE family)."""
    img, r_img, disp = synth.stereo_sequence(H, W, MF, dmax=24.0)
    R = _gen(700)
    gt = disp.clone() + 0.3 * R(1, MF, 1, H, W)
    gt[:, :, :, 20:30, 40:80] = 0.0  # invalid ground truth (below disp_range[0])
    gt[:, 1, :, 60:70, 100:120] = 250.0  # above disp_range[1]
    flow = 2.5 * R(1, MF, 2, H, W)
    flow[:, :, :, 90:100, 10:30] = 300.0  # |flow| >= BF_DEFAULT: excluded by compute_valid_mask
    meta = [dict(img_shape=(h, w, 3), disp_range=(1, 210), intrinsics=[100.0, 100.0, W / 2.0, H / 2.0])]
    return img, r_img, gt, flow, meta


def kitti_metric_case():
    """metric_case() with KITTI-style ground truth (reference model/codd.py:350-363, 478-499): disparity for the first
    frame of every pair only (frame 1 has none at all -> the dummy-mask branch), second-frame disparity ``gt_disp2``
    (with holes) and an occlusion map ``gt_disp_occ`` (> 0 = occluded)."""
    img, r_img, gt, flow, meta = metric_case()
    R = _gen(750)
    gt = gt.clone()
    gt[:, 1] = 0.0
    gt2 = (gt + 0.8 * R(*gt.shape)).clamp(min=0.0)
    gt2[:, 0] = (gt[:, 0] + 0.8 * R(*gt[:, 0].shape)).abs()
    gt2[:, :, :, 50:60, 60:90] = 0.0
    occ = (R(*gt.shape) > 1.0).float()
    return img, r_img, gt, flow, gt2, occ, meta


def sceneflow_case(H=64, W=96, MF=3, h=60, w=90):
    """Inputs of the reference's calc_metric INCLUDING its scene-flow block (model/codd.py:519-575): per-frame
    predictions, ground truth (disparity, flow, disparity change, flow occlusion) and a dense SE3 field."""
    from oracle import se3
    R = _gen(900)
    pred = (R(1, MF, 1, H, W) * 4 + 30).abs() + 1.0
    pred[:, 1, :, 3:6, 5:9] = 0.0  # zero disparity: depth = BF / 0 -> clipped to BF
    gt = pred + 0.5 * R(1, MF, 1, H, W)
    gt[:, :, :, 10:14, 20:40] = 0.0
    flow = 2.0 * R(1, MF, 2, H, W)
    flow[:, :, :, 40:44, 10:20] = 300.0
    dchange = 0.7 * R(1, MF, 1, H, W)
    dchange[:, :, :, 30:33, 50:70] = 400.0  # |disp change| >= BF_DEFAULT: excluded
    occ = R(1, MF, 1, H, W) > 1.2
    Ts = se3.exp(0.02 * R(1, MF, H, W, 6) + torch.tensor([0.05, -0.03, 0.1, 0.0, 0.0, 0.0]))
    meta = dict(img_shape=(h, w, 3), disp_range=(1, 210), intrinsics=[80.0, 82.0, W / 2.0, H / 2.0])
    return dict(pred=pred, gt=gt, flow=flow, dchange=dchange, occ=occ, Ts=Ts, meta=meta, h=h, w=w)


def ablation_case(H=32, W=48, hg=30, wg=44):
    """Inputs of the GT / Kalman ablation plug-ins (model/fusion/others.py, model/motion/others.py)."""
    R = _gen(800)
    pred = (R(1, 1, H, W) * 3 + 12).abs()
    warp = pred + 0.8 * R(1, 1, H, W)
    warp[:, :, 5:9, 7:15] = 0.0  # holes
    gt = (pred + R(1, 1, H, W))[:, :, :hg, :wg].contiguous()
    gt[:, :, 12:15, 20:30] = 0.0
    img_prev, feat_prev, disp_prev = R(1, 3, H, W), R(1, 32, H // 4, W // 4), (R(1, H, W) * 3 + 12).abs()
    gt_flow = 3.0 * R(1, 2, hg, wg)
    gt_dc = 0.5 * R(1, 1, hg, wg)
    occ = (R(1, 1, hg, wg) > 1.0)
    return dict(pred=pred, warp=warp, gt=gt, img_prev=img_prev, feat_prev=feat_prev, disp_prev=disp_prev,
                gt_flow=gt_flow, gt_disp_change=gt_dc, gt_flow_occ=occ, left_feat=R(1, 24, H // 4, W // 4))


def cfg1_sequence():
    """BASELINE.json configs[0]: 2-frame 512x256 synthetic stereo pair sequence."""
    img, r_img, _ = synth.stereo_sequence(256, 512, 2, dmax=40.0)
    return img, r_img
