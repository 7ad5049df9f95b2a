"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's ablation
plug-ins -- KalmanFusion / GTFusion (model/fusion/others.py:40-168) and GTMotion
(model/motion/others.py:11-66, utils/warp.py:69-92).  Pinned against the imported reference by
tests/golden (ablation_* arrays)."""
import torch
import torch.nn.functional as F


def kalman_fuse(pred, pred_warp, R=1e-5, Q=1e-5):
    """model/fusion/others.py:124-153.  The reference never updates P (it stays 0), so the gain is the
    constant Q / (Q + R)."""
    K = Q / (Q + R)
    fused = pred_warp + K * (pred - pred_warp)
    fused = torch.where(pred_warp <= 0.0, pred, fused)
    return torch.where((pred_warp - pred).abs() > 1, pred, fused)


def gt_fuse(pred, pred_warp, gt):
    """model/fusion/others.py:54-86.  gt [B,1,hg,wg] is zero-padded to the prediction's size."""
    h, w = pred.shape[-2:]
    gt = F.pad(gt, (0, w - gt.shape[-1], 0, h - gt.shape[-2]))
    d = (pred - gt).abs() - (pred_warp - gt).abs()
    fused = torch.where(d < -1, pred, torch.where(d > 1, pred_warp, (pred + pred_warp) / 2))
    fused = torch.where(pred_warp <= 0.0, pred, fused)
    return torch.where(gt > 0.0, fused, pred)


def _warp_nearest(img, flow):
    """utils/warp.py:69-92 with mode='nearest', padding_mode='zeros' -> (warped, valid)."""
    B, _, H, W = img.shape
    y, x = torch.meshgrid(torch.arange(H, dtype=img.dtype), torch.arange(W, dtype=img.dtype), indexing="ij")
    gx = 2 * ((x[None] + flow[:, 0]) / (W - 1)) - 1
    gy = 2 * ((y[None] + flow[:, 1]) / (H - 1)) - 1
    grid = torch.stack([gx, gy], -1)
    out = F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=True)
    valid = F.grid_sample(torch.ones_like(img), grid, mode="nearest", padding_mode="zeros", align_corners=True) > 0.9999
    return out, valid


def gt_motion(img_prev, feat_prev, disp_prev, gt_flow, gt_disp_change, gt_flow_occ):
    """model/motion/others.py:17-61 -> the 5-entry memory [img_warp, feat_warp, confidence, disp_warp, flow3]."""
    h, w = disp_prev.shape[-2:]
    pad = (0, w - gt_flow.shape[-1], 0, h - gt_flow.shape[-2])
    gt_flow, gt_dc, occ = F.pad(gt_flow, pad), F.pad(gt_disp_change, pad), F.pad(gt_flow_occ.float(), pad) > 0
    to_warp, valid = _warp_nearest(torch.cat([img_prev, disp_prev.unsqueeze(1)], 1), gt_flow)
    to_warp = torch.where(valid & ~occ, to_warp, torch.zeros_like(to_warp))
    disp_warp = to_warp[:, -1:] - gt_dc
    disp_warp = torch.where(valid[:, :1] & ~occ, disp_warp, torch.zeros_like(disp_warp))
    # quarter-resolution features are moved by the FULL-resolution flow sampled at [2::4, 2::4] (unscaled)
    feat_warp, fvalid = _warp_nearest(feat_prev, gt_flow[:, :, 2::4, 2::4])
    feat_warp = torch.where(fvalid, feat_warp, torch.zeros_like(feat_warp))
    flow3 = torch.cat([gt_flow, gt_dc], 1)
    return [to_warp[:, :3], feat_warp, torch.ones_like(flow3), disp_warp, flow3]
