"""Ablation plug-ins of the registry surface: NullFusion, KalmanFusion, GTFusion
(reference model/fusion/others.py:8-168) and GTMotion (model/motion/others.py:11-66).  Same constructor
kwargs, same reads / writes of the ``outputs`` / ``state`` dicts; the arithmetic runs in two HIP kernels
(codd_fusion_select, codd_gt_motion)."""
import torch
import torch.nn as nn

from . import ops
from .registry import register


class _MemoryFusion(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.loss = None

    def init_weights(self, pretrained=None):
        pass

    def memory_query(self, outputs, state, *args, **kwargs):
        pass

    def memory_update(self, outputs, state, *args, **kwargs):
        state["memory"] = [outputs["left_img"], outputs["left_feat"], outputs["pred_disp"].squeeze(1)]

    def _publish(self, outputs, pred, pred_warp, fused):
        outputs["pred_disp"] = fused
        outputs["fusion_weights"] = torch.zeros_like(pred)
        outputs["reset_weights"] = torch.zeros_like(pred)
        outputs["pred_curr"] = pred
        outputs["pred_warp"] = pred_warp


@register
class NullFusion(_MemoryFusion):
    """Keeps the per-frame prediction (reference others.py:8-37)."""


@register
class KalmanFusion(_MemoryFusion):
    """reference others.py:102-168.  The reference never updates its covariance P (it is reset to zero
    and only read), so the gain is the constant Q / (Q + R); reproduced as is."""

    def __init__(self, R=1e-5, Q=1e-5, **kwargs):
        super().__init__()
        self.R, self.Q = R, Q

    def memory_query(self, outputs, state, *args, **kwargs):
        if "memory" not in state:
            return
        pred = outputs["pred_disp"].contiguous()
        pred_warp = state["memory"][3].reshape(pred.shape).contiguous()
        self._publish(outputs, pred, pred_warp, ops.fusion_select("kalman", pred, pred_warp, K=self.Q / (self.Q + self.R)))


@register
class GTFusion(_MemoryFusion):
    """reference others.py:40-99: oracle fusion that picks the estimate closer to the ground truth."""

    def memory_query(self, outputs, state, *args, **kwargs):
        if "memory" not in state:
            return
        pred = outputs["pred_disp"].contiguous()
        pred_warp = state["memory"][3].reshape(pred.shape).contiguous()
        gt = state["gt_disp"][-1].contiguous()
        self._publish(outputs, pred, pred_warp, ops.fusion_select("gt", pred, pred_warp, gt=gt))


@register
class GTMotion(nn.Module):
    """reference model/motion/others.py:11-66: the memory is moved by the ground-truth flow."""

    def __init__(self):
        super().__init__()
        self.loss = None

    def forward(self, state, outputs, img_metas=None, train_mode=False, **kwargs):
        if "memory" not in state:
            return
        img_prev, feat_prev, disp_prev = [t.contiguous() for t in state["memory"]]
        state["memory"] = ops.gt_motion(img_prev, feat_prev, disp_prev, state["gt_flow"][-1].contiguous(),
                                        state["gt_disp_change"][-1].contiguous(), state["gt_flow_occ"][-1])
        outputs["Ts"] = ops.se3_identity(img_prev.shape[0], img_prev.shape[2], img_prev.shape[3], img_prev.device)
