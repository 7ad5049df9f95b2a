"""ORACLE (test infrastructure, not product): the per-frame CODD driver on CPU.

reference model/codd.py:80-126 (consistent_online_depth_estimation) and :290-398 (inference
loop, evaluate=False branch).
"""
import torch

from . import fusion as ofusion
from . import motion as omotion
from . import stereo as ostereo


def frame(sd, left, right, state, intrinsics, max_disp=320, iters=16, with_motion=True, with_fusion=True,
          trace=None):
    """One frame of stereo -> motion -> fusion.  Mutates ``state``; returns ``outputs``."""
    outputs = ostereo.stereo_matching(sd, left, right, max_disp)
    if with_motion:
        omotion.motion_forward(sd, state, outputs, intrinsics, iters, trace=trace)
    if with_fusion:
        ofusion.memory_query(sd, outputs, state)
        ofusion.memory_update(outputs, state)
    return outputs


def inference(sd, img, r_img, img_metas, max_disp=320, iters=16, with_motion=True, with_fusion=True):
    """img, r_img [B,MF,3,H,W] -> disparities [B,MF,h,w] cropped to img_shape."""
    state = {}
    meta = img_metas[0][0] if isinstance(img_metas[0], (list, tuple)) else img_metas[0]
    ih, iw = meta["img_shape"][:2]
    outs = []
    with torch.no_grad():
        for l, r in zip(torch.unbind(img, 1), torch.unbind(r_img, 1)):
            o = frame(sd, l, r, state, meta.get("intrinsics"), max_disp, iters, with_motion, with_fusion)
            outs.append(o["pred_disp"][:, :, :ih, :iw])
    return torch.cat(outs, 1)
