"""Fusion on MI355X (reference model/fusion/fusion.py:41-425).

The reference's "fusion" has no GRU: it is key projection, pixel-to-patch
correlation cues, two small conv heads and a blend.  The cue tensors are produced by two fused
kernels (csrc/fusion.hip); the heads run on the MFMA conv family.
"""

import torch
import torch.nn as nn

from . import ops
from .ops import Slice
from .registry import build_loss, register
from .stereo import cv

PREFETCH_KEY = True  # (A/B switch; Fusion.prefetch_key)
FUSE_FORGET = True  # (A/B switch: Fusion.memory_query forget branch)


class BasicBlock(nn.Module):
    """reference fusion.py:17-38 (Mish)."""

    def __init__(self, c1, c2, s, p, d):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(c1, c2, 3, s, d if d > 1 else p, d), nn.Mish(inplace=True))
        self.conv2 = nn.Conv2d(c2, c2, 3, 1, d if d > 1 else p, d)


@register
class Fusion(ops.RuntimeState, nn.Module):
    def __init__(self, in_channels, fusion_channel, loss=None, corr_cfg=dict(), ds_scale=4):
        super().__init__()
        self.loss = build_loss(loss) if loss is not None else None
        self.fusion_channel, self.ds_scale, self.in_channels = fusion_channel, ds_scale, in_channels
        self.patch_size = corr_cfg.get("patch_size", 3)
        if self.patch_size not in (3, 5) or ds_scale < 2 or ds_scale % 2:
            raise NotImplementedError("the HIP cue kernels are instantiated for patch_size 3 and 5 and even ds_scale "
                                      "(configs/models/codd.py:82-86 uses 3 and 4)")
        fc = fusion_channel
        p2 = self.patch_size ** 2  # cue channels (reference fusion.py:82-87): cross p2, self 2 (p2 - 1), stereo cost 6
        self.key_layer = nn.Sequential(nn.Conv2d(in_channels, fc, 1), nn.ReLU(inplace=True),
                                       BasicBlock(fc, fc, s=1, p=1, d=1), nn.ReLU(inplace=True), nn.Conv2d(fc, fc, 1))
        self.conv_corr = nn.Sequential(nn.Conv2d(2 * (p2 - 1) + p2 + 6, fc * 2, 1), nn.ReLU(inplace=True),
                                       nn.Conv2d(fc * 2, fc, 1), nn.ReLU(inplace=True))
        self.conv_disp = nn.Sequential(nn.Conv2d(2, fc, 7, padding=3), nn.ReLU(inplace=True),
                                       nn.Conv2d(fc, fc, 3, padding=1), nn.ReLU(inplace=True))
        self.motion_conv = nn.Sequential(nn.Conv2d(fc * 2, fc - 2, 7, padding=3), nn.ReLU(inplace=True))
        self.weight_head = nn.Sequential(nn.Conv2d(fc, fc, 3, padding=1), nn.Conv2d(fc, 1, 1), nn.Identity(),
                                         nn.Sigmoid())
        self.forget_head = nn.Sequential(nn.Conv2d(6 + 2 * (p2 - 1) + p2 + 1, 16, 1), nn.Conv2d(16, 8, 3, padding=1),
                                         nn.Conv2d(8, 1, 1), nn.Identity(), nn.Sigmoid())
        self.residual_conv = nn.Sequential(nn.Conv2d(fc * 2, fc, 3, padding=1), nn.ReLU(inplace=True))

    def forget_matrix(self):
        """forget_head (1x1 nc->16, 3x3 16->8, 1x1 8->1: no non-linearity in between, reference fusion.py:123-132) merged
        into one 3x3 convolution of the cue map, as codd_fusion_forget takes it: [W_eff 9 x nc | beta 9 | c0] fp32,
        products formed in fp64; cached per parameter version."""
        m0, m1, m2 = self.forget_head[0], self.forget_head[1], self.forget_head[2]
        ver = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in (m0, m1, m2))
        c = self.__dict__.get("_codd_forget_matrix")
        if c is None or c[0] != ver:
            W0, b0 = m0.weight.detach().double()[:, :, 0, 0], m0.bias.detach().double()      # [16, nc], [16]
            W1, b1 = m1.weight.detach().double(), m1.bias.detach().double()                  # [8, 16, 3, 3], [8]
            w2, b2 = m2.weight.detach().double()[0, :, 0, 0], m2.bias.detach().double()[0]   # [8]
            A = torch.einsum("o,oikl->kli", w2, W1).reshape(9, -1)                           # [9, 16]: w2 . W1[:, :, k]
            weff = torch.cat([(A @ W0).reshape(-1), A @ b0, (w2 @ b1 + b2).reshape(1)]).float().contiguous()
            c = self.__dict__["_codd_forget_matrix"] = (ver, weff)
        return c[1]

    def prefetch_key(self, left_feat):
        """key_layer(left_feat) (reference fusion.py:74-80, 361) depends on the stereo network only: issued on a side
        stream BEFORE the motion stage so that its four small launches run beside the update loop instead of between
        the loop and the cue kernels; memory_query joins it."""
        if ops.Fork.serial or not PREFETCH_KEY:
            return
        dev = left_feat.device
        side = self.__dict__.get("_kside")
        if side is None or side.device != dev:
            side = self.__dict__["_kside"] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), ops.stage("fusion"):
            self.__dict__["_pending"] = (left_feat, self._key(left_feat), side)

    def _key(self, x):
        """reference fusion.py:74-80."""
        k = self.key_layer
        t = cv(k[0], x, act="relu")
        u = cv(k[2].conv1[0], t, act="mish")
        u = cv(k[2].conv2, u, res1=t, act="relu")  # relu(block(t)) : key_layer[3] fused
        return cv(k[4], u)

    def memory_query(self, outputs, state, *args, **kwargs):
        """reference fusion.py:357-402."""
        with ops.stage("fusion"):  # conv precision of the stage (ops._STAGE_PRECISION)
            return self._memory_query(outputs, state, *args, **kwargs)

    def _memory_query(self, outputs, state, *args, **kwargs):
        left_feat, pred_curr = outputs["left_feat"], outputs["pred_disp"]
        pend = self.__dict__.pop("_pending", None)
        if pend is not None:  # projected beside the motion stage (prefetch_key): always join the side stream
            torch.cuda.current_stream(left_feat.device).wait_stream(pend[2])
        if pend is not None and pend[0] is left_feat:
            feat_curr = pend[1]
        else:
            feat_curr = self._key(left_feat)
        if "memory" not in state:
            outputs["left_feat"] = feat_curr
            return
        _, feat_warp, conf_warp, pred_warp, flow_warp = [t.contiguous() for t in state["memory"]]
        fea_l, fea_r = outputs["left_feat"].contiguous(), outputs["right_feat"].contiguous()
        B, _, H, W = pred_curr.shape
        fc = self.fusion_channel
        # [mo (fc-2) | pc | pw] : second half of residual_conv's input (reference fuse(), :343-346)
        ds = self.ds_scale
        tail = torch.empty(B, fc, H // ds, W // ds, device=pred_curr.device, dtype=torch.float32)
        # the full-resolution forget-head chain (reference fusion.py:123-132) only needs the warped state: it runs
        # on a side stream beside the quarter-resolution cue / weight-head chain
        if getattr(self, "_fk", None) is None or self._fk.dev != pred_curr.device:
            self._fk = ops.Fork(pred_curr.device, 1)

        def forget_chain():
            if FUSE_FORGET and self.forget_head[1].kernel_size == (3, 3) and self.forget_head[1].padding == (1, 1):
                # cues + the (linear) forget head + sigmoid as ONE kernel: the 32-channel full-resolution cue tensor
                # (70.8 MB at 960x576) and the 16- / 8-channel maps are never materialised
                return ops.fusion_forget(pred_curr, pred_warp, flow_warp, conf_warp, self.forget_matrix(),
                                         patch=self.patch_size)
            cues_fr = ops.fusion_cues_fr(pred_curr, pred_warp, flow_warp, conf_warp, patch=self.patch_size)
            t = cv(self.forget_head[1], cv(self.forget_head[0], cues_fr))
            return cv(self.forget_head[2], t, act="sigmoid")

        wr = self._fk.run(0, forget_chain)
        corr_feat = ops.fusion_cues_lr(pred_curr, pred_warp, feat_curr, feat_warp, fea_l, fea_r, Slice(tail, fc - 2, 2),
                                       patch=self.patch_size, ds=ds)
        corr = cv(self.conv_corr[2], cv(self.conv_corr[0], corr_feat, act="relu"), act="relu")
        disp = cv(self.conv_disp[0], Slice(tail, fc - 2, 2), act="relu")
        disp = cv(self.conv_disp[2], disp, act="relu")
        cv(self.motion_conv[0], corr, x2=disp, act="relu", out=Slice(tail, 0, fc - 2))
        net = cv(self.residual_conv[0], feat_curr, x2=tail, act="relu", post=corr)
        wf_lr = cv(self.weight_head[1], cv(self.weight_head[0], net), act="sigmoid")
        self._fk.join()
        fused, wf, wr = ops.fusion_blend(pred_curr, pred_warp, wf_lr, wr, self.ds_scale)
        outputs["pred_disp"] = fused
        outputs["fusion_weights"] = wf
        outputs["reset_weights"] = wr
        outputs["pred_curr"] = pred_curr
        outputs["pred_warp"] = pred_warp
        outputs["left_feat"] = feat_curr

    def memory_update(self, outputs, state, *args, **kwargs):
        """reference fusion.py:404-410."""
        state["memory"] = [outputs["left_img"], outputs["left_feat"], outputs["pred_disp"].squeeze(1)]
