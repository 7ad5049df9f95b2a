"""ORACLE (test infrastructure, not product): CPU fp32 restatement of Fusion.

reference model/fusion/fusion.py:41-425.  Pinned against the imported reference by
``tests/golden/make_golden.py``.
"""
import math

import torch
import torch.nn.functional as F

from .stereo import conv, warp_x


def mish(x):
    return x * torch.tanh(F.softplus(x))


def key_layer(sd, p, x):
    """reference fusion.py:74-80, 17-38."""
    t = F.relu(conv(sd, p + ".0", x))
    u = mish(conv(sd, p + ".2.conv1.0", t, 1, 1))
    u = conv(sd, p + ".2.conv2", u, 1, 1) + t
    return conv(sd, p + ".4", F.relu(u))


def px2patch(k, m, self_corr=False, P=3):
    """reference fusion.py:168-198 (+ unfold_feat :412-425; nn.Unfold(kernel P, padding P-1, dilation 2), :66-70):
    out[b, ky*P+kx, y, x] = <k[b,:,y,x], m~[b,:,y+2ky-(P-1), x+2kx-(P-1)]> / sqrt(C)  (C > 1)
                          = (k - m~) / sqrt(1)                                         (C == 1)
    zero padding; ``self_corr`` drops tap index P*P // 2 (the centre for odd P)."""
    B, C, H, W = k.shape
    mp = F.pad(m, (P - 1, P - 1, P - 1, P - 1))
    outs = []
    for ky in range(P):
        for kx in range(P):
            if self_corr and ky * P + kx == (P * P) // 2:
                continue
            sh = mp[:, :, 2 * ky:2 * ky + H, 2 * kx:2 * kx + W]
            outs.append((k - sh) if C == 1 else (k * sh).sum(1, keepdim=True))
    return torch.cat(outs, 1) / math.sqrt(C)


def disparity_confidence(pred_curr, pred_warp, fea_l, fea_r, ds=4, in_channels=24):
    """reference fusion.py:200-241."""
    o = ds // 2 - 1
    pc = pred_curr[..., o::ds, o::ds]
    pw = pred_warp[..., o::ds, o::ds]
    cw, cp = [], []
    for k in (-1, 0, 1):
        cw.append((fea_l - warp_x(fea_r, pw / ds + k)).abs().sum(1, keepdim=True) / (in_channels / 24.0))
        cp.append((fea_l - warp_x(fea_r, pc / ds + k)).abs().sum(1, keepdim=True) / (in_channels / 24.0))
    return torch.cat(cp, 1), torch.cat(cw, 1)


def input_cues(pred_curr, pred_warp, feat_curr, feat_warp, flow_warp, conf_warp, fea_l, fea_r, P=3, ds=4):
    """reference fusion.py:243-318 -> corr_feat [B,3P^2+4,H/ds,W/ds], corr_feat_fr [B,3P^2+5,H,W] (31 / 32 at P = 3)."""
    cost_curr, cost_warp = disparity_confidence(pred_curr, pred_warp, fea_l, fea_r, ds=ds, in_channels=fea_l.shape[1])
    f_cross = px2patch(feat_curr, feat_warp, P=P)
    f_self = torch.cat([px2patch(feat_curr, feat_curr, True, P), px2patch(feat_warp, feat_warp, True, P)], 1)
    d_cross = px2patch(pred_curr, pred_warp, P=P).abs()
    d_self = torch.cat([px2patch(pred_curr, pred_curr, True, P), px2patch(pred_warp, pred_warp, True, P)], 1).abs()
    corr_feat = torch.cat([f_cross, f_self, cost_curr, cost_warp], 1)
    corr_feat_fr = torch.cat([d_cross, d_self, flow_warp, (pred_warp > 0).float(), conf_warp], 1)
    return corr_feat, corr_feat_fr


def fuse(sd, p, corr_feat, pred_curr, pred_warp, feat_curr, ds=4):
    """reference fusion.py:320-355."""
    o = ds // 2 - 1
    pc = pred_curr[..., o::ds, o::ds]
    pw = pred_warp[..., o::ds, o::ds]
    corr = F.relu(conv(sd, p + ".conv_corr.2", F.relu(conv(sd, p + ".conv_corr.0", corr_feat))))
    disp = F.relu(conv(sd, p + ".conv_disp.0", torch.cat([pc, pw], 1), 1, 3))
    disp = F.relu(conv(sd, p + ".conv_disp.2", disp, 1, 1))
    mo = F.relu(conv(sd, p + ".motion_conv.0", torch.cat([corr, disp], 1), 1, 3))
    net = F.relu(conv(sd, p + ".residual_conv.0", torch.cat([feat_curr, mo, pc, pw], 1), 1, 1)) + corr
    w = torch.sigmoid(conv(sd, p + ".weight_head.1", conv(sd, p + ".weight_head.0", net, 1, 1)))
    return w.repeat_interleave(ds, 2).repeat_interleave(ds, 3)


def forget_head(sd, p, x):
    """reference fusion.py:123-132."""
    t = conv(sd, p + ".forget_head.0", x)
    t = conv(sd, p + ".forget_head.1", t, 1, 1)
    return torch.sigmoid(conv(sd, p + ".forget_head.2", t))


def memory_query(sd, outputs, state, p="fusion", patch_size=3, ds=4):
    """reference fusion.py:357-402 (mutates ``outputs``)."""
    left_feat, pred_curr = outputs["left_feat"], outputs["pred_disp"]
    feat_curr = key_layer(sd, p + ".key_layer", left_feat)
    if "memory" not in state:
        outputs["left_feat"] = feat_curr
        return
    _, feat_warp, conf_warp, pred_warp, flow_warp = state["memory"]
    corr_feat, corr_feat_fr = input_cues(pred_curr, pred_warp, feat_curr, feat_warp, flow_warp, conf_warp,
                                         outputs["left_feat"], outputs["right_feat"], P=patch_size, ds=ds)
    valid = (pred_warp > 0.0).float()
    wf = fuse(sd, p, corr_feat, pred_curr, pred_warp, feat_curr, ds=ds) * valid
    wr = forget_head(sd, p, corr_feat_fr) * valid
    outputs["pred_disp"] = pred_curr * (1 - wf * wr) + pred_warp * wf * wr
    outputs["fusion_weights"] = wf
    outputs["reset_weights"] = wr
    outputs["pred_curr"] = pred_curr
    outputs["pred_warp"] = pred_warp
    outputs["left_feat"] = feat_curr
    outputs["corr_feat"] = corr_feat  # oracle-only extras (for kernel-level parity tests)
    outputs["corr_feat_fr"] = corr_feat_fr


def memory_update(outputs, state):
    """reference fusion.py:404-410."""
    state["memory"] = [outputs["left_img"], outputs["left_feat"], outputs["pred_disp"].squeeze(1)]
