"""On-device evaluation metrics and the cross-GPU reduction.

reference: model/codd.py:435-575 (calc_metric), utils/metric.py:19-54, utils/misc.py:12-36,62-77,
utils/warp.py:69-92, utils/running_stats.py:132-183, apis/inference.py:145-154.

The reference syncs the host ~10 times per frame (``.item()``) and gathers pickled Python objects
across ranks.  Here every per-frame metric is accumulated on the device as (sum, count) pairs, a
video's 12-column row is formed once per sequence, and ranks exchange ONE ``all_reduce(SUM)`` of a
[3, 12] fp64 tensor (sum, sum of squares, non-NaN count per column) -- RCCL over xGMI on the GPU
node, gloo in the CPU tests.  This is evaluation plumbing, not the hot path: torch ops are used.
"""
import math

import torch
import torch.nn.functional as F

COLUMNS = ("epe", "th3", "tepe", "th3_tepe", "tepe_rel", "th1_tepe_rel", "flow_mag", "count",
           "epe2d_scene_flow", "epe2d_optical_flow", "1px_scene_flow", "1px_optical_flow")
BF_DEFAULT = 1050 * 0.2


def valid_mask(gt_disp, meta, gt_flow_prev=None, seg=None):
    """reference utils/misc.py:12-36 (``seg``: gt_semantic_seg > 0, i.e. the non-occluded mask of KITTI-style data)."""
    m = (gt_disp > meta["disp_range"][0]) & (gt_disp < meta["disp_range"][1])
    if seg is not None:
        m &= seg
    if gt_flow_prev is not None:
        m &= torch.sum(gt_flow_prev ** 2, dim=1, keepdim=True).sqrt() < BF_DEFAULT
    return m


def apply_seg(gt_disp, seg):
    """Ground truth with the pixels outside ``seg`` set to 0 (= invalid for any disp_range with lo >= 0): lets the
    metric kernels, whose masks are functions of the ground-truth value, honour the reference's gt_semantic_seg mask."""
    return gt_disp if seg is None else torch.where(seg, gt_disp, torch.zeros_like(gt_disp))


def temporal_mask_source(gt_disp, seg=None):
    """The map the temporal validity mask of the current frame is computed from (reference model/codd.py:478-486): the
    ground truth itself, or -- when the frame has no positive ground-truth disparity at all (KITTI provides disparity for
    one frame only) -- the constant BF_DEFAULT / 2; masked by ``seg``.  No host sync (0-dim device condition)."""
    dummy = torch.full_like(gt_disp, BF_DEFAULT / 2.0)
    return apply_seg(torch.where((gt_disp > 0.0).any(), gt_disp, dummy), seg)


def disp_change_from_disp2(gt_disp, gt_disp2):
    """reference model/codd.py:352-357: disparity change from the second-frame disparity of the data set."""
    dc = gt_disp2 - gt_disp
    bad = (gt_disp2 <= 0.0) | (gt_disp <= 0.0)
    return torch.where(bad, torch.full_like(dc, BF_DEFAULT), dc)


def disp_change_from_flow(gt_flow_occ_prev, gt_disp_prev, gt_disp_curr, gt_flow_prev):
    """reference utils/misc.py:39-59 (compute_gt_disp_change): warp the current ground truth to the previous frame with
    the ground-truth flow (nearest), difference to the previous ground truth, BF_DEFAULT where the warp leaves the image
    or the flow is occluded."""
    warped, valid = flow_warp_nearest(gt_disp_curr, gt_flow_prev)
    dc = warped - gt_disp_prev
    return torch.where(~valid | gt_flow_occ_prev, torch.full_like(dc, BF_DEFAULT), dc)


def scene_flow_sums(Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ, meta, K):
    """reference model/codd.py:519-575 restated with torch ops: the five accumulators (count, sum of 3-D end-point
    errors, sum of 2-D ones, #(3-D < 1 px), #(2-D < 1 px)) of one frame pair as a [5] fp64 tensor.
    Ts [B,h,w,7] (t, q_xyzw); pred_prev, gt_disp_prev, gt_disp_change [B,1,h,w]; gt_flow_prev [B,2,h,w];
    gt_flow_occ [B,1,h,w] bool or None.  The SE3 action is the one of the HIP kernels' oracle (oracle/se3.py)."""
    fx, fy, cx, cy = [float(v) for v in K]
    mask = valid_mask(gt_disp_prev, meta, gt_flow_prev=gt_flow_prev) & (gt_disp_change.abs() < BF_DEFAULT)
    if gt_flow_occ is not None:
        mask = mask & ~gt_flow_occ.bool()
    depth1 = torch.clip(BF_DEFAULT / pred_prev, max=BF_DEFAULT, min=0).squeeze(1)  # [B,h,w]
    h, w = depth1.shape[-2:]
    y, x = torch.meshgrid(torch.arange(h, device=depth1.device, dtype=depth1.dtype),
                          torch.arange(w, device=depth1.device, dtype=depth1.dtype), indexing="ij")
    X0 = torch.stack([depth1 * ((x - cx) / fx), depth1 * ((y - cy) / fy), depth1], -1)
    q, t = Ts[..., 3:], Ts[..., :3]
    u, wq = q[..., :3], q[..., 3:4]
    uv = 2.0 * torch.cross(u, X0, dim=-1)
    X1 = X0 + wq * uv + torch.cross(u, uv, dim=-1) + t

    def proj(X):
        Z = X[..., 2] + 1e-5
        return torch.stack([fx * (X[..., 0] / Z) + cx, fy * (X[..., 1] / Z) + cy, 1.0 / Z], -1)

    est = proj(X1) - proj(X0)
    est = torch.cat([est[..., :2], est[..., 2:] * BF_DEFAULT], -1)
    gt = torch.cat([gt_flow_prev.permute(0, 2, 3, 1), gt_disp_change.permute(0, 2, 3, 1)], -1)
    d = est - gt
    e3 = (d ** 2).sum(-1).sqrt()[mask.squeeze(1)]
    e2 = (d[..., :2] ** 2).sum(-1).sqrt()[mask.squeeze(1)]
    return torch.stack([torch.as_tensor(float(e3.numel())), e3.double().sum(), e2.double().sum(),
                        (e3 < 1.0).double().sum(), (e2 < 1.0).double().sum()]).double().to(depth1.device)


def flow_warp_nearest(img, flow):
    """reference utils/warp.py:69-92 with padding_mode='zeros', mode='nearest'."""
    B, _, H, W = img.shape
    y, x = torch.meshgrid(torch.arange(H, device=img.device, dtype=img.dtype),
                          torch.arange(W, device=img.device, dtype=img.dtype), indexing="ij")
    gx = 2 * ((x[None] + flow[:, 0]) / (W - 1)) - 1
    gy = 2 * ((y[None] + flow[:, 1]) / (H - 1)) - 1
    grid = torch.stack([gx, gy], -1)
    out = F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=True)
    valid = F.grid_sample(torch.ones_like(img[:, :1]), grid, mode="nearest", padding_mode="zeros",
                          align_corners=True) > 0.9999
    return out, valid


class _Meter:
    """Mean of per-frame means (reference AverageMeter, utils/running_stats.py:9-31), kept on device."""

    def __init__(self, device):
        self.s = torch.zeros((), device=device, dtype=torch.float64)
        self.n = torch.zeros((), device=device, dtype=torch.float64)

    def update(self, values, mask=None):
        """adds mean(values[mask]) if the mask is non-empty -- without a host sync."""
        if mask is None:
            mask = torch.ones_like(values, dtype=torch.bool)
        cnt = mask.sum().double()
        mean = (values.double() * mask).sum() / cnt.clamp(min=1)
        ok = (cnt > 0).double()
        self.s += mean * ok
        self.n += ok

    def avg(self):
        return torch.where(self.n > 0, self.s / self.n.clamp(min=1), torch.full_like(self.s, float("nan")))


class SequenceMetrics:
    """Per-video metric row (disparity + temporal columns of COLUMNS)."""

    def __init__(self, meta, device):
        self.meta = meta
        self.m = {k: _Meter(device) for k in COLUMNS[:7]}
        self.prev = None
        self.device = device
        self.sf = None  # [5] fp64 scene-flow accumulators (None until the first update)

    def update_disparity_device(self, pred, gt, crop_hw):
        """EPE / 3-px rate of one frame through the HIP metric kernel (2 launches, no torch ops, no
        host sync).  pred, gt: full padded [B,1,H,W] device tensors; crop_hw = img_shape[:2]."""
        from . import ops
        if getattr(self, "_dev_meters", None) is None:
            self._dev_meters = torch.zeros(3, device=self.device, dtype=torch.float64)
            self._dev_scratch = torch.empty(3 * 128 * pred.shape[0], device=self.device, dtype=torch.float64)
        ops.disp_metrics(pred, gt, crop_hw, self.meta["disp_range"][0], self.meta["disp_range"][1], 3.0,
                         self._dev_meters, self._dev_scratch)

    def update_temporal_device(self, pred, gt, pred_prev, gt_prev, flow_prev, crop_hw, gt_mask=None, gt2_prev=None):
        """TEPE family + flow magnitude of one frame pair through the HIP kernel (no host sync).
        flow_prev [B,2,H,W]: GT flow of the PREVIOUS frame (reference state['gt_flow'][-2]); gt_mask / gt2_prev: see
        temporal_mask_source / ops.tepe_metrics."""
        from . import ops
        if getattr(self, "_dev_tmeters", None) is None:
            self._dev_tmeters = torch.zeros(7, device=self.device, dtype=torch.float64)
            self._dev_tscratch = torch.empty(6 * 128 * pred.shape[0], device=self.device, dtype=torch.float64)
        ops.tepe_metrics(pred, gt, pred_prev, gt_prev, flow_prev, crop_hw, self.meta["disp_range"][0],
                         self.meta["disp_range"][1], BF_DEFAULT, self._dev_tmeters, self._dev_tscratch, gt_mask=gt_mask,
                         gt2_prev=gt2_prev)

    def update_scene_flow(self, Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ=None):
        """torch restatement of the reference's scene-flow block for one frame pair (cropped [h,w] maps)."""
        s = scene_flow_sums(Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ, self.meta,
                            self.meta["intrinsics"])
        self.sf = s if self.sf is None else self.sf + s

    def update_scene_flow_device(self, Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ, crop_hw):
        """The same through the HIP kernel on full padded [B,*,H,W] device maps (no host sync)."""
        from . import ops
        if self.sf is None:
            self.sf = torch.zeros(5, device=self.device, dtype=torch.float64)
            self._dev_sscratch = torch.empty(5 * 128 * pred_prev.shape[0], device=self.device, dtype=torch.float64)
        ops.sceneflow_metrics(Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change, gt_flow_occ, crop_hw,
                              self.meta["disp_range"][0], self.meta["disp_range"][1], BF_DEFAULT,
                              self.meta["intrinsics"], self.sf, self._dev_sscratch)

    def update(self, pred, gt, gt_flow=None, seg=None, gt_disp2=None):
        """pred, gt [B,1,h,w]; gt_flow [B,2,h,w] = flow from THIS frame to the next (reference
        state['gt_flow'][-2] semantics when the next frame arrives); seg: the frame's non-occluded mask (gt_disp_occ
        <= 0); gt_disp2: the data set's second-frame disparity of THIS frame (model/codd.py:350-363, 478-499)."""
        mask = valid_mask(gt, self.meta, seg=seg)
        err = (pred - gt).abs()
        self.m["epe"].update(err, mask)
        self.m["th3"].update((err > 3.0).to(err.dtype), mask)
        if self.prev is not None:
            p_pred, p_gt, p_mask, flow, p_gt2 = self.prev
            if flow is not None:
                src = torch.where((gt > 0.0).any(), gt, torch.full_like(gt, BF_DEFAULT / 2.0))  # KITTI: dummy gt
                mk = valid_mask(src, self.meta, gt_flow_prev=flow, seg=seg)
                warped, valid = flow_warp_nearest(torch.cat([gt, pred, mk.to(gt.dtype)], 1), flow)
                w_gt, w_pred, w_mask = warped[:, 0:1], warped[:, 1:2], warped[:, 2:3]
                m_curr = valid & w_mask.bool() & mk
                if p_gt2 is not None:
                    w_gt = p_gt2
                    m_curr = m_curr & (p_gt2 > 0.0)
                both = p_mask & m_curr
                d_est, d_gt = w_pred - p_pred, w_gt - p_gt
                tepe = (d_est - d_gt).abs()
                rel = tepe / (d_gt.abs() + 1e-3)
                self.m["tepe"].update(tepe, both)
                self.m["tepe_rel"].update(rel, both)
                self.m["th1_tepe_rel"].update((rel > 1.0).to(rel.dtype), both)
                self.m["th3_tepe"].update((tepe > 3.0).to(rel.dtype), both)
                self.m["flow_mag"].update(torch.sum(flow ** 2, dim=1).sqrt())
        self.prev = (pred, gt, mask, gt_flow, gt_disp2)

    def row(self):
        """[12] fp64 tensor; meter columns without data are NaN (reference nanmean semantics), the five scene-flow
        accumulators are sums (0 without data, as the reference reports them)."""
        nan = torch.full((), float("nan"), device=self.device, dtype=torch.float64)
        # scene-flow columns are raw sums in the reference (collect_metric, utils/misc.py:62-77): 0 without data
        sf = self.sf if self.sf is not None else torch.zeros(5, device=self.device, dtype=torch.float64)
        vals = [self.m[k].avg() for k in COLUMNS[:7]] + [sf[i] for i in range(5)]
        if getattr(self, "_dev_meters", None) is not None:  # HIP-kernel meters take precedence
            dm = self._dev_meters
            ok = dm[2] > 0
            vals[0] = torch.where(ok, dm[0] / dm[2].clamp(min=1), nan)
            vals[1] = torch.where(ok, dm[1] / dm[2].clamp(min=1), nan)
        if getattr(self, "_dev_tmeters", None) is not None:
            tm = self._dev_tmeters
            ok = tm[4] > 0
            for col, k in ((2, 0), (3, 1), (4, 2), (5, 3)):  # tepe, th3_tepe, tepe_rel, th1_tepe_rel
                vals[col] = torch.where(ok, tm[k] / tm[4].clamp(min=1), nan)
            vals[6] = torch.where(tm[6] > 0, tm[5] / tm[6].clamp(min=1), nan)
        return torch.stack(vals)


def reduce_rows(rows, device=None, group=None):
    """rows: list of [12] fp64 tensors (this rank's videos).  ONE all_reduce of [3,12]
    (sum, sum sq, non-NaN count) -> dict column -> (mean, std, n) with nanmean / nanvar semantics
    (reference utils/running_stats.py:176-183, apis/inference.py:145-154)."""
    import torch.distributed as dist
    if device is None:
        device = rows[0].device if rows else torch.device("cpu")
    acc = torch.zeros(3, len(COLUMNS), dtype=torch.float64, device=device)
    for r in rows:
        r = r.to(device=device, dtype=torch.float64)
        ok = ~torch.isnan(r)
        v = torch.where(ok, r, torch.zeros_like(r))
        acc[0] += v
        acc[1] += v * v
        acc[2] += ok.double()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    acc = acc.cpu()
    out = {}
    for i, k in enumerate(COLUMNS):
        n = acc[2, i].item()
        if n == 0:
            out[k] = (float("nan"), float("nan"), 0)
            continue
        mean = acc[0, i].item() / n
        var = max(acc[1, i].item() / n - mean * mean, 0.0)
        out[k] = (mean, math.sqrt(var), int(n))
    return out


def shard_videos(num_videos, rank, world_size):
    """One video per GPU, round robin (what DistributedSampler(shuffle=False) gives the reference,
    inference.py:108-115; padding duplicates are dropped instead of de-duplicated later)."""
    return list(range(rank, num_videos, world_size))
