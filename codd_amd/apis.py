"""Evaluation loops: one video per GPU, metric rows on the device, one collective at the end.

reference apis/inference.py:16-77 (single_gpu_inference), :80-154 (multi_gpu_inference),
utils/running_stats.py:109-183 (RunningStatsWithBuffer), inference.py:108-135 (sampler / wrap).

Differences by design: the reference all_gathers a pickled stats object from
every rank and merges on rank 0; here the cross-video mean / std come from ONE ``all_reduce(SUM)``
of a [3,12] fp64 tensor (RCCL over xGMI, ``metrics.reduce_rows``) and every rank ends with the
same numbers.  The per-video CSV (file names are host strings) is gathered to rank 0 only when an
output directory is given - control-plane data, never tensors.
"""
import csv
import os
import os.path as osp
import re

import numpy as np
import torch
import torch.distributed as dist

from . import metrics


class RunningStatsWithBuffer:
    """Per-video rows keyed by file name: de-duplicated push, nan-aware mean / std, CSV dump
    (reference utils/running_stats.py:109-183)."""

    def __init__(self, path=None, header=None):
        self.path, self.header = path, header
        self.row_id_map, self.data = {}, []

    @property
    def n(self):
        return len(self.data)

    def push(self, id, value):
        if id in self.row_id_map:  # DistributedSampler padding repeats videos (running_stats.py:132-137)
            return
        self.row_id_map[id] = len(self.data)
        self.data.append(list(value) if isinstance(value, (list, tuple)) else [value])

    def __add__(self, other):
        for k, v in other.row_id_map.items():
            self.push(k, other.data[v])
        if self.header is None:
            self.header = other.header
        return self

    def _array(self):
        return np.array(self.data, dtype=np.float32).reshape(len(self.data), -1)

    @property
    def mean(self):
        return np.nanmean(self._array(), 0)

    @property
    def std(self):
        return np.sqrt(np.nanvar(self._array(), 0))

    def dump(self):
        def key(row):
            return [int(c) if c.isdigit() else c.lower() for c in re.split("([0-9]+)", str(row[0]))]

        rows = sorted(([k] + self.data[v] for k, v in self.row_id_map.items()), key=key)
        with open(self.path, "w", newline="") as f:
            csv.writer(f).writerows([self.header] + rows)


def _dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _run(model, data_loader, out_dir, show, evaluate, distributed, log=print):
    model.eval()
    module = getattr(model, "module", model)
    rank, world = _dist_info() if distributed else (0, 1)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    rs = RunningStatsWithBuffer(osp.join(out_dir, "stats.csv") if out_dir else None) if evaluate else None
    rows = []
    for data in data_loader:
        with torch.no_grad():
            result = model(return_loss=False, rescale=True, evaluate=evaluate, **data)
        if evaluate:
            rows.append(torch.cat([result[0][k] for k in metrics.COLUMNS]))
        for img_meta in data["img_metas"][0]:
            out_file = osp.join(out_dir, img_meta.get("ori_filename", "out.png")) if out_dir else None
            if show and out_file is None:
                continue
            module.show_result(img_meta.get("filename"), result, show=show, out_file=out_file, running_stats=rs)
    if not evaluate:
        return None
    device = rows[0].device if rows else None
    if device is None and distributed and world > 1:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    summary = metrics.reduce_rows(rows, device=device)  # the one data-path collective
    if distributed and world > 1 and out_dir:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(rs, gathered, dst=0)
        if rank == 0:
            merged = gathered[0]
            for other in gathered[1:]:
                merged = merged + other
            rs = merged
    if rank == 0:
        log("\n%d samples, " % max(v[2] for v in summary.values())
            + ", ".join("%s %.4f (std %.4f)" % (k, v[0], v[1]) for k, v in summary.items() if v[2] > 0))
        if out_dir and rs.header is not None:
            rs.dump()
    return summary


def single_gpu_inference(model, data_loader, out_dir=None, show=False, evaluate=False, **kwargs):
    """reference apis/inference.py:16-77.  data_loader yields dicts ``{img: [T[B,MF,3,H,W]],
    r_img: [T], img_metas: [[dict]], gt_disp: [T[B,MF,1,H,W]], gt_flow: [T[B,MF,2,H,W]]}``."""
    return _run(model, data_loader, out_dir, show, evaluate, distributed=False)


def multi_gpu_inference(model, data_loader, out_dir=None, show=False, evaluate=False, **kwargs):
    """reference apis/inference.py:80-154: every rank walks its own shard of videos
    (``shard_loader``), no tensor crosses GPUs until the final [3,12] all-reduce."""
    return _run(model, data_loader, out_dir, show, evaluate, distributed=True)


def shard_loader(videos, rank=None, world_size=None):
    """round-robin one-video-per-GPU shard of an indexable collection of data dicts
    (DistributedSampler(shuffle=False), reference inference.py:108-115, without padding)."""
    if rank is None:
        rank, world_size = _dist_info()
    return [videos[i] for i in metrics.shard_videos(len(videos), rank, world_size)]


def _apply_key_map(sd, key_map):
    """``key_map``: a callable ``name -> new name | None`` (None drops the tensor), or a sequence of
    ``(regex, replacement)`` pairs applied in order with ``re.sub`` (mmcv's ``revise_keys`` convention,
    mmcv/runner/checkpoint.py ``load_checkpoint(revise_keys=[(r'^module\\.', '')])``).  Two checkpoint names landing
    on one key is an error, not a silent overwrite."""
    import re
    if key_map is None:
        return sd
    out = {}
    for k, v in sd.items():
        if callable(key_map):
            nk = key_map(k)
        else:
            nk = k
            for pat, rep in key_map:
                nk = re.sub(pat, rep, nk)
        if nk is None:
            continue
        if nk in out:
            raise RuntimeError("key_map sends two checkpoint tensors to %r (second: %r)" % (nk, k))
        out[nk] = v
    return out


def load_checkpoint(model, filename, map_location="cpu", strict=False, log=print, key_map=None):
    """Load a published CODD ``.pth`` by key name (reference inference.py:123 -> mmcv load_checkpoint).
    Tolerates the ``state_dict`` wrapper, a ``module.`` prefix and keys that only exist for training
    (``stereo.loss.*``); BatchNorm statistics of the HRNet are folded when the packed convolutions are
    built (codd_amd/hrnet.py), so they load like any other tensor.

    ``key_map`` renames checkpoint keys before they are matched (see ``_apply_key_map``).  It exists for ONE known
    risk: the HRNet context network's parameter names (``motion.raft3d.cnet.0.*``) follow mmseg 0.x's ``HRNet``
    module tree as read from its call site (reference configs/models/codd.py:44-74) -- mmseg is not vendored in the
    reference, so those 600-odd names were never compared with a real checkpoint (tests/golden/state_dict_keys.json
    pins everything else against the imported reference).  If a published ``.pth`` reports them under
    ``missing`` / ``unexpected``, pass the remap instead of editing the module tree, e.g.
    ``key_map=[(r"^backbone\\.", "motion.raft3d.cnet.0."), (r"\\.norm(\\d)\\.", r".bn\\1.")]`` (a checkpoint whose context network sits
    under ``backbone.`` and names its BatchNorm layers ``norm1 / norm2``)."""
    ckpt = torch.load(filename, map_location=map_location)
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    sd = _apply_key_map(sd, key_map)
    own = model.state_dict()
    train_only = [k for k in sd if ".loss." in k or k.endswith("num_batches_tracked") and k not in own]
    for k in train_only:
        sd.pop(k)
    bad_shape = [k for k in sd if k in own and tuple(own[k].shape) != tuple(sd[k].shape)]
    if bad_shape:
        raise RuntimeError("checkpoint tensors with wrong shape: %s" % bad_shape[:8])
    res = model.load_state_dict(sd, strict=False)
    missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
    if strict and (missing or unexpected):
        raise RuntimeError("missing keys %s; unexpected keys %s" % (missing[:8], unexpected[:8]))
    if missing or unexpected:
        log("load_checkpoint: %d missing, %d unexpected keys" % (len(missing), len(unexpected)))
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()
    return dict(missing=missing, unexpected=unexpected, meta=ckpt.get("meta") if isinstance(ckpt, dict) else None)
