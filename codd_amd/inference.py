"""Command-line launcher: ``python -m codd_amd.inference <checkpoint> --img-dir L --r-img-dir R ...``

reference inference.py:12-135 (arguments, dist init, dataloader, checkpoint, DP/DDP wrap) and
datasets/custom_stereo_mf.py (folder of frames -> one multi-frame sample), configs/datasets/custom.py
(pseudo intrinsics / calib / disp range).  One process per GPU (``--launcher pytorch`` under torchrun,
RCCL); each rank takes every world_size-th video; images are decoded on the host (PIL), uploaded as
uint8 and normalised + reflect-padded on the GPU by ``codd_preprocess``.
"""
import argparse
import os
import os.path as osp
import re

import numpy as np
import torch

from . import apis, configs, ops
from .registry import build_estimator

CUSTOM = dict(intrinsics=[640, 360, 1050, 1050], calib=210, disp_range=(1, 210))  # configs/datasets/custom.py:4-7


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="CODD inference on MI355X")
    p.add_argument("checkpoint", nargs="?", default=None, help="published .pth (omit: synthetic weights)")
    p.add_argument("--iters", type=int, default=16, help="RAFT3D update iterations (configs/models/motion.py)")
    p.add_argument("--stereo-only", action="store_true")
    p.add_argument("--img-dir", help="left frames, or a directory of per-video sub-directories")
    p.add_argument("--r-img-dir", help="right frames, same layout as --img-dir")
    p.add_argument("--img-suffix", default=".png")
    p.add_argument("--num-frames", type=int, default=-1, help="frames per sample (reference caps at 50)")
    p.add_argument("--show", action="store_true", help="write <name>.disp.pred.npz per video")
    p.add_argument("--show-dir", default="./work_dirs/output")
    p.add_argument("--eval", action="store_true", help="needs --disp-dir with .npy ground truth")
    p.add_argument("--disp-dir")
    p.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay of the frame")
    p.add_argument("--no-autotune", action="store_true", help="static launch heuristics (bit-reproducible runs)")
    return p.parse_args(argv)


def _natural(names):
    return sorted(names, key=lambda s: [int(c) if c.isdigit() else c.lower() for c in re.split("([0-9]+)", s)])


def list_videos(img_dir, r_img_dir, suffix):
    """[(name, [left paths], [right paths])]: sub-directories are videos, else the directory is one."""
    subs = _natural([d for d in os.listdir(img_dir) if osp.isdir(osp.join(img_dir, d))])
    out = []
    for name, ld, rd in ([(s, osp.join(img_dir, s), osp.join(r_img_dir, s)) for s in subs] or
                         [(osp.basename(osp.normpath(img_dir)), img_dir, r_img_dir)]):
        lf = _natural([f for f in os.listdir(ld) if f.endswith(suffix)])
        rf = _natural([f for f in os.listdir(rd) if f.endswith(suffix)])
        assert len(lf) == len(rf) and lf, "left / right frame lists differ in %s" % name
        out.append((name, [osp.join(ld, f) for f in lf], [osp.join(rd, f) for f in rf]))
    return out


def _load_rgb(path, device):
    from PIL import Image
    arr = np.array(Image.open(path).convert("RGB"))
    return ops.preprocess(torch.from_numpy(np.ascontiguousarray(arr)).to(device), bgr=False)


def make_sample(name, lefts, rights, device, num_frames=-1, disp_paths=None):
    """One data dict in the layout the estimator's forward_test expects (datasets/formating.py:65-85)."""
    if num_frames > 0:
        lefts, rights = lefts[:num_frames], rights[:num_frames]
    img = torch.stack([_load_rgb(p, device)[0] for p in lefts])[None]
    r_img = torch.stack([_load_rgb(p, device)[0] for p in rights])[None]
    from PIL import Image
    w, h = Image.open(lefts[0]).size
    H, W = img.shape[-2:]
    meta = dict(filename=lefts[0], ori_filename=name + ".png", ori_shape=(h, w, 3), img_shape=(h, w, 3),
                pad_shape=(H, W, 3), **CUSTOM)
    data = dict(img=[img], r_img=[r_img], img_metas=[[meta]])
    if disp_paths:
        gt = torch.zeros(1, len(lefts), 1, H, W, device=device)
        for i, p in enumerate(disp_paths[:len(lefts)]):
            gt[0, i, 0, :h, :w] = torch.from_numpy(np.load(p).astype(np.float32)).to(device)
        data["gt_disp"] = [gt]
    return data


def main(argv=None):
    args = parse_args(argv)
    distributed = args.launcher != "none"
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")  # RCCL on ROCm
    device = torch.device("cuda", torch.cuda.current_device())
    model = build_estimator(configs.stereo_only() if args.stereo_only else configs.codd(iters=args.iters))
    if args.checkpoint:
        apis.load_checkpoint(model, args.checkpoint, map_location="cpu")
    else:
        from . import synth
        print("no checkpoint given: synthetic weights (outputs are meaningless, timing is not)")
        synth.load_synthetic_weights(model, gain=1.4)
    model = model.to(device).eval()
    model.use_graph = not args.no_graph  # steady-state frames by hipGraph replay (estimator.inference / FrameRunner)
    ops.enable_autotune(not args.no_autotune)  # time the conv launch configurations once per layer shape
    videos = list_videos(args.img_dir, args.r_img_dir, args.img_suffix)
    mine = apis.shard_loader(videos) if distributed else videos

    def loader():
        for name, lf, rf in mine:
            dp = None
            if args.eval:
                assert args.disp_dir, "--eval needs --disp-dir"
                dd = osp.join(args.disp_dir, name) if osp.isdir(osp.join(args.disp_dir, name)) else args.disp_dir
                dp = [osp.join(dd, f) for f in _natural([f for f in os.listdir(dd) if f.endswith(".npy")])]
            yield make_sample(name, lf, rf, device, args.num_frames, dp)

    fn = apis.multi_gpu_inference if distributed else apis.single_gpu_inference
    res = fn(model, loader(), args.show_dir, show=args.show, evaluate=args.eval)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
