#!/usr/bin/env python
"""bench.py -- frames/sec of the full CODD forward (stereo -> motion -> fusion) at 960x540.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched with
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL); a plain ``python bench.py --gpus N``
launches itself that way (self_launch).

A "step" is ONE steady-state frame (frame index >= 1, so motion and fusion run; the reference's own
benchmark_speed.py:36-65 only ever times frame 0) of full CODD on a synthetic 960x540 stereo
sequence reflect-free padded to 960x576 (reference pipeline pads to a multiple of 64,
datasets/transforms.py:147-161), fp32, iters=16, max_disp=320, random-init weights
(deterministic filler).  Inputs are resident in HBM when the timed region starts.  Each rank
processes its own video (weak scaling; no data-path collective); the only collective is the single
all_reduce of the [3,12] metric tensor after the timed frames, inside the timed region.

Prints ONE JSON line on rank 0 (see README / task contract), including
  roofline     -- MFMA roofline of the dominant kernel family (conv_bf16_kernel, the split-bf16 convolutions of
                  RAFT3D): MFMA FLOPs ISSUED (3 bf16 MFMAs per product) by its launches of one frame /
                  their summed duration measured with HIP events on the launch stream, vs the 2.5 PFLOP/s dense
                  bf16 matrix peak; the same on ALGORITHMIC direct-conv FLOPs, and the exact-fp32 family (HITNet)
                  vs the 157.3 TFLOP/s fp32 matrix peak, are reported beside it;
                  roofline.traffic is measured by this run itself (two rocprofv3 --pmc child passes of this script,
                  FETCH_SIZE and WRITE_SIZE separately; --no-pmc-traffic quotes the committed capture instead);
  fps_two_videos_per_gpu -- two independent videos resident on the GPU (two frame graphs on two streams): throughput
                  headroom reported beside the one-video-per-GPU headline;
  cpu_baseline -- the CPU oracle (port of the reference's PyTorch-CPU path) timed on this host's
                  cores: BASELINE.json configs[0] (512x256, 2 frames, stereo only) and ONE measured
                  steady-state frame of the benchmarked configuration (no scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))


def _loaded_library():
    """the libcodd_hip.so this process loaded, relative to the repo when it is the in-tree one (CODD_LIB_AB)"""
    from codd_amd import _abi
    p = _abi.LOADED or ""
    return os.path.relpath(p, ROOT) if p.startswith(ROOT + os.sep) else p
sys.path.insert(0, ROOT)

RAW_H, RAW_W = 540, 960
PAD_H, PAD_W = 576, 960
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip-level parameters
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (same table)
HBM_PEAK_GBS = 8000.0  # HBM3E (same table)


def _hbm_specs():
    """HBM-bound kernels of the path -> ALGORITHMIC bytes of one call from its C-ABI arguments
    (fp32 words; every operand read once, every result written once)."""
    def costvol(a):  # L, R, B, C, Ht, Wt, Wr, D, ...: reads both tile-feature maps, writes (cost, d)
        B, C, Ht, Wt, Wr = a[2], a[3], a[4], a[5], a[6]
        return 4 * B * (C * Ht * Wt + C * Ht * Wr + 2 * Ht * Wt)

    def tile_warp(a):  # fl, fr, B, C, Ht, Wt, hyp0, hyp1, nhyp, ...: both feature maps once for all hypothesis sets
        B, C, Ht, Wt, nh = a[2], a[3], a[4], a[5], a[8]
        return 4 * B * (2 * C * 16 * Ht * Wt + nh * (3 + 64) * Ht * Wt)

    def lookup(a):  # ..., B, h, w, ...: 4 levels x 8x8 integer taps read, 196 features + 9 motion channels written
        B, h, w = a[7], a[8], a[9]
        return 4 * B * h * w * (4 * 64 + 196 + 9 + 3 + 7 + 2)

    def splat(a):  # T, depth, HT, WT, oy, ox, ds, featA, CA, featB, CB, with_flow, B, H, W, ...
        HT, WT, ds, CA, CB, wf, B, H, W = a[2], a[3], a[6], a[8], a[10], a[11], a[12], a[13], a[14]
        n = -(-HT // ds) * -(-WT // ds)
        C = CA + CB + (3 if wf else 0)
        return 4 * B * (n * (1 + 7 + C) + (C + 1) * H * W)

    def cvx(a):  # data, mask, B, h, w, dim, mode, out
        B, h, w, D = a[2], a[3], a[4], a[5]
        return 4 * B * h * w * (576 + D + 64 * D)

    def cvx2(a):  # T, weight, mask, B, h, w, ...: mask read once, SE3 field (7) + confidence (3) in and up-sampled out
        B, h, w = a[3], a[4], a[5]
        return 4 * B * h * w * (576 + 10 + 64 * 10)

    return {"codd_tile_costvol_argmin": ("costvol_argmin (S3+S4)", costvol), "codd_tile_warp_cost": ("tile_warp (S6)", tile_warp),
            "codd_cvx_upsample_se3_weight": ("cvx_upsample SE3 + confidence, one pass (M8)", cvx2),
            "codd_raft_geometry_lookup_xs": ("corr_lookup + geometry (M4+M5)", lookup),
            "codd_raft_geometry_lookup": ("corr_lookup + geometry (M4+M5)", lookup),
            "codd_splat": ("splat (M9: project+count+reserve+fill+gather)", splat), "codd_cvx_upsample": ("cvx_upsample (M8)", cvx)}


_T0 = time.time()


def log(msg):
    if os.environ.get("CODD_BENCH_VERBOSE"):
        print(f"[bench +{time.time() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steady-state frames (reference benchmark_speed.py:36-65 uses 200)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm", type=int, default=150,
                    help="untimed steady-state frames run before the W warm-up steps so that the GPU clocks and "
                         "the captured graph are in steady state (reported in config.prewarm_frames)")
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--precision", default="split", choices=["split", "split16", "fp32", "bf16", "bf16mix", "fp16", "fp16mix"],
                    help="arithmetic of the convolution family (codd_amd.ops.set_conv_precision): split = split-bf16 "
                         "operands / fp32 accumulate for RAFT3D and exact fp32 for HITNet / context network / Fusion (default, parity-"
                         "tested at 1e-3 px); fp32 = exact-fp32 kernels everywhere; bf16 = bf16 operands / fp32 "
                         "accumulate everywhere (BASELINE.json configs[4]); bf16mix = bf16 operands for RAFT3D's feature encoder "
                         "and update block only, exact fp32 for HITNet / context network / Fusion; fp16mix = the same policy on IEEE fp16 "
                         "operands (the reference's auto_fp16 grade); fp16 / split16 = fp16 / hi|lo-fp16 operands in EVERY stage -- "
                         "fp16's range applies there (|x| > 65504 -> inf; HITNet's channels carry raw disparities): dev modes")
    ap.add_argument("--stereo-only", action="store_true")
    ap.add_argument("--fp32-steps", type=int, default=30,
                    help="after the timed region (rank 0, N = 1, split precision only): time this many steady-state frames "
                         "with EVERY convolution on the exact-fp32 kernels and report them as fp32_exact_fps (0 = skip)")
    ap.add_argument("--two-video-steps", type=int, default=60,
                    help="after the timed region (rank 0, N = 1): frames per video of a pass with TWO independent videos "
                         "resident on the GPU (two frame graphs on two streams), reported as fps_two_videos_per_gpu -- the "
                         "throughput headroom beside the one-video-per-GPU headline (0 = skip)")
    ap.add_argument("--no-pmc-traffic", dest="pmc_traffic", action="store_false",
                    help="do not measure roofline.traffic with two rocprofv3 --pmc child passes (N = 1 only, ~40 s); the "
                         "newest committed capture under profiles/ is quoted instead")
    ap.add_argument("--tune-db", default=None,
                    help="JSON file of tuned launch configurations: loaded if it exists (no tuning launches, e.g. under "
                         "a profiler), written at the end otherwise")
    ap.add_argument("--retune", action="store_true", help="ignore the shipped codd_amd/tuned/mi355x.json and time every shape")
    ap.add_argument("--no-autotune", action="store_true",
                    help="use the heuristic conv launch configurations instead of timing the alternatives once per "
                         "layer shape during the (untimed) first frames")
    ap.add_argument("--serial-streams", action="store_true",
                    help="disable the fork/join side streams (every launch on one stream) -- used for the "
                         "rocprofv3 run whose per-kernel averages are compared with the roofline numbers")
    ap.add_argument("--height", type=int, default=PAD_H)
    ap.add_argument("--width", type=int, default=PAD_W)
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)  # tests/test_bench_dist.py: launch path on gloo / CPU with a stub runner
    return ap.parse_args()


def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: re-execute this script under
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` (one rank per GPU, rendezvous on 127.0.0.1 and a
    free port) -- what the reference's scripts/inference_dist.sh:11-12 does for inference.py:88,130-135 -- and pass the
    children's stdout (rank 0's single JSON line) and exit code through."""
    import subprocess
    # --standalone: torchrun's own c10d rendezvous on a port IT picks and holds (no bind-close-reuse race with another job
    # on the box); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run_cpu(args, rank, local, world):
    """The launch path and the timed region's control flow without a GPU (gloo, stub runner): what
    tests/test_bench_dist.py drives at N = 2 to check that ``--gpus N`` launches itself and rank 0 prints ONE line."""
    import torch.distributed as dist
    from codd_amd import metrics
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    seqm = metrics.SequenceMetrics(dict(disp_range=(1, 210)), dev)
    gt = torch.full((1, 1, 8, 12), 20.0)

    def step(l, r):
        time.sleep(0.002)
        return gt + (1.0 + rank)

    from codd_amd import ops
    ops.TUNE_DB[f"stub|rank{rank}"] = (rank, 4, 16)  # what every rank timing for itself would leave behind: N different tables
    note = sync_launch_configurations(rank, world, lambda: ops.TUNE_DB.update({"stub|settled": (1, 4, 32)}))
    dt, red = timed_region(step, lambda i: (None, None, gt), args.steps, lambda d, g: seqm.update(d, g), seqm.row, dev, True)
    mine = torch.tensor([rank, local, args.steps], dtype=torch.int64)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, dict(rank=rank, gpu=local, frames=args.steps, fps=round(args.steps / timed_region.own_seconds, 3),
                                          tune_db=tune_db_digest()))
    if rank == 0:
        print(json.dumps({"metric": "dry run (CPU stub runner, gloo)", "value": round(world * args.steps / dt, 3), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                          "epe_vs_synthetic_gt": red["epe"][0],
                          "config": {"ranks_seen": [t.tolist() for t in allr], "per_rank": per_rank, "launch_configuration_sync": note,
                                     "launch_configurations_identical_on_all_ranks": len({r["tune_db"] for r in per_rank}) == 1}}), flush=True)
    dist.destroy_process_group()


def build_model(args, device):
    import codd_amd  # noqa: F401
    from codd_amd import configs, synth
    from codd_amd.registry import build_estimator
    cfg = configs.stereo_only() if args.stereo_only else configs.codd(iters=args.iters)
    est = build_estimator(cfg).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    return est.to(device)


def conv_roofline(runner, frames, device):
    """Time every conv launch of ONE extra steady-state frame (eager) with HIP events recorded on
    the launch stream, and count its algorithmic FLOPs (2*Cin*Cout*kh*kw*Hout*Wout*B)."""
    from codd_amd import ops
    recs = []
    orig = ops._launch_conv

    def timed(lib, p, stream):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream(device))
        rc = orig(lib, p, stream)
        e.record(torch.cuda.current_stream(device))
        cout = p.Cout * (4 if p.store_mode else 1)
        recs.append((s, e, 2.0 * (p.C0 + p.C1) * cout * p.kh * p.kw * p.Hout * p.Wout * p.B,
                     (p.B, p.C0 + p.C1, cout, p.kh, p.kw, p.Hout, p.Wout, p.sy, p.store_mode),
                     p.terms if p.layout == 2 else 0))
        if p.layout == 2:  # which instantiation conv_bf16_kernel<PGW, CGW, A, B, TERMS, OUTF, KS> serves which layer
            a_ = -(-(p.nw * p.npb) // p.pgw)
            inst = "<%d, %d, %d, %d, %d, %d, %d>" % (p.pgw, p.cgw, a_, p.mb // p.cgw, p.terms, 1 if (p.xso or p.gate) else 0, 2 if p.ksplit == 2 else 1)
            layer = "%dx%d%s %d->%d @%dx%d%s" % (p.kh // 2 if p.dil2 else p.kh, p.kw, " dual-tap (dil %d + %d)" % (p.dil2, p.dil_y) if p.dil2 else
                                                  (" dil %d" % p.dil_y if p.dil_y > 1 else ""), p.C0 + p.C1, cout, p.Hout, p.Wout,
                                                  {0: "", 1: ", gate 1 (z|r pre-activation)", 2: ", gate 2 (z, r*h, q input)", 3: ", gate 3 (state update)"}[p.gate])
            th_, tw_ = p.nw, 16 * p.npb
            grid = -(-p.Hout // th_) * -(-p.Wout // tw_) * -(-cout // (16 * p.mb)) * p.B
            insts.append((inst, layer, grid, s, e))
        return rc

    orig_roll = ops._launch_roll

    def timed_roll(lib, p, stream):  # rolling-window launch (one or two layers): algorithmic FLOPs of its layers
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream(device))
        rc = orig_roll(lib, p, stream)
        e.record(torch.cuda.current_stream(device))
        cin = p.C0 + p.C1
        per_px = {0: cin * 9 * p.cout_store, 1: cin * 9 * p.C + p.C * 9 * p.cout_store, 2: cin * p.C + p.C * 9 * p.cout_store}[p.mode]
        recs.append((s, e, 2.0 * per_px * p.H * p.W * p.B, (p.B, cin, p.cout_store, 3, -10 - p.mode, p.H, p.W, 1, 0), 0))
        rolls.append(1 if p.mode == 0 else 2)
        return rc

    orig_multi = ops._launch_conv_multi

    def timed_multi(lib, params, n, stream):  # several independent small convolutions in one launch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream(device))
        rc = orig_multi(lib, params, n, stream)
        e.record(torch.cuda.current_stream(device))
        fl = sum(2.0 * params[i].C0 * params[i].Cout * params[i].kh * params[i].kw * params[i].Hout * params[i].Wout * params[i].B
                 for i in range(n))
        p0 = params[0]
        recs.append((s, e, fl, (p0.B, p0.C0, p0.Cout, p0.kh, p0.kw, p0.Hout, p0.Wout, -n, 0), 0))
        multis.append(n)
        return rc

    insts = []
    rolls = []
    ops._launch_roll = timed_roll
    multis = []
    ops._launch_conv_multi = timed_multi
    ops._launch_conv = timed
    serial_before = ops.Fork.serial
    ops.Fork.serial = True  # one launch at a time, so that every event pair brackets exactly one kernel
    # the HBM-bound kernels of the same frame: event brackets around their C-ABI entry points
    from codd_amd import _abi
    lib = _abi.load()
    hbm_recs, saved = [], {}
    for fname, (label, nbytes) in _hbm_specs().items():
        fn = getattr(lib, fname)
        saved[fname] = fn

        def wrapped(*a, _fn=fn, _label=label, _nbytes=nbytes):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(torch.cuda.current_stream(device))
            rc = _fn(*a)
            e_.record(torch.cuda.current_stream(device))
            hbm_recs.append((_label, s_, e_, float(_nbytes(a))))
            return rc

        setattr(lib, fname, wrapped)
    try:
        l, r = frames
        runner.eager_frame_on_static_state(l, r)
        torch.cuda.synchronize(device)
    finally:
        ops._launch_conv = orig
        ops._launch_roll = orig_roll
        ops._launch_conv_multi = orig_multi
        ops.Fork.serial = serial_before
        for fname, fn in saved.items():
            setattr(lib, fname, fn)
    hbm = {}
    for label, s_, e_, nb in hbm_recs:
        c = hbm.setdefault(label, [0, 0.0, 0.0])
        c[0] += 1; c[1] += s_.elapsed_time(e_); c[2] += nb
    hbm = {k: dict(calls_per_frame=v[0], ms_per_frame=round(v[1], 4), algorithmic_mb_per_frame=round(v[2] / 1e6, 2),
                   achieved_gbs=round(v[2] / 1e6 / v[1], 1), frac_of_hbm_peak=round(v[2] / 1e6 / v[1] / HBM_PEAK_GBS, 4))
           for k, v in hbm.items() if v[1] > 0}
    t_ms = sum(r[0].elapsed_time(r[1]) for r in recs)
    flops = sum(r[2] for r in recs)
    fam = {}  # kernel family -> [launches, ms, algorithmic flop, issued MFMA flop]
    for s, e, f, _, terms in recs:
        c = fam.setdefault("split_bf16" if terms == 3 else "split_fp16" if terms == 48 else "bf16" if terms == 1 else "fp16" if terms == 16 else "fp32", [0, 0.0, 0.0, 0.0])
        c[0] += 1; c[1] += s.elapsed_time(e); c[2] += f; c[3] += f * (3 if terms in (3, 48) else 1)
    if os.environ.get("CODD_BENCH_VERBOSE"):  # per-shape table (dev aid): count, total ms, TFLOP/s
        by = {}
        for s, e, f, key, terms in recs:
            key = key + (terms,)
            c = by.setdefault(key, [0, 0.0, 0.0])
            c[0] += 1; c[1] += s.elapsed_time(e); c[2] += f
        for key, (n, ms, f) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            log("conv B%d Cin%-4d Cout%-4d k%dx%d out %3dx%-3d s%d m%d terms%d : n=%3d  %7.3f ms  %6.1f us/launch  %5.1f TF"
                % (*key, n, ms, ms / n * 1e3, f / ms / 1e9))
    by_inst = {}
    for inst, layer, grid, s_, e_ in insts:
        c = by_inst.setdefault(inst, {}).setdefault(layer, [0, 0.0, grid])
        c[0] += 1; c[1] += s_.elapsed_time(e_)
    by_inst = {k: [dict(layer=l, launches_per_frame=v[0], us_per_launch=round(v[1] / v[0] * 1e3, 1), workgroups=v[2]) for l, v in ls.items()]
               for k, ls in sorted(by_inst.items())}
    return dict(launches=len(recs), time_ms=t_ms, gflop=flops / 1e9, hbm=hbm, instantiations=by_inst,
                roll_launches=len(rolls), roll_layers=sum(rolls), multi_launches=len(multis), multi_jobs=sum(multis),
                families={k: dict(launches=v[0], ms=round(v[1], 3), gflop=round(v[2] / 1e9, 2),
                                  issued_gflop=round(v[3] / 1e9, 2)) for k, v in fam.items()})


def pmc_traffic(args, kernel_pattern):
    """HBM-side traffic per launch of the kernels whose name contains ``kernel_pattern``, measured with the PMC counters
    as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (--kernel-trace --pmc only),
    gfx950 correction traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE counts wide reads at half size).  Each pass
    re-runs this script for 4 eager, serial-stream frames.  -> (bytes per launch, source text) or (None, None)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, None  # no profiler, or this process already runs under one
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "p", "--",
                       sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--fp32-steps", "0", "--two-video-steps", "0",
                       "--no-pmc-traffic", "--serial-streams", "--no-graph", "--steps", "4", "--prewarm", "2", "--warmup", "1",
                       "--precision", args.precision, "--iters", str(args.iters), "--height", str(args.height),
                       "--width", str(args.width)] + (["--stereo-only"] if args.stereo_only else [])
                env = dict(os.environ, TMPDIR="/tmp")
                for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                    env.pop(k, None)
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
                path = None
                for root, _, files in os.walk(td):
                    for f in files:
                        if f.endswith("counter_collection.csv"):
                            path = os.path.join(root, f)
                tot, n = 0.0, 0
                for r in csv.DictReader(open(path)):
                    if r["Counter_Name"] == counter and kernel_pattern in r["Kernel_Name"]:
                        tot += float(r["Counter_Value"]); n += 1
                if n == 0:
                    return None, None
                sums[counter] = (tot, n)
    except Exception as e:  # pragma: no cover  (profiler missing / refused / timed out: the committed capture is quoted)
        log(f"pmc traffic passes failed: {e!r}")
        return None, None
    f, nf = sums["FETCH_SIZE"]
    w, nw = sums["WRITE_SIZE"]
    return round((2.0 * f / nf + w / nw) * 1024.0), (
        f"measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate child passes of bench.py, "
        f"4 eager serial-stream frames, {nf} launches of *{kernel_pattern}*); traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB "
        f"(gfx950: FETCH_SIZE counts wide reads at half size)")


def cpu_baseline(args):
    """Oracle (CPU port of the reference's PyTorch path), MEASURED, no scaling:
      (a) BASELINE.json configs[0]: HITNetMF stereo-only, 2-frame 512x256 sequence;
      (b) one steady-state frame (frame 1: motion + fusion run) of the benchmarked configuration at its full
          size (960x576, iters=16) -- this is `value`.
    min(host cores, --cpu-threads) threads (the 256-hardware-thread GPU host stalls torch's CPU pool when all of
    them are requested)."""
    from codd_amd import configs, synth
    from codd_amd.registry import build_estimator
    from oracle import codd as oc
    cores = max(1, min(os.cpu_count() or 1, args.cpu_threads))
    torch.set_num_threads(cores)
    # (a) configs[0]
    est = build_estimator(configs.stereo_only()).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    sd = est.state_dict()
    img, r_img, _ = synth.stereo_sequence(256, 512, 2)
    with torch.no_grad():
        t0 = time.perf_counter()
        for f in range(2):
            oc.frame(sd, img[:, f], r_img[:, f], {}, None, with_motion=False, with_fusion=False)
        dt1 = time.perf_counter() - t0
    # (b) the benchmarked configuration, one steady-state frame
    h, w = args.height, args.width
    est = build_estimator(configs.stereo_only() if args.stereo_only else configs.codd(iters=args.iters)).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    sd = est.state_dict()
    img, r_img, _ = synth.stereo_sequence(h, w, 2)
    intr = (1050.0, 1050.0, 480.0, 270.0)
    state = {}
    with torch.no_grad():
        oc.frame(sd, img[:, 0], r_img[:, 0], state, intr, iters=args.iters, with_motion=not args.stereo_only,
                 with_fusion=not args.stereo_only)
        t0 = time.perf_counter()
        oc.frame(sd, img[:, 1], r_img[:, 1], state, intr, iters=args.iters, with_motion=not args.stereo_only,
                 with_fusion=not args.stereo_only)
        dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 5), unit="frames/s", cores=cores, kind="port",
                sample=f"1 steady-state frame (frame 1) of the CPU oracle at the benchmarked configuration {w}x{h}, "
                       f"iters={args.iters}: {dt:.2f} s on {cores} threads (measured, not scaled); "
                       f"BASELINE.json configs[0] (HITNetMF stereo-only, 2 frames 512x256): {dt1:.2f} s = "
                       f"{2.0 / dt1:.3f} frames/s on the same threads",
                configs0_frames_per_s=round(2.0 / dt1, 4))


def cpu_baseline_subprocess(args, timeout=420):
    """Run the CPU leg in a child process with a hard wall-clock bound so it can never stall the bench."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--iters", str(args.iters),
           "--cpu-threads", str(args.cpu_threads), "--height", str(args.height), "--width", str(args.width)] + (
               ["--stereo-only"] if args.stereo_only else [])
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="frames/s", cores=args.cpu_threads, kind="port",
                    sample=f"CPU oracle sample did not finish within {timeout} s")
    except Exception as e:  # pragma: no cover
        return dict(value=None, unit="frames/s", cores=args.cpu_threads, kind="port", sample=f"failed: {e!r}")


def timed_region(step, frame_fn, steps, update_metric, metric_row, device, use_dist):
    """The contract's timed region: barrier + device sync on both sides of EXACTLY ``steps`` steps, the job's single
    collective (the all_reduce of the [3,12] metric tensor, codd_amd.metrics.reduce_rows) inside it, and the MAX over
    ranks of the elapsed time.  ``step(l, r) -> disparity``; ``frame_fn(i) -> (l, r, gt)``; ``update_metric(d, gt)``
    accumulates on the device without a host sync.  Device-agnostic so that tests/test_bench_dist.py can drive it
    under gloo with a stub runner.  Returns (seconds, reduced metric dict)."""
    import torch.distributed as dist
    from codd_amd import metrics

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        l, r, g = frame_fn(i)
        d = step(l, r)
        update_metric(d, g)  # on-device EPE meters: 2 HIP launches, no sync
    red = metrics.reduce_rows([metric_row()], device)  # the job's only collective (RCCL all_reduce)
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    timed_region.own_seconds = dt  # this rank's own clock (the JSON line lists every rank's frames/s beside the MAX-based value)
    if use_dist:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    return dt, red


def sync_launch_configurations(rank, world, settle):
    """Every rank must run the SAME launch configurations (tile, chunk depth = fp32 summation order): layer shapes the shipped
    db does not know are timed on the fly, and N ranks timing for themselves can settle on N different picks (VERDICT r5 item
    9).  Rank 0 runs ``settle()`` (which meets and times every layer shape), then its table replaces every other rank's
    BEFORE they launch anything.  Returns a note for the JSON line."""
    import torch.distributed as dist
    from codd_amd import ops
    if rank == 0:
        settle()
    picks = [{k: list(v) for k, v in ops.TUNE_DB.items()} if rank == 0 else None]
    dist.broadcast_object_list(picks, src=0)
    if rank != 0:
        ops.TUNE_DB.clear()
        ops.TUNE_DB.update({k: tuple(v) for k, v in picks[0].items()})
    return f"rank 0's {len(picks[0])} launch configurations broadcast to {world - 1} rank(s) before their first launch"


def tune_db_digest():
    import hashlib
    from codd_amd import ops
    return hashlib.sha256(json.dumps({k: list(v) for k, v in ops.TUNE_DB.items()}, sort_keys=True).encode()).hexdigest()[:12]


def pin_rank_to_cores(local_rank, local_world):
    """Give every rank its own contiguous slice of the host cores (the slice of a GPU's own NUMA node when the node
    exposes one domain per GPU pair, which contiguous numbering does on the 2-socket MI355X hosts) and cap the
    intra-op thread pools: N launcher processes otherwise all spin on the same cores."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // max(1, local_world))
        mine = cores[local_rank * per:(local_rank + 1) * per] or cores
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
        return mine
    except (AttributeError, OSError):  # pragma: no cover (non-Linux)
        return None


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run -> RCCL even at N = 1
    if world != args.gpus and "RANK" in os.environ:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: one rank per GPU is the contract")
    if args.gpus > 1 and "RANK" not in os.environ:  # plain `python bench.py --gpus N`: become the launcher
        raise SystemExit(self_launch(args))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, local, world)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pin_rank_to_cores(local, max(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))))

    from codd_amd import metrics, synth
    from codd_amd.runtime import FrameRunner

    from codd_amd import ops as _ops_tune
    # shipped = codd_amd/tuned/mi355x.json: launch configurations found by the autotuner on an MI355X for the layer
    # shapes of this workload and committed with the library (like a find-db); shapes it does not know are still
    # timed on the fly.  --retune ignores it, --tune-db replaces it.
    _ops_tune.enable_autotune(not args.no_autotune, shipped=not args.retune and not args.tune_db)
    if args.tune_db and os.path.exists(args.tune_db):
        _ops_tune.load_tune_db(args.tune_db)
    _ops_tune.set_conv_precision(args.precision)
    est = build_model(args, device)
    if args.serial_streams:
        from codd_amd import ops as _ops
        _ops.Fork.serial = True
    H, W = args.height, args.width
    MF = 6  # distinct synthetic frames, cycled (frame t+1 = frame t translated by a sub-pixel flow)
    img, r_img, gt = synth.stereo_sequence(H, W, MF, flow=(0.75 + 0.125 * rank, 0.25))  # a different video per rank
    img, r_img, gt = img.to(device), r_img.to(device), gt.to(device)
    raw_h, raw_w = (RAW_H, RAW_W) if (H, W) == (PAD_H, PAD_W) else (H, W)
    metas = synth.default_metas(H, W, img_shape=(raw_h, raw_w, 3))
    runner = FrameRunner(est, metas[0], use_graph=not args.no_graph)
    step = runner.step

    def frame(i):
        k = i % MF
        return img[:, k].contiguous(), r_img[:, k].contiguous(), gt[:, k]

    tune_note = None
    if use_dist and not args.no_autotune:  # (also at N = 1 under torch.distributed.run: the RCCL object broadcast is exercised wherever a launcher is)
        def settle():  # rank 0 meets every layer shape of the steady-state frame in two eager frames
            r0 = FrameRunner(est, metas[0], use_graph=False)
            for i in range(2):
                r0.step(*frame(i)[:2])
            torch.cuda.synchronize(device)
        tune_note = sync_launch_configurations(rank, world, settle)
        log(tune_note)
    # frame 0 primes the recurrent state; then W untimed warm-up frames (graph capture happens here)
    log("model built, inputs resident")
    l, r, _ = frame(0)
    step(l, r)
    torch.cuda.synchronize(device)
    log("frame 0 done")
    for i in range(1, 1 + args.prewarm):
        l, r, _ = frame(i)
        step(l, r)
    torch.cuda.synchronize(device)
    # the W warm-up steps are a dress rehearsal of the timed region (same loop, metric kernels, metric row, collective)
    # on throw-away meters: everything the timed region touches for the first time -- torch's reduction / copy kernels
    # of the metric row, the communicator -- is paged in and initialised here (on a freshly booted box the first
    # process otherwise pays ~100 ms of code-object paging inside the timed region: 74 vs 80 frames/s at K = 100)
    warm = metrics.SequenceMetrics(metas[0][0], device)
    timed_region(step, lambda i: frame(1 + args.prewarm + i), max(args.warmup, 1),
                 lambda d, g: warm.update_disparity_device(d, g, (raw_h, raw_w)), warm.row, device, use_dist)
    log(f"{args.prewarm} pre-warm + {max(args.warmup, 1)} warm-up frames done")
    seqm = metrics.SequenceMetrics(metas[0][0], device)

    dt, red = timed_region(step, lambda i: frame(1 + args.prewarm + args.warmup + i), args.steps,
                           lambda d, g: seqm.update_disparity_device(d, g, (raw_h, raw_w)), seqm.row, device, use_dist)

    log(f"timed region done: {dt:.3f} s")
    tuned_for_headline = len(_ops_tune.AUTOTUNE_LOG)  # layer signatures of THIS workload that the shipped db did not know (timed in frames 0-1)
    # every rank reports in: (rank, local GPU index, frames it timed) gathered over RCCL -> ranks_seen in the JSON line
    ranks_seen = [[rank, local, args.steps]]
    per_rank = [dict(rank=rank, gpu=local, frames=args.steps, fps=round(args.steps / timed_region.own_seconds, 3), tune_db=tune_db_digest())]
    if use_dist:
        mine = torch.tensor([rank, local, args.steps], device=device, dtype=torch.int64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = [t.tolist() for t in allr]
        objs = [None] * world
        dist.all_gather_object(objs, per_rank[0])
        per_rank = objs
    if args.tune_db and rank == 0:  # (re)write: shapes met for the first time in this run were tuned on the fly
        _ops_tune.save_tune_db(args.tune_db)
    if os.environ.get("CODD_BENCH_VERBOSE"):
        for r in sorted(_ops_tune.AUTOTUNE_LOG, key=lambda r: -((r[2] or 0) - r[4])):
            if r[1] != r[3]:
                log("autotune %-34s heuristic %s %.1f us -> %s %.1f us" % (r[0], r[1], r[2] or -1, r[3], r[4]))
    roof = None
    cpu = None
    fp32_fps = None
    two_fps = None
    batched_fps = None
    if rank == 0:
        try:
            l, r, _ = frame(1)
            cr = conv_roofline(runner, (l, r), device)
            fams = cr["families"]
            # dominant family = where the frame's MFMA work is (the eager event brackets of this pass over-count the
            # launch-bound small fp32 layers: ~8 us of host launch gap each; profiles/r0N_kernel_stats_serial.md hold the rocprofv3 durations of every round)
            dom = max(fams, key=lambda k_: fams[k_]["issued_gflop"])
            fd = fams[dom]
            peak = FP32_MATRIX_PEAK_TFLOPS if dom == "fp32" else BF16_MATRIX_PEAK_TFLOPS
            ach = fd["gflop"] / fd["ms"]  # GFLOP/ms = TFLOP/s of ALGORITHMIC direct-convolution work
            issued = fd["issued_gflop"] / fd["ms"]  # MFMA work issued (3 bf16 MFMAs per product on the split path)
            # `traffic` needs the PMC counters of a separate rocprofv3 pass and cannot be measured from inside this
            # process: the figure of the newest committed capture is quoted with its source, else null
            traffic, tsrc = None, None
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            if world == 1 and args.pmc_traffic and not args.serial_streams:
                # measured by THIS run: two child passes of this script under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE
                # need separate passes on gfx950), 4 eager serial-stream frames each, ~20 s per pass, hard timeouts
                pat = {"split_bf16": "conv_bf16_kernel", "split_fp16": "conv_bf16_kernel", "bf16": "conv_bf16_kernel", "fp16": "conv_bf16_kernel", "fp32": "conv_"}[dom]
                traffic, tsrc = pmc_traffic(args, pat)
                log(f"pmc traffic passes done: {traffic}")
            for tname in ([] if traffic is not None else sorted((f for f in os.listdir(pdir) if f.endswith("_conv_traffic.json")), reverse=True)):
                tj = json.load(open(os.path.join(pdir, tname)))
                if tj.get("family") == dom:
                    traffic, tsrc = round(tj["traffic_bytes_per_launch"]), f"profiles/{tname} (committed rocprofv3 --pmc passes, not this run): " + tj["correction"]
                    break
            kern = {"split_bf16": "conv_bf16_kernel<*, TERMS=3> (split-bf16: 3 bf16 MFMAs per product, fp32 accumulate)",
                    "split_fp16": "conv_bf16_kernel<*, TERMS=48> (split-fp16: 3 fp16 MFMAs per product on 22-bit operands, fp32 accumulate)",
                    "bf16": "conv_bf16_kernel<*, TERMS=1> (bf16 operands, fp32 accumulate)",
                    "fp16": "conv_bf16_kernel<*, TERMS=16> (IEEE fp16 operands on v_mfma_f32_16x16x32_f16, fp32 accumulate)",
                    "fp32": "conv_mfma_kernel<*> + conv_quad_kernel<*> (exact fp32 MFMA)"}[dom]
            roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                        issued_tflops=round(issued, 2), issued_frac=round(issued / peak, 4),
                        traffic=traffic, traffic_unit="bytes/launch (PMC, separate rocprofv3 passes)", traffic_source=tsrc,
                        kernel=kern, basis="ALGORITHMIC direct-convolution FLOPs (2 Cin Cout kh kw Hout Wout) of the family's "
                        "launches of one frame / their summed HIP-event durations; issued_* counts the MFMA work actually "
                        "issued (x3 bf16 MFMAs per product on the split path)",
                        launches_per_frame=fd["launches"], ms_per_frame=fd["ms"],
                        algorithmic_frac_of_fp32_matrix_peak=round(ach / FP32_MATRIX_PEAK_TFLOPS, 4),
                        families=fams, conv_launches_per_frame=cr["launches"],
                        # which layer each conv_bf16_kernel<PGW, CGW, A, B, TERMS, OUTF, KS> instantiation serves in this frame
                        # (launches, HIP-event microseconds per launch, workgroups per launch): the key to the rocprofv3 / PMC tables
                        instantiations=cr["instantiations"],
                        rolling_window_launches_per_frame=cr["roll_launches"], conv_layers_inside_rolling_launches=cr["roll_layers"],
                        multi_job_launches_per_frame=cr["multi_launches"], convs_inside_multi_job_launches=cr["multi_jobs"], conv_gflop_per_frame=round(cr["gflop"], 2),
                        conv_ms_per_frame=round(cr["time_ms"], 3),
                        whole_conv_algorithmic_tflops=round(cr["gflop"] / cr["time_ms"], 2),
                        # the HBM-bound kernels of the same frame against SURVEY.md 8(d)'s algorithmic bytes and the 8 TB/s peak
                        hbm=dict(peak_gbs=HBM_PEAK_GBS, basis="algorithmic bytes of the call (operands read once, results "
                                 "written once) / HIP-event duration of the call in the serial eager frame",
                                 kernels=cr["hbm"]))
        except Exception as e:  # pragma: no cover
            roof = dict(error=repr(e))
        log(f"roofline pass done: {roof}")
        if world == 1 and args.precision in ("split", "split16") and args.fp32_steps > 0 and not args.stereo_only:
            # secondary figure: the same frame with every convolution on the exact-fp32 MFMA kernels (own graph)
            try:
                prev = _ops_tune.set_conv_precision("fp32")
                r32 = FrameRunner(est, metas[0], use_graph=not args.no_graph)
                for i in range(12):  # frame 0 primes the state, capture, clocks
                    r32.step(*frame(i)[:2])
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for i in range(args.fp32_steps):
                    r32.step(*frame(12 + i)[:2])
                torch.cuda.synchronize(device)
                fp32_fps = round(args.fp32_steps / (time.perf_counter() - t0), 3)
                del r32
            except Exception as e:  # pragma: no cover
                fp32_fps = repr(e)
            finally:
                _ops_tune.set_conv_precision(prev)
            log(f"fp32-exact pass done: {fp32_fps}")
        if world == 1 and args.two_video_steps > 0 and not args.stereo_only and not args.no_graph:
            # throughput headroom, reported BESIDE the headline (one video per GPU): two independent videos resident on
            # this GPU -- two model replicas (own persistent buffers), two captured frame graphs replayed on two streams
            try:
                est2 = build_model(args, device)
                runners = [runner, FrameRunner(est2, metas[0], use_graph=True)]
                runners[1].reset()
                streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]

                def both(i):
                    for rr, ss in zip(runners, streams):
                        with torch.cuda.stream(ss):
                            rr.step(*frame(i)[:2])

                for ss in streams:
                    ss.wait_stream(torch.cuda.current_stream(device))
                for i in range(20):  # primes the second video's state, captures its graph, settles the clocks
                    both(i)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for i in range(args.two_video_steps):
                    both(20 + i)
                torch.cuda.synchronize(device)
                two_fps = round(2 * args.two_video_steps / (time.perf_counter() - t0), 3)
                del runners, est2
            except Exception as e:  # pragma: no cover
                two_fps = repr(e)
            log(f"two-videos pass done: {two_fps}")
            # the same two videos through the SAME launches (B = 2, lock-step: one frame graph whose every kernel sees both
            # videos) -- what a per-launch fixed cost would amortise over; the headline stays one video per GPU
            # (reference inference.py:109-110).  tests/test_gpu_headline_parity.py::test_two_videos_in_lock_step holds each
            # video of the batch to its B = 1 run.
            try:
                img_b, r_img_b, _ = synth.stereo_sequence(H, W, MF, flow=(0.875, 0.25))
                img_b, r_img_b = img_b.to(device), r_img_b.to(device)
                ins2 = [(torch.cat([img[:, k], img_b[:, k]], 0).contiguous(), torch.cat([r_img[:, k], r_img_b[:, k]], 0).contiguous())
                        for k in range(MF)]
                rb = FrameRunner(est, metas[0], use_graph=True)
                for i in range(20):
                    rb.step(*ins2[i % MF])
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for i in range(args.two_video_steps):
                    rb.step(*ins2[(20 + i) % MF])
                torch.cuda.synchronize(device)
                batched_fps = round(2 * args.two_video_steps / (time.perf_counter() - t0), 3)
                del rb, ins2
            except Exception as e:  # pragma: no cover
                batched_fps = repr(e)
            log(f"two-videos lock-step pass done: {batched_fps}")
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_subprocess(args)
            log("cpu baseline done")
    if rank == 0:
        fps = world * args.steps / dt
        out = {
            "metric": ("frames/sec HITNetMF stereo-only forward @960x540 (whole job)" if args.stereo_only else
                       "frames/sec full CODD forward @960x540 (whole job)"),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"split": "f32 (split-bf16 MFMA operands, f32 accumulate; HITNet + context network + Fusion exact f32)", "fp32": "f32",
                      "split16": "f32 (split-fp16 MFMA operands: 22-bit hi + lo, three MFMAs per product, f32 accumulate; HITNet + context network + Fusion exact f32)",
                      "bf16": "bf16 (MFMA operands; f32 accumulate, f32 everywhere outside the convolutions)",
                      "bf16mix": "bf16 MFMA operands / f32 accumulate for RAFT3D's encoder + update block; HITNet + context "
                                 "network + Fusion exact f32",
                      "fp16": "f16 (MFMA operands; f32 accumulate, f32 everywhere outside the convolutions)",
                      "fp16mix": "f16 MFMA operands / f32 accumulate for RAFT3D's encoder + update block (the reference's auto_fp16 "
                                 "precision); HITNet + context network + Fusion exact f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("HITNetMF stereo-only" if args.stereo_only else
                                    "full CODD (HITNetMF + Motion/RAFT3D iters=%d + Fusion)" % args.iters) +
                                   f" {raw_w}x{raw_h} padded to {W}x{H}, max_disp=320, one video per GPU, "
                                   "steady-state frames (idx>=1), synthetic stereo sequence, random-init weights",
                       "conv_precision": args.precision,
                       "hip_graph": bool(runner.graph is not None), "frames_per_gpu": args.steps,
                       "prewarm_frames": args.prewarm, "side_streams": not args.serial_streams,
                       "conv_autotune": ("off" if args.no_autotune else "%d layer signatures in the table; %d of the headline workload's were "
                                         "not in the shipped db and timed in its first two frames, %d more in the secondary runs "
                                         "(fp32 / two-video passes)" % (
                                             len(_ops_tune.TUNE_DB), tuned_for_headline, len(_ops_tune.AUTOTUNE_LOG) - tuned_for_headline)),
                       "fps_per_gpu": round(fps / world, 3),
                       "library": _loaded_library(),
                       "per_rank": per_rank,  # every rank's own frames/s (its own clock) and the digest of its launch-configuration table
                       "launch_configurations_identical_on_all_ranks": len({r["tune_db"] for r in per_rank}) == 1,
                       "launch_configuration_sync": tune_note or "single rank",
                       "ranks_seen": ranks_seen, "frames_timed_all_ranks": sum(r[2] for r in ranks_seen)},
            "epe_vs_synthetic_gt": red["epe"][0],
            # every convolution on the exact-fp32 kernels (--precision fp32), same frame, shorter run
            "fp32_exact_fps": fp32_fps,
            "fps_two_videos_per_gpu": two_fps,
            # two videos as ONE batch through the same launches (B = 2 lock-step), frames/s of both videos together
            "fps_two_videos_batched": batched_fps,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
