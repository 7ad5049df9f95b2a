"""bench.py's multi-rank control flow under gloo, world_size 2, with a stub runner (no GPU): barrier -> exactly K timed
steps -> ONE metric all-reduce -> barrier -> MAX over ranks of the elapsed time; one video per rank (weak scaling)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from codd_amd import metrics
    dev = torch.device("cpu")
    meta = dict(disp_range=(1, 210))
    seqm = metrics.SequenceMetrics(meta, dev)
    calls = dict(step=0, coll=0)
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls["coll"] += 1
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = counting_all_reduce
    gt = torch.full((1, 1, 8, 12), 20.0)

    def step(l, r):  # stub runner: rank 1 is the slow one, its prediction is off by (1 + rank) px
        calls["step"] += 1
        time.sleep(0.01 * (1 + 3 * rank))
        return gt + (1.0 + rank)

    steps = 5
    dt, red = bench.timed_region(step, lambda i: (None, None, gt), steps, lambda d, g: seqm.update(d, g), seqm.row,
                                 dev, True)
    out[rank] = dict(dt=dt, steps=calls["step"], coll=calls["coll"], epe=red["epe"])
    dist.destroy_process_group()


def test_bench_timed_region_world_size_2():
    port = 29600 + os.getpid() % 300
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = out[0], out[1]
    assert r0["steps"] == r1["steps"] == 5                      # exactly K steps on every rank
    assert r0["coll"] == r1["coll"] == 2                        # the metric all-reduce + the MAX of dt, nothing else
    assert r0["dt"] == r1["dt"] and r0["dt"] >= 5 * 0.04 * 0.9  # every rank reports the SLOWEST rank's time
    mean, std, n = r0["epe"]
    assert n == 2 and abs(mean - 1.5) < 1e-9 and abs(std - 0.5) < 1e-9 and r1["epe"] == r0["epe"]  # one video per rank


def test_bench_gpus_n_launches_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher around it: the script must re-execute itself under
    torch.distributed.run, one rank per GPU, and rank 0 must print exactly one JSON line for the whole job.  Driven on
    gloo / CPU through the hidden --dry-run-cpu mode (stub runner; same self_launch(), timed_region() and rank plumbing)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert sorted(out["config"]["ranks_seen"]) == [[0, 0, 4], [1, 1, 4]]
    # every rank runs rank 0's launch-configuration table (broadcast before the first launch) and reports its own frame rate
    assert out["config"]["launch_configurations_identical_on_all_ranks"] is True
    assert [r["rank"] for r in out["config"]["per_rank"]] == [0, 1] and all(r["fps"] > 0 for r in out["config"]["per_rank"])
    assert "broadcast to 1 rank" in out["config"]["launch_configuration_sync"]
    # launched WITH a world size that contradicts --gpus: refused, not silently run
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env2,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_rank_core_pinning_partitions_the_host():
    import bench
    before = os.sched_getaffinity(0)
    try:
        if len(before) < 2:
            return
        a = bench.pin_rank_to_cores(0, 2)
        os.sched_setaffinity(0, before)
        b = bench.pin_rank_to_cores(1, 2)
        assert a and b and not (set(a) & set(b)) and set(a) | set(b) <= set(before)
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(max(1, min(8, len(before))))


def test_hbm_algorithmic_bytes_match_survey_8d():
    """bench.py's `roofline.hbm` basis: the algorithmic bytes of the HBM-bound kernels computed from their C-ABI
    arguments must be SURVEY.md section 8(d)'s per-frame figures at 960x576 -- S3+S4 15.1 MB over the five levels,
    S6 121 MB with both hypothesis sets sharing one read of the features, M9 call 1 37.6 MB in + 22.1 MB out."""
    import bench
    spec = {k: v[1] for k, v in bench._hbm_specs().items()}
    # tile cost volume + arg-min, levels 16x ... 1x of a 576 x 960 image: args (L, R, B, C, Ht, Wt, Wr, D, ...)
    tot = sum(spec["codd_tile_costvol_argmin"]((0, 0, 1, 16, 576 // (4 << l), 960 // (4 << l), 960 // (1 << l), 320 >> l))
              for l in range(5))
    assert abs(tot / 1e6 - 15.1) < 0.15, tot
    # tile warp: level 1x..8x with two hypothesis sets, 16x with one; feature channels 16, 16, 24, 24, 32
    # args (fl, fr, B, C, Ht, Wt, hyp0, hyp1, nhyp, ...)
    tot = sum(spec["codd_tile_warp_cost"]((0, 0, 1, c, 144 >> l, 240 >> l, None, None, 1 if l == 4 else 2))
              for l, c in enumerate((16, 16, 24, 24, 32)))
    assert abs(tot / 1e6 - 121.0) < 2.5, tot  # (+ the 3-channel hypothesis reads SURVEY rounds away)
    # splat call 1: 552 960 points x (1 + 7 + 9) words in, 10 x 552 960 words out
    # args (T, depth, HT, WT, oy, ox, ds, featA, CA, featB, CB, with_flow, B, H, W, ...)
    b = spec["codd_splat"]((0, 0, 576, 960, 0, 0, 1, 0, 3, 0, 3, 1, 1, 576, 960))
    assert abs(b / 1e6 - (37.6 + 22.1)) < 0.2, b
    # convex up-sampling of the SE3 field: 19.9 MB mask + 0.2 MB data + 13.3 MB out (x 7/6 for the stored quaternion)
    b = spec["codd_cvx_upsample"]((0, 0, 1, 72, 120, 6, 1, 0))
    assert abs(b / 1e6 - (19.9 + 0.2 + 13.3)) < 0.3, b
