"""CPU tests of the evaluation plumbing: metric rows, video sharding and the single all_reduce
(gloo, world_size 2) -- the N>1 path of bench.py without GPUs."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from codd_amd import metrics


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(n, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.rand(n, 12, generator=g, dtype=torch.float64)
    rows[:, 7:] = float("nan")
    rows[0, 2] = float("nan")
    return rows


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allrows = _rows(5, 0)
    mine = [allrows[i] for i in metrics.shard_videos(5, rank, world)]
    res = metrics.reduce_rows(mine, torch.device("cpu"))
    q.put((rank, res))
    dist.destroy_process_group()


def test_shard_videos_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in metrics.shard_videos(11, r, world))
        assert seen == list(range(11))


def test_reduce_rows_single_process_matches_nanmean():
    rows = _rows(6, 3)
    res = metrics.reduce_rows(list(rows), torch.device("cpu"))
    arr = rows.numpy()
    for i, k in enumerate(metrics.COLUMNS):
        if np.isnan(arr[:, i]).all():
            assert res[k][2] == 0
        else:
            assert abs(res[k][0] - np.nanmean(arr[:, i])) < 1e-12
            assert abs(res[k][1] - np.sqrt(np.nanvar(arr[:, i]))) < 1e-9


def test_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = metrics.reduce_rows(list(_rows(5, 0)), torch.device("cpu"))
    for r in (0, 1):
        for k in metrics.COLUMNS:
            a, b = got[r][k], ref[k]
            assert a[2] == b[2]
            if b[2]:
                assert abs(a[0] - b[0]) < 1e-12 and abs(a[1] - b[1]) < 1e-9


def test_sequence_metrics_epe_tepe():
    H, W = 32, 48
    meta = dict(disp_range=(1, 210))
    sm = metrics.SequenceMetrics(meta, torch.device("cpu"))
    gt0 = torch.full((1, 1, H, W), 10.0)
    flow = torch.zeros(1, 2, H, W)
    sm.update(gt0 + 0.5, gt0, flow)
    sm.update(gt0 + 1.5, gt0, flow)
    row = sm.row()
    assert abs(row[0].item() - 1.0) < 1e-9  # mean of per-frame EPEs 0.5 and 1.5
    assert abs(row[2].item() - 1.0) < 1e-9  # TEPE: |(1.5) - (0.5)| = 1
    assert row[8].item() == 0.0  # scene-flow accumulators are sums: 0 without data (reference collect_metric)
