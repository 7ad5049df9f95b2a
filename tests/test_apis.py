"""CPU tests of the evaluation harness (codd_amd.apis): stats buffer, CSV dump, checkpoint loader and
the world_size-2 gloo path of multi_gpu_inference with a stand-in estimator (the GPU estimator itself
is covered by the -m gpu tests)."""
import csv
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from codd_amd import apis, metrics


class _FakeEstimator(torch.nn.Module):
    """Returns a deterministic metric row per video; show_result mirrors the estimator's."""

    def forward(self, img=None, img_metas=None, return_loss=False, evaluate=True, **kw):
        vid = img_metas[0][0]["vid"]
        row = torch.arange(12, dtype=torch.float64) + 100.0 * vid
        row[7:] = float("nan")
        if vid == 1:
            row[2] = float("nan")
        return [{k: row[i:i + 1] for i, k in enumerate(metrics.COLUMNS)}]

    def show_result(self, filename, result, show=False, out_file=None, running_stats=None, **kw):
        from codd_amd.estimator import ConsistentOnlineDynamicDepth
        return ConsistentOnlineDynamicDepth.show_result(self, filename, result, show, out_file, running_stats)


def _videos(n):
    return [dict(img=[torch.zeros(1)], img_metas=[[dict(vid=i, filename="seq_%d.png" % i, ori_filename="seq_%d.png" % i)]])
            for i in range(n)]


def _expected(n):
    rows = np.stack([np.arange(12, dtype=np.float64) + 100.0 * v for v in range(n)])
    rows[:, 7:] = np.nan
    rows[1, 2] = np.nan
    return rows


def test_running_stats_dedup_sort_and_dump(tmp_path):
    rs = apis.RunningStatsWithBuffer(str(tmp_path / "s.csv"), header=["filename", "a", "b"])
    rs.push("v10", [1.0, float("nan")])
    rs.push("v2", [3.0, 4.0])
    rs.push("v10", [100.0, 100.0])  # duplicate id is ignored
    assert rs.n == 2
    assert np.allclose(rs.mean, [2.0, 4.0]) and np.allclose(rs.std, [1.0, 0.0])
    rs.dump()
    table = list(csv.reader(open(tmp_path / "s.csv")))
    assert [r[0] for r in table] == ["filename", "v2", "v10"]  # natural sort


def test_single_gpu_inference_loop_and_csv(tmp_path):
    out = apis.single_gpu_inference(_FakeEstimator(), _videos(3), out_dir=str(tmp_path), evaluate=True)
    exp = _expected(3)
    for i, k in enumerate(metrics.COLUMNS):
        if i >= 7:
            assert out[k][2] == 0
        else:
            assert abs(out[k][0] - np.nanmean(exp[:, i])) < 1e-9
    table = list(csv.reader(open(tmp_path / "stats.csv")))
    assert table[0] == ["filename"] + list(metrics.COLUMNS) and len(table) == 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vids = apis.shard_loader(_videos(5))
    res = apis.multi_gpu_inference(_FakeEstimator(), vids, out_dir=out_dir, evaluate=True)
    q.put((rank, len(vids), res))
    dist.destroy_process_group()


def test_multi_gpu_inference_gloo_world2(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    assert sorted(g[1] for g in got) == [2, 3]
    exp = _expected(5)
    for _, _, res in got:  # every rank holds the same global summary
        for i, k in enumerate(metrics.COLUMNS[:7]):
            assert abs(res[k][0] - np.nanmean(exp[:, i])) < 1e-9
            assert abs(res[k][1] - np.sqrt(np.nanvar(exp[:, i]))) < 1e-6
            assert res[k][2] == (4 if i == 2 else 5)
    table = list(csv.reader(open(tmp_path / "stats.csv")))
    assert [r[0] for r in table[1:]] == ["seq_%d.png" % i for i in range(5)]


def test_load_checkpoint_by_key_name(tmp_path):
    from codd_amd import configs, synth
    from codd_amd.registry import build_estimator
    src = build_estimator(configs.codd(iters=2))
    synth.load_synthetic_weights(src, gain=1.4)
    sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    sd["module.stereo.loss.convx.weight"] = torch.zeros(1, 1, 3, 3)  # training-only keys of the published files
    sd["module.stereo.loss.convy.weight"] = torch.zeros(1, 1, 3, 3)
    path = str(tmp_path / "codd.pth")
    torch.save(dict(state_dict=sd, meta=dict(epoch=1)), path)
    dst = build_estimator(configs.codd(iters=2))
    info = apis.load_checkpoint(dst, path, strict=True, log=lambda *a: None)
    assert info["missing"] == [] and info["unexpected"] == [] and info["meta"]["epoch"] == 1
    a, b = src.state_dict(), dst.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    bad = dict(sd)
    k0 = next(k for k in bad if k.endswith("weight") and bad[k].dim() == 4)
    bad[k0] = bad[k0][:, :1]
    torch.save(dict(state_dict=bad), path)
    with pytest.raises(RuntimeError):
        apis.load_checkpoint(build_estimator(configs.codd(iters=2)), path)
