"""Dev tool: how data-dependent is the Gauss-Newton builder's time?

Builds libcodd_hip.so three times on the GPU box (plain, -DGN_STATS, -DGN_NO_SKIP), runs the benchmarked configuration
(960x576, iters 16, synthetic weights / frames of bench.py) eagerly and reports
  * the fraction of (wave, neighbour) visits the `every lane's affinity < 1e-9` test skips, and
  * the HIP-event duration of codd_se3_gn_step_heads (record packing + builder + solve) with the skip on / off.
    python tools/gn_skip_rate.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    import torch
    from codd_amd import _abi, configs, ops, synth
    from codd_amd.registry import build_estimator
    from codd_amd.runtime import FrameRunner
    lib = _abi.load()
    est = build_estimator(configs.codd(iters=16)).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    est = est.cuda()
    H, W = 576, 960
    img, r_img, _ = synth.stereo_sequence(H, W, 6, flow=(0.75, 0.25))
    img, r_img = img.cuda(), r_img.cuda()
    metas = synth.default_metas(H, W, img_shape=(540, 960, 3))
    ops.enable_autotune(True, shipped=True)
    ops.Fork.serial = True
    runner = FrameRunner(est, metas[0], use_graph=False)
    recs = []
    fn = lib.codd_se3_gn_step_heads

    def timed(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*a)
        e.record()
        recs.append((s, e))
        return rc

    for f in range(3):
        runner.step(img[:, f].contiguous(), r_img[:, f].contiguous())
    torch.cuda.synchronize()
    if tag == "stats":
        lib.codd_gn_stats.argtypes = [C.c_void_p, C.c_int]
        lib.codd_gn_stats(None, 1)
    lib.codd_se3_gn_step_heads = timed
    for f in range(3, 6):
        runner.step(img[:, f].contiguous(), r_img[:, f].contiguous())
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in recs]
    print(f"[{tag}] codd_se3_gn_step_heads: {len(ms)} calls, mean {sum(ms) / len(ms) * 1e3:.1f} us, min {min(ms) * 1e3:.1f}, max {max(ms) * 1e3:.1f}")
    if tag == "stats":
        out = (C.c_ulonglong * 2)()
        lib.codd_gn_stats(out, 0)
        print(f"[stats] wave-level neighbour visits {out[0]}, skipped by the a < 1e-9 test {out[1]} = {out[1] / max(1, out[0]):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for tag, flags in (("skip", ""), ("stats", "-DGN_STATS"), ("noskip", "-DGN_NO_SKIP")):
        env = dict(os.environ, CODD_EXTRA_FLAGS=flags)
        os.utime(os.path.join(ROOT, "codd_amd", "csrc", "motion.hip"))
        subprocess.check_call([sys.executable, "-c", "from codd_amd import build; build.build(verbose=False)"], cwd=ROOT, env=env)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag], cwd=ROOT, env=env)
    os.utime(os.path.join(ROOT, "codd_amd", "csrc", "motion.hip"))
    subprocess.check_call([sys.executable, "-c", "from codd_amd import build; build.build(verbose=False)"], cwd=ROOT)
