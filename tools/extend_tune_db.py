"""dev (GPU box): add the launch configurations of layer signatures the shipped tune db does not know yet (new stage
policies, new shapes) -- existing entries are kept as they are.  Writes gpurun_out/mi355x_extended.json; copy it over
codd_amd/tuned/mi355x.json to ship it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner

ops.enable_autotune(True, shipped=True)
before = dict(ops.TUNE_DB)
# B = 2: bench.py's fps_two_videos_batched pass and tests/test_gpu_headline_parity.py::test_two_videos_in_lock_step; fp32: bench.py's
# fp32_exact_fps pass -- with their signatures shipped a default bench.py run times nothing on the fly
RUNS = [(name, prec, 1) for name, c in T.CASES.items() for prec in (("split", "bf16mix", "fp16mix", "fp32") if not c[4] else ("split",))]
RUNS += [("cfg3_codd_960x576", "split", 2), ("cfg5_tartanair_640x512", "split", 2)]
for name, prec, Bn in RUNS:
    H, W, intr, img_shape, stereo_only, MF = T.CASES[name]
    if True:
        prev = ops.set_conv_precision(prec)
        try:
            est = T._build(stereo_only)[0].to("cuda:0")
            img, r_img, _ = synth.stereo_sequence(H, W, 3)
            metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
            runner = FrameRunner(est, metas[0], use_graph=False)
            for f in range(3):
                runner.step(img[:, f].repeat(Bn, 1, 1, 1).to("cuda:0").contiguous(), r_img[:, f].repeat(Bn, 1, 1, 1).to("cuda:0").contiguous())
            torch.cuda.synchronize()
        finally:
            ops.set_conv_precision(prev)
        print(name, prec, "B =", Bn, "db entries:", len(ops.TUNE_DB), flush=True)
new = {k: v for k, v in ops.TUNE_DB.items() if k not in before}
print("new signatures:", len(new))
for k, v in sorted(new.items()):
    print("  ", k, v)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mi355x_extended.json")
json.dump({k: list(v) for k, v in sorted(ops.TUNE_DB.items())}, open(out, "w"), indent=0)
print("wrote", out)
