import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_headline_parity as T
from codd_amd import ops, synth
from codd_amd.runtime import FrameRunner
name = "cfg5_tartanair_640x512"
H, W, intr, img_shape, stereo_only, MF = T.CASES[name]
est, sd = T._build(stereo_only)
ref = T.oracle_frames(name, sd, allow_compute=True)[0]
est = est.to("cuda:0")
img, r_img, _ = synth.stereo_sequence(H, W, MF)
metas = synth.default_metas(H, W, img_shape=img_shape, intrinsics=intr)
ops.enable_autotune(True, shipped=True)
def fp32(fn):
    def w(*a, **k):
        with ops.stage("stereo"):  # the stage name whose policy is exact fp32
            return fn(*a, **k)
    return w
r3 = est.motion.raft3d
for what in sys.argv[1:]:
    if what in ("fp32", "split", "bf16"): ops.set_conv_precision(what)
    elif what == "fnet": r3.fnet.forward = fp32(r3.fnet.forward)
    elif what == "cnet": r3.context = fp32(r3.context)
    elif what == "update": r3.update_block.run = fp32(r3.update_block.run)
    elif what == "fusion": est.fusion.memory_query = fp32(est.fusion.memory_query)
runner = FrameRunner(est, metas[0], use_graph=False)
for f in range(MF):
    d = runner.step(img[:, f].to("cuda:0").contiguous(), r_img[:, f].to("cuda:0").contiguous()).cpu()
    diff = (d - ref[f]).abs()[0, 0]
    ys, xs = torch.nonzero(diff > 0.25, as_tuple=True)
    print("frame", f, "flipped", len(ys))
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:12]:
        print("  (%d,%d) gpu %.3f ref %.3f" % (y, x, d[0, 0, y, x].item(), ref[f][0, 0, y, x].item()))
    if len(ys):
        print("  rows", ys.min().item(), ys.max().item(), "cols", xs.min().item(), xs.max().item())
