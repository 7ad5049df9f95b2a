"""dev: time the CPU oracle stages on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
nt = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
torch.set_num_threads(nt)
print("cpu_count", os.cpu_count(), "threads", torch.get_num_threads(), flush=True)
import codd_amd
from codd_amd import configs, synth
from codd_amd.registry import build_estimator
from oracle import stereo as ost, motion as om, fusion as ofu
h, w = 192, 320
est = build_estimator(configs.codd(iters=16)).eval()
synth.load_synthetic_weights(est, 1.4)
sd = est.state_dict()
img, r_img, _ = synth.stereo_sequence(h, w, 2)
intr = (350.0, 350.0, w / 2.0, h / 2.0)
state = {}
with torch.no_grad():
    t = time.time(); out = ost.stereo_matching(sd, img[:, 0], r_img[:, 0]); print("stereo", time.time() - t, flush=True)
    t = time.time(); om.motion_forward(sd, state, out, intr, 16); print("motion first", time.time() - t, flush=True)
    ofu.memory_query(sd, out, state); ofu.memory_update(out, state)
    out = ost.stereo_matching(sd, img[:, 1], r_img[:, 1])
    t = time.time(); om.motion_forward(sd, state, out, intr, 2); print("motion 2 iters", time.time() - t, flush=True)
    t = time.time(); ofu.memory_query(sd, out, state); print("fusion", time.time() - t, flush=True)
