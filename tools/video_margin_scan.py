"""dev (CPU): does a synthetic video keep the STEREO stage away from coarse-tile arg-min ties on every frame?

The stereo network is a per-frame function of (left, right): its tile cost-volume arg-mins are discontinuous selections,
and the synthetic texture (codd_amd/synth.py: a sum of six sinusoids) is nearly periodic along x, so at some frames two
disparity candidates of a coarse tile cost the same to the last bit and ANY two correct fp32 implementations may pick
different ones (round 4: frame 14 of the first video; round 5: frame 1 of the (0.737, 0.263) video, an 83 x 85-pixel
block, tools/frame_event_vs_oracle.py).  For every frame of a candidate video this runs the oracle's stereo stage on the
exact images and on K perturbed copies (relative noise NOISE, far ABOVE the 1e-7-level differences between two fp32
evaluation orders) and reports the fraction of pixels that move by more than 0.25 px: a frame whose selections survive
1e-5 noise has margin to spare for rounding differences.

    python tools/video_margin_scan.py 0.737,0.263 0.61,0.37 ...        # FRAMES=50 K=2 NOISE=1e-5
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_headline_parity as T
from codd_amd import synth
from oracle import stereo as ostereo

FRAMES = int(os.environ.get("FRAMES", "50"))
K = int(os.environ.get("K", "2"))
NOISE = float(os.environ.get("NOISE", "1e-5"))
H, W = T.CASES["cfg3_codd_960x576"][:2]
torch.set_num_threads(int(os.environ.get("THREADS", "8")))
sd = T._build(False, 16)[1]
kw0 = dict(texture=os.environ["TEXTURE"]) if os.environ.get("TEXTURE") else {}
for arg in sys.argv[1:]:  # fx,fy[,dphase]
    v = [float(v) for v in arg.split(",")]
    flow, kw = tuple(v[:2]), dict(kw0, **({"dphase": v[2]} if len(v) > 2 else {}))
    img, r_img, _ = synth.stereo_sequence(H, W, FRAMES, flow=flow, **kw)
    bad, t0 = [], time.time()
    with torch.no_grad():
        for f in range(FRAMES):
            base = ostereo.stereo_matching(sd, img[:, f], r_img[:, f], 320)["pred_disp"]
            worst = 0.0
            for k in range(K):
                g = torch.Generator().manual_seed(100 * f + k)
                l = img[:, f] * (1 + NOISE * torch.randn(img[:, f].shape, generator=g))
                r = r_img[:, f] * (1 + NOISE * torch.randn(img[:, f].shape, generator=g))
                d = (ostereo.stereo_matching(sd, l, r, 320)["pred_disp"] - base).abs()
                worst = max(worst, (d > 0.25).float().mean().item())
            if worst > 2e-5:
                bad.append((f, worst))
            print(f"flow {flow} frame {f:2d}: flipped fraction under {NOISE:g} noise {worst:.2e}" + ("   <-- near-tie" if worst > 2e-5 else ""), flush=True)
    print(f"== flow {flow}{' ' + str(kw) if kw else ''}: {len(bad)} of {FRAMES} frames with a near-tie {bad}  [{time.time() - t0:.0f} s]", flush=True)
