"""dev (CPU only): how well-conditioned is a synthetic weight set + video, measured on the ORACLE alone?

  stereo   per frame: disparity error of the oracle's HITNet against the video's ground truth (the tile initialisation
           per level and the final map) and how far the output moves under 1e-6 / 1e-5 relative input noise
  frames   the oracle's full recurrence (stereo -> motion -> fusion) evaluated twice -- oneDNN convolutions and ATen's
           im2col + sgemm path (torch.backends.mkldnn.flags(enabled=False): another summation order of the same fp32 sums)
           -- per frame: mean |delta| between the two, flipped fraction, range of the two Fusion heads' logits

    python tools/cond_probe.py stereo|frames <case of CASES> <frames> [random|conditioned]
    env: TEXTURE=waves|sines  TAPER=<columns>  FLOW=fx,fy  THREADS=8
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import test_gpu_headline_parity as T
from codd_amd import configs, synth
from codd_amd.registry import build_estimator
from oracle import codd as oc
from oracle import fusion as ofusion
from oracle import stereo as ostereo


def exact_gt(H, W, t, dmax=48.0, taper=0.0):
    """left-referenced ground truth D(x) = d(x - D(x), y): synth.stereo_sequence defines d on the RIGHT image's grid"""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")

    def d(xx):
        s = 0.5 + 0.25 * torch.sin(2 * math.pi * (xx / W) + 0.1 * t) + 0.25 * torch.cos(2 * math.pi * (y / H) * 1.5)
        v = 1.0 + (dmax - 1.0) * s.clamp(0, 1)
        return v * (xx / taper).clamp(0, 1) if taper > 0 else v
    D = d(x)
    for _ in range(40):
        D = d(x - D)
    return D.float()[None, None]


def main():
    what, case, MF = sys.argv[1], sys.argv[2], int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "conditioned"
    H, W, intr, _, _, _ = T.CASES[case]
    torch.set_num_threads(int(os.environ.get("THREADS", "8")))
    tex, taper = os.environ.get("TEXTURE", "waves"), float(os.environ.get("TAPER", "96"))
    flow = tuple(float(v) for v in os.environ.get("FLOW", "0.737,0.263").split(","))
    import json
    synth._COND.update(json.loads(os.environ.get("COND", "{}")))  # dev override of the conditioned set's constants
    est = build_estimator(configs.codd(iters=16)).eval()
    synth.load_synthetic_weights(est, gain=1.4, mode=mode)
    sd = {k: v.clone() for k, v in est.state_dict().items()}
    img, r_img, _ = synth.stereo_sequence(H, W, MF, flow=flow, texture=tex, left_taper=taper)
    print(f"# {what} {case} {MF} frames, weights {mode}, texture {tex}, left taper {taper:g}, flow {flow}", flush=True)
    if what == "stereo":
        with torch.no_grad():
            for f in range(MF):
                G = exact_gt(H, W, float(f), taper=taper)
                o = ostereo.stereo_matching(sd, img[:, f], r_img[:, f], 320, return_intermediates=True)
                d = o["pred_disp"]
                e = (d - G).abs()
                lv = " ".join(f"{((h[:, :1] - F.avg_pool2d(G, 64 >> l) / (16 >> l)).abs() > 1).float().mean():.3f}" for l, h in enumerate(o["init"]))
                line = f"frame {f:2d}: error median {e.median():.3f} mean {e.mean():.3f} >3px {(e > 3).float().mean():.4f} max disparity {d.max():.1f}; init tiles off by > 1 step per level: {lv};"
                for noise in (1e-6, 1e-5):
                    g = torch.Generator().manual_seed(7 + f)
                    l = img[:, f] * (1 + noise * torch.randn(img[:, f].shape, generator=g))
                    r = r_img[:, f] * (1 + noise * torch.randn(img[:, f].shape, generator=g))
                    dd = (ostereo.stereo_matching(sd, l, r, 320)["pred_disp"] - d).abs()
                    line += f"  noise {noise:g}: mean |delta| {dd.mean():.2e} flipped {(dd > 0.25).float().mean():.2e}"
                print(line, flush=True)
        return
    logits = {}
    orig = torch.sigmoid

    def spy(x):
        logits.setdefault(tuple(x.shape[-2:]), []).append((x.min().item(), x.max().item(), x.mean().item(), x.abs().mean().item(), (x.abs() > 4).float().mean().item()))
        return orig(x)
    runs = {}
    for variant in ("default", "nomkldnn"):
        st, outs = {}, []
        ctx = torch.backends.mkldnn.flags(enabled=False) if variant == "nomkldnn" else None
        if ctx:
            ctx.__enter__()
        try:
            with torch.no_grad():
                for f in range(MF):
                    t0 = time.time()
                    ofusion.torch.sigmoid = spy if variant == "default" else orig
                    o = oc.frame(sd, img[:, f], r_img[:, f], st, intr, iters=16)
                    ofusion.torch.sigmoid = orig
                    outs.append(o["pred_disp"].clone())
                    msg = ""
                    if variant == "default" and f > 0:
                        G = exact_gt(H, W, float(f), taper=taper)
                        pw, pc = o["pred_warp"], o["pred_curr"]
                        v = pw > 0
                        msg = (f" err vs gt: fused {(outs[-1] - G).abs().mean():.3f} curr {(pc - G).abs().mean():.3f}; holes {1 - v.float().mean():.4f}; |warp - curr| mean {(pw - pc).abs()[v].mean():.3f} max {(pw - pc).abs()[v].max():.1f};"
                               f" wf mean {o['fusion_weights'].mean():.3f} wr mean {o['reset_weights'].mean():.3f}; logits (min, max, mean, mean |x|, frac |x| > 4) " + " ".join(f"{k}: {tuple(round(q, 3) for q in v_[-1])}" for k, v_ in logits.items()))
                    if variant == "nomkldnn":
                        dd = (outs[-1] - runs["default"][f]).abs()
                        msg = f" vs default: mean |delta| {dd.mean():.3e} sub4 {dd[..., ::4, ::4].mean():.3e} flipped {(dd > 0.25).float().mean():.2e} ({int((dd > 0.25).sum())} px) max {dd.max():.2f}"
                    print(f"[{variant}] frame {f:2d} mean disparity {outs[-1].mean():.3f} max {outs[-1].max():.1f}{msg}  [{time.time() - t0:.0f} s]", flush=True)
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
        runs[variant] = outs


if __name__ == "__main__":
    main()
