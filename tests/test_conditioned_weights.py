"""The conditioned synthetic weight set (codd_amd.synth, mode="conditioned") does what DESIGN.md section 2 says it does -- measured
on the CPU oracle at a small size, so that the claim is checked wherever the tests run (no GPU, no reference needed):
HITNet is a stereo matcher on the conditioned video (sub-pixel median error against the video's ground truth, no garbage
disparities), its output does not move under 1e-6 relative input noise, and the random filler it replaces is NOT one."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codd_amd import configs, synth  # noqa: E402
from codd_amd.registry import build_estimator  # noqa: E402
from oracle import stereo as ostereo  # noqa: E402

H, W, TAPER = 256, 384, 96.0


def _ground_truth(t):
    """left-referenced ground truth D(x) = d(x - D(x), y) of synth.stereo_sequence (d is defined on the right image's grid)"""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")

    def d(xx):
        s = 0.5 + 0.25 * torch.sin(2 * math.pi * (xx / W) + 0.1 * t) + 0.25 * torch.cos(2 * math.pi * (y / H) * 1.5)
        return (1.0 + 47.0 * s.clamp(0, 1)) * (xx / TAPER).clamp(0, 1)
    D = d(x)
    for _ in range(40):
        D = d(x - D)
    return D.float()[None, None]


def _stereo(mode):
    est = build_estimator(configs.stereo_only()).eval()
    synth.load_synthetic_weights(est, gain=1.4, mode=mode)
    return {k: v.clone() for k, v in est.state_dict().items()}


def test_conditioned_hitnet_is_a_stereo_matcher_and_the_random_filler_is_not():
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 8)))
    img, r_img, _ = synth.stereo_sequence(H, W, 2, flow=(0.737, 0.263), texture="waves", left_taper=TAPER)
    res = {}
    with torch.no_grad():
        for mode in ("conditioned", "random"):
            sd = _stereo(mode)
            for f in range(2):
                d = ostereo.stereo_matching(sd, img[:, f], r_img[:, f], 320)["pred_disp"]
                e = (d - _ground_truth(float(f))).abs()
                res[(mode, f)] = (e.median().item(), (e > 3).float().mean().item(), d.max().item())
                if mode == "conditioned":
                    g = torch.Generator().manual_seed(f)
                    l = img[:, f] * (1 + 1e-6 * torch.randn(img[:, f].shape, generator=g))
                    r = r_img[:, f] * (1 + 1e-6 * torch.randn(img[:, f].shape, generator=g))
                    dd = (ostereo.stereo_matching(sd, l, r, 320)["pred_disp"] - d).abs()
                    assert dd.mean().item() < 1e-4 and (dd > 0.25).float().mean().item() < 1e-4, (f, dd.mean().item())
    for f in range(2):
        med, bad, dmax = res[("conditioned", f)]
        print(f"conditioned frame {f}: median error {med:.3f} px, {bad:.4f} of the pixels off by > 3 px, max disparity {dmax:.1f}")
        assert med < 0.6 and bad < 5e-3 and dmax < 64.0, res[("conditioned", f)]
        med, bad, dmax = res[("random", f)]
        print(f"random      frame {f}: median error {med:.3f} px, {bad:.4f} of the pixels off by > 3 px, max disparity {dmax:.1f}")
        assert med > 2.0 and bad > 0.3, res[("random", f)]  # (what the goldens up to round 5 were made with)


def test_conditioned_fill_is_deterministic_and_leaves_the_random_mode_alone():
    a = synth.fill_state_dict([("stereo.tile_update.tile_update2.conv0.0.weight", (32, 64, 1, 1)), ("fusion.weight_head.1.bias", (1,))],
                              1.4, "conditioned")
    b = synth.fill_state_dict([("stereo.tile_update.tile_update2.conv0.0.weight", (32, 64, 1, 1)), ("fusion.weight_head.1.bias", (1,))],
                              1.4, "conditioned")
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["stereo.tile_update.tile_update2.conv0.0.weight"][:, :, 0, 0]
    assert w[0, 16] == 1.0 and w[0, 48] == -1.0 and w[1, 16] == -1.0 and w[2, 0] == 1.0 and w[2, 32] == -1.0  # cost / consistency rows
    assert a["fusion.weight_head.1.bias"].item() == 1.5
    r = synth.fill_state_dict([("stereo.backbone.down1.0.weight", (16, 16, 4, 4))], 1.4)
    assert torch.equal(r["stereo.backbone.down1.0.weight"], synth.fill_tensor("stereo.backbone.down1.0.weight", (16, 16, 4, 4), 1.4))
