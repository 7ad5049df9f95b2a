"""GPU parity of the whole HITNetMF forward against the CPU oracle on identical inputs/weights.

north_star tolerance: mean |disp_gpu - disp_oracle| (EPE delta) <= 1e-3 px."""
import pytest
import torch

pytestmark = pytest.mark.gpu

STEREO_CFG = dict(type="HITNetMF", backbone=dict(type="HITUNet"),
                  initialization=dict(type="TileInitialization", max_disp=320),
                  propagation=dict(type="TilePropagation"))


@pytest.mark.parametrize("H,W", [(128, 256), (192, 320)])
def test_hitnet_matches_oracle(H, W):
    import codd_amd  # noqa: F401
    from codd_amd import synth
    from codd_amd.registry import build_estimator
    from oracle import stereo as ost
    est = build_estimator(dict(type="ConsistentOnlineDynamicDepth", stereo=STEREO_CFG)).eval()
    synth.load_synthetic_weights(est, gain=1.4)
    sd = {k: v.clone() for k, v in est.state_dict().items()}
    img, r_img, _ = synth.stereo_sequence(H, W, 1)
    with torch.no_grad():
        ref = ost.stereo_matching(sd, img[:, 0], r_img[:, 0], return_intermediates=True)
    est = est.to("cuda:0")
    out = est.stereo.stereo_matching(img[:, 0].to("cuda:0"), r_img[:, 0].to("cuda:0"))
    for k in ("left_feat", "right_feat"):
        err = (out[k].cpu() - ref[k]).abs().max().item()
        assert err < 1e-4 * max(1.0, ref[k].abs().max().item()), (k, err)
    d, dr = out["pred_disp"].cpu(), ref["pred_disp"]
    assert d.shape == dr.shape == (1, 1, H, W)
    epe = (d - dr).abs().mean().item()
    flipped = ((d - dr).abs() > 0.5).float().mean().item()
    print(f"EPE delta {epe:.3e}, >0.5px {flipped:.3e}, max {(d - dr).abs().max().item():.3e}")
    assert epe < 1e-3, epe


def test_cfg1_stereo_only_sequence_matches_reference_output():
    """BASELINE.json configs[0] on the GPU: 2-frame 512x256 stereo-only sequence, HIP path vs the
    REFERENCE's own output (tests/golden cfg1_pred_disp, every 2nd pixel): EPE <= 1e-3 px."""
    import os
    import sys
    import numpy as np
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    import cases
    from codd_amd import configs
    from codd_amd.registry import build_estimator
    G = np.load(os.path.join(gdir, "reference_outputs.npz"))
    est = build_estimator(configs.stereo_only())
    sd = cases.state_dict()
    est.load_state_dict({k: v for k, v in sd.items() if k.startswith("stereo.")}, strict=True)
    est = est.to("cuda").eval()
    img, r_img = cases.cfg1_sequence()
    meta = [[dict(img_shape=(256, 512, 3), disp_range=(1, 210), intrinsics=[1050.0, 1050.0, 256.0, 128.0])]]
    pred = est(img=[img.cuda()], r_img=[r_img.cuda()], img_metas=meta, return_loss=False, evaluate=False)[0]
    ref = torch.from_numpy(G["cfg1_pred_disp"])
    epe = (pred.cpu()[:, :, ::2, ::2] - ref).abs().mean().item()
    assert epe < 1e-3, epe
    assert (ref > 0).float().mean() > 0.9


def test_stereo_only_graph_replay_equals_eager():
    from codd_amd import configs, synth
    from codd_amd.registry import build_estimator
    from codd_amd.runtime import FrameRunner
    est = build_estimator(configs.stereo_only()).to("cuda").eval()
    synth.load_synthetic_weights(est, gain=1.4)
    img, r_img, _ = synth.stereo_sequence(128, 256, 3, 24.0)
    img, r_img = img.cuda(), r_img.cuda()
    metas = synth.default_metas(128, 256)[0]
    eager, graph = FrameRunner(est, metas, use_graph=False), FrameRunner(est, metas, use_graph=True)
    for f in range(3):
        a = eager.step(img[:, f].contiguous(), r_img[:, f].contiguous()).clone()
        b = graph.step(img[:, f].contiguous(), r_img[:, f].contiguous()).clone()
        assert torch.equal(a, b), f
    assert graph.graph is not None

