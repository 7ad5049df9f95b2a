"""Invariant tests of the oracle ops whose reference counterparts are absent third-party CUDA code
(lietorch SE3, corr_index, se3_build + cholesky, pytorch3d splat, mmseg HRNet): PARITY UNPINNED by
the reference, so the specification in oracle/ is validated physically."""
import math

import torch

from oracle import hrnet as oh, motion as om, se3


def R(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g)


def test_se3_log_exp_compose():
    xi = R(4, 5, 6) * 0.7
    T = se3.exp(xi)
    assert (se3.log(T) - xi).abs().max() < 1e-5
    small = R(4, 5, 6, seed=1) * 1e-5
    assert (se3.log(se3.exp(small)) - small).abs().max() < 1e-9
    assert (T[..., 3:].norm(dim=-1) - 1).abs().max() < 1e-6
    T2, P = se3.exp(R(4, 5, 6, seed=2) * 0.3), R(4, 5, 3, seed=3)
    assert (se3.act(se3.compose(T, T2), P) - se3.act(T, se3.act(T2, P))).abs().max() < 1e-5
    I = se3.identity(4, 5)
    assert (se3.act(I, P) - P).abs().max() == 0
    # pure rotation about z by 90 degrees
    T90 = se3.exp(torch.tensor([0, 0, 0, 0, 0, math.pi / 2]))
    assert (se3.act(T90, torch.tensor([1.0, 0, 0])) - torch.tensor([0, 1.0, 0])).abs().max() < 1e-6


def test_lookup_of_delta_volume_returns_bilinear_weights():
    vol = torch.zeros(1, 2, 2, 6, 7)
    vol[0, :, :, 3, 4] = 1.0  # delta at (y=3, x=4)
    co = torch.tensor([3.25, 2.5]).view(1, 2, 1, 1).expand(1, 2, 2, 2).contiguous()
    lk = om.corr_lookup_level(vol, co, 3).view(1, 7, 7, 2, 2)[0, :, :, 0, 0]  # [i (x off), j (y off)]
    exp = torch.zeros(7, 7)
    exp[3, 3], exp[3, 4], exp[4, 3], exp[4, 4] = 0.25 * 0.5, 0.25 * 0.5, 0.75 * 0.5, 0.75 * 0.5
    assert (lk - exp).abs().max() < 1e-6


def test_identity_motion_zero_flow_and_splat_is_half_pixel_blur():
    B, H, W = 1, 16, 24
    K = torch.tensor([[30.0, 30.0, 12.0, 8.0]])
    depth = torch.full((B, H, W), 10.0)
    T = se3.identity(B, H, W)
    assert om.induced_flow2d(T, depth, K).abs().max() < 1e-6
    feat = torch.ones(B, 2, H, W)
    out, z = om.splat(T, depth, feat, K, 2.0)
    # a point at integer (x, y) lands on the corner shared by pixels {x-1, x} x {y-1, y} (pixel centres
    # at +0.5): every pixel but the last row / column is covered by 4 points at squared distance 0.5
    # -> alpha 0.5 each -> composite 1 - 0.5^4; nearest depth = 10
    assert (out[:, :, :-1, :-1] - (1 - 0.5 ** 4)).abs().max() < 1e-4
    assert (z[:, :, :-1, :-1] - 10.0).abs().max() < 1e-5
    assert abs(out[0, 0, -1, -1].item() - 0.5) < 1e-4  # bottom-right corner: one point


def test_gn_step_recovers_rigid_motion():
    B, h, w = 1, 10, 14
    K = torch.tensor([[20.0, 20.0, 7.0, 5.0]])
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(B, h, w, generator=g) * 5 + 3
    xi = torch.tensor([0.05, -0.03, 0.02, 0.01, -0.02, 0.015])
    T_true = se3.exp(xi).expand(B, h, w, 7).contiguous()
    X0 = om.inv_project(depth, K)
    p = se3.act(T_true, X0)
    target = torch.stack([K[0, 0] * p[..., 0] / p[..., 2] + K[0, 2], K[0, 1] * p[..., 1] / p[..., 2] + K[0, 3],
                          1.0 / p[..., 2]], -1).permute(0, 3, 1, 2).contiguous()
    T = se3.identity(B, h, w)
    ae = torch.zeros(B, 32, h, w)
    weight = torch.ones(B, 3, h, w) * torch.tensor([1.0, 1.0, 100.0]).view(1, 3, 1, 1)
    pts = X0.permute(0, 3, 1, 2).contiguous()
    for _ in range(12):
        H_, b_ = om.se3_build(T, ae, pts, target, weight, K)
        dx = om.gn_solve(H_, b_, lm=0.0, ep=1e-6)
        T = se3.compose(se3.exp(dx), T)
    assert (se3.log(T) - xi).abs().max() < 2e-3


def test_hrnet_shape_contract_and_bn_fold():
    spec = oh.state_dict_spec("c")
    from codd_amd import synth
    sd = synth.fill_state_dict(spec, gain=1.4)
    x = R(1, 3, 64, 128)
    ys = oh.hrnet(sd, "c.0", x)
    assert [tuple(y.shape[1:]) for y in ys] == [(18, 16, 32), (36, 8, 16), (72, 4, 8), (144, 2, 4)]
    out = oh.cnet(sd, "c", x)
    assert tuple(out.shape) == (1, 512, 8, 16) and out.min() >= 0
