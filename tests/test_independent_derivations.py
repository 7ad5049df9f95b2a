"""Second, independent derivations of the ops whose specification the reference does not pin
(lietorch SE3 algebra, lietorch_extras' Gauss-Newton builder, pytorch3d's point rasteriser: un-vendored CUDA
dependencies).  The oracle (oracle/se3.py, oracle/motion.py) states them in closed form; here
each is re-derived by a DIFFERENT route -- matrix exponentials of 4x4 twists, fp64 automatic differentiation of the
projection, a brute-force O(pixels x points) rasteriser -- so that a sign, ordering or index slip in the closed forms
cannot hide.  CPU only; the HIP kernels are then compared with the oracle in tests/test_gpu_motion_ops.py."""
import math

import torch

from oracle import motion as om
from oracle import se3


def _hat6(xi):
    """[tau, phi] -> 4x4 twist matrix."""
    tau, phi = xi[:3], xi[3:]
    M = torch.zeros(4, 4, dtype=xi.dtype)
    M[0, 1], M[0, 2], M[1, 2] = -phi[2], phi[1], -phi[0]
    M[1, 0], M[2, 0], M[2, 1] = phi[2], -phi[1], phi[0]
    M[:3, 3] = tau
    return M


def _mat(T):
    """[t, q_xyzw] -> 4x4 homogeneous matrix through the textbook quaternion -> rotation formula."""
    x, y, z, w = [float(v) for v in T[3:]]
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
    M = torch.eye(4, dtype=torch.float64)
    M[:3, :3] = R
    M[:3, 3] = T[:3].double()
    return M


def test_se3_exp_log_compose_act_against_matrix_exponential():
    g = torch.Generator().manual_seed(0)
    for scale in (1e-4, 0.05, 0.7, 2.5):
        for _ in range(8):
            xi = (torch.rand(6, generator=g, dtype=torch.float64) - 0.5) * 2 * scale
            xj = (torch.rand(6, generator=g, dtype=torch.float64) - 0.5) * 2 * scale
            Ti, Tj = se3.exp(xi.float()), se3.exp(xj.float())
            Mi, Mj = torch.linalg.matrix_exp(_hat6(xi)), torch.linalg.matrix_exp(_hat6(xj))
            assert (_mat(Ti) - Mi).abs().max() < 2e-6 * max(1.0, scale)           # exp
            assert (_mat(se3.compose(Ti, Tj)) - Mi @ Mj).abs().max() < 1e-5      # T1 * T2 = left multiplication
            X = torch.rand(3, generator=g, dtype=torch.float64) * 4 - 2
            Y = se3.act(Ti, X.float())
            assert (Y.double() - (Mi[:3, :3] @ X + Mi[:3, 3])).abs().max() < 1e-5  # T * X
            if float(xi[3:].norm()) < 3.0:  # log o exp = id inside the injectivity radius |phi| < pi
                assert (se3.log(Ti).double() - xi).abs().max() < 3e-6 * max(1.0, scale)
    assert (se3.exp(torch.zeros(6)) - se3.identity()).abs().max() == 0


def test_gauss_newton_builder_against_autograd_jacobians():
    """H_i = sum_j a_ij J^T W J, b_i = sum_j a_ij J^T W r with J = d project(exp(xi) T_i X_j) / d xi at 0 obtained by
    fp64 automatic differentiation through the MATRIX exponential (no closed-form Jacobian, no quaternion algebra),
    affinity = sigmoid(-|ae_i - ae_j|^2), window |dy|, |dx| <= radius, pairs with a depth below MIN_DEPTH skipped."""
    torch.manual_seed(1)
    h, w, C, radius = 5, 7, 6, 2
    N = h * w
    K = torch.tensor([[9.0, 8.5, 3.2, 2.4]])
    T = se3.exp(torch.randn(1, h, w, 6) * 0.08)
    depth = torch.rand(1, h, w) * 4 + 1.0
    depth[0, 1, 2] = 0.01  # below MIN_DEPTH: skipped as a neighbour
    ae = torch.randn(1, C, h, w) * 0.6
    pts = om.inv_project(depth, K).permute(0, 3, 1, 2).contiguous()
    target = om.project(se3.act(T, om.inv_project(depth, K)), K).permute(0, 3, 1, 2) + torch.randn(1, 3, h, w) * 0.1
    weight = torch.rand(1, 3, h, w)
    H, b = om.se3_build(T, ae, pts, target, weight, K, radius=radius)

    fx, fy, cx, cy = [float(v) for v in K[0]]

    def proj_of_twist(xi, Mi, X):
        Y = (torch.linalg.matrix_exp(_hat6(xi)) @ Mi @ torch.cat([X, X.new_ones(1)]))[:3]
        return torch.stack([fx * Y[0] / Y[2] + cx, fy * Y[1] / Y[2] + cy, 1.0 / Y[2]])

    A = ae[0].reshape(C, N).t().double()
    Xs = pts[0].reshape(3, N).t().double()
    tg = target[0].reshape(3, N).t().double()
    wt = weight[0].reshape(3, N).t().double()
    for i in (0, 9, 17, 23, N - 1):
        yi, xi_ = divmod(i, w)
        Mi = _mat(T[0, yi, xi_])
        Hi = torch.zeros(6, 6, dtype=torch.float64)
        bi = torch.zeros(6, dtype=torch.float64)
        for j in range(N):
            yj, xj = divmod(j, w)
            if abs(yj - yi) > radius or abs(xj - xi_) > radius:
                continue
            Y = (Mi @ torch.cat([Xs[j], Xs.new_ones(1)]))[:3]
            if Xs[j, 2] < om.MIN_DEPTH or Y[2] < om.MIN_DEPTH:
                continue
            a = torch.sigmoid(-((A[i] - A[j]) ** 2).sum())
            J = torch.autograd.functional.jacobian(lambda x: proj_of_twist(x, Mi, Xs[j]), torch.zeros(6, dtype=torch.float64))
            r = tg[j] - proj_of_twist(torch.zeros(6, dtype=torch.float64), Mi, Xs[j])
            Wm = torch.diag(a * wt[j])
            Hi += J.t() @ Wm @ J
            bi += J.t() @ Wm @ r
        Ho, bo = H[0, :, :, yi, xi_].double(), b[0, :, 0, yi, xi_].double()
        assert (Ho - Hi).abs().max() < 2e-4 * max(1.0, Hi.abs().max()), (i, (Ho - Hi).abs().max())
        assert (bo - bi).abs().max() < 2e-4 * max(1.0, bi.abs().max()), (i, (bo - bi).abs().max())


def test_factored_normal_equations_of_the_pair_builder_equal_jt_w_j():
    """se3_gn_build3_kernel (csrc/motion.hip) never forms the 3x6 Jacobian: it accumulates
        S^ = a A^T W A^  (A^ = [fx 0 -fx xn; 0 fy -fy yn; 0 0 -d], Y = (xn, yn, 1) / d),   N = S^ Q,   Q = -[(xn, yn, 1)]x,
        H_tt = d^2 S^,  H_tr = d N,  H_rr = Q^T N,   g^ = a A^T W r,  b_t = d g^,  b_r = Q^T g^
    with the pixel residuals taken in units of fx, fy (normalised targets, weights x f^2: the record image ``geo2``).  Here
    the same formulas in fp64, entry by entry as the kernel writes them, against J^T W J / J^T W r with J obtained by
    automatic differentiation through the matrix exponential."""
    torch.manual_seed(3)
    fx, fy, cx, cy = 9.0, 8.5, 3.2, 2.4
    for trial in range(6):
        Mi = _mat(se3.exp(torch.randn(6) * 0.2))
        X = torch.tensor([0.4, -0.3, 2.5], dtype=torch.float64) + torch.randn(3, dtype=torch.float64) * 0.3
        tgt = torch.randn(3, dtype=torch.float64) * torch.tensor([3.0, 3.0, 0.2], dtype=torch.float64)
        w = torch.rand(3, dtype=torch.float64)
        a = float(torch.rand(1)) * 0.9 + 0.05

        def proj_of_twist(xi):
            Y = (torch.linalg.matrix_exp(_hat6(xi)) @ Mi @ torch.cat([X, X.new_ones(1)]))[:3]
            return torch.stack([fx * Y[0] / Y[2] + cx, fy * Y[1] / Y[2] + cy, 1.0 / Y[2]])

        z6 = torch.zeros(6, dtype=torch.float64)
        J = torch.autograd.functional.jacobian(proj_of_twist, z6)
        r = tgt - proj_of_twist(z6)
        Wm = torch.diag(a * w)
        H_ref, b_ref = J.t() @ Wm @ J, J.t() @ Wm @ r

        # the kernel's formulas (names as in se3_gn_build3_kernel)
        Y = (Mi @ torch.cat([X, X.new_ones(1)]))[:3]
        d = 1.0 / Y[2]
        xn, yn = Y[0] * d, Y[1] * d
        tnx, tny, tz = (tgt[0] - cx) / fx, (tgt[1] - cy) / fy, tgt[2]  # geo2: normalised target
        Wx, Wy, wz = w[0] * fx * fx, w[1] * fy * fy, w[2]               # geo2: weights x f^2
        rx, ry, rz = tnx - xn, tny - yn, tz - d
        S00, S11, t = a * Wx, a * Wy, a * wz * d
        S02, S12 = -(S00 * xn), -(S11 * yn)
        S22 = t * d - (S12 * yn + S02 * xn)
        g0, g1 = S00 * rx, S11 * ry
        g2 = -(t * rz + g1 * yn + g0 * xn)
        dd = d * d
        N00, N01, N02 = yn * S02, -xn * S02 + S00, -(yn * S00)
        N10, N11, N12 = yn * S12 - S11, -(xn * S12), xn * S11
        N20, N21, N22 = yn * S22 - S12, -xn * S22 + S02, xn * S12 - N00
        H = torch.zeros(6, 6, dtype=torch.float64)
        H[0, 0], H[1, 1], H[0, 2], H[1, 2], H[2, 2] = dd * S00, dd * S11, dd * S02, dd * S12, dd * S22
        H[0, 3], H[0, 4], H[0, 5] = d * N00, d * N01, d * N02
        H[1, 3], H[1, 4], H[1, 5] = d * N10, d * N11, d * N12
        H[2, 3], H[2, 4], H[2, 5] = d * N20, d * N21, d * N22
        H[3, 3], H[3, 4], H[3, 5] = yn * N20 - N10, yn * N21 - N11, yn * N22 - N12
        H[4, 4], H[4, 5] = -xn * N21 + N01, -xn * N22 + N02
        H[5, 5] = xn * N12 - yn * N02
        H = torch.triu(H) + torch.triu(H, 1).t()
        b = torch.stack([d * g0, d * g1, d * g2, yn * g2 - g1, -xn * g2 + g0, xn * g1 - yn * g0])
        assert (H - H_ref).abs().max() < 1e-10 * max(1.0, H_ref.abs().max()), (trial, (H - H_ref).abs().max())
        assert (b - b_ref).abs().max() < 1e-10 * max(1.0, b_ref.abs().max()), (trial, (b - b_ref).abs().max())
        assert abs(H_ref[0, 1]) < 1e-12  # (the structural zero the builders do not accumulate)


def test_gauss_newton_solve_is_the_damped_normal_equation():
    torch.manual_seed(2)
    J = torch.randn(1, 4, 5, 12, 6)
    H = (J.transpose(-1, -2) @ J).permute(0, 3, 4, 1, 2).contiguous()  # [B,6,6,h,w] SPD
    b = torch.randn(1, 6, 1, 4, 5)
    dx = om.gn_solve(H, b, lm=1e-4, ep=10.0)
    for y in range(4):
        for x in range(5):
            A = H[0, :, :, y, x].double().clone()
            A += torch.diag(torch.diagonal(A) * 1e-4 + 10.0)
            ref = torch.linalg.solve(A, b[0, :, 0, y, x].double())
            assert (dx[0, y, x].double() - ref).abs().max() < 1e-6


def _brute_force_splat(P, feat, H, W, fx, fy, cx, cy, R, K=8):
    """O(pixels x points) rasteriser + compositor, written from the pytorch3d semantics the oracle states: a point
    covers the pixels whose centre (x + 0.5, y + 0.5) lies strictly within R of its projection; per pixel the K
    nearest points in z (ties: lower index) are composited front to back with alpha = 1 - d^2 / R^2."""
    C = feat.shape[0]
    out = torch.zeros(C, H, W, dtype=torch.float64)
    zb = torch.zeros(H, W, dtype=torch.float64)
    proj = []
    for n in range(P.shape[0]):
        X, Y, Z = [float(v) for v in P[n]]
        if not Z > 0:
            continue
        proj.append((n, fx * X / Z + cx, fy * Y / Z + cy, Z))
    for py in range(H):
        for px in range(W):
            cand = []
            for (n, u, v, z) in proj:
                d2 = (u - (px + 0.5)) ** 2 + (v - (py + 0.5)) ** 2
                if d2 < R * R:
                    cand.append((z, n, 1.0 - d2 / (R * R)))
            cand.sort()
            tr = 1.0
            for k, (z, n, a) in enumerate(cand[:K]):
                out[:, py, px] += tr * a * feat[:, n].double()
                tr *= 1.0 - a
                if k == 0:
                    zb[py, px] = z
    return out, zb


def test_splat_against_brute_force_rasteriser_with_pileups_and_ties():
    torch.manual_seed(3)
    H, W, C = 10, 14, 3
    K = torch.tensor([[11.0, 10.5, 7.0, 5.0]])
    depth = torch.rand(1, H, W) * 6 + 2
    depth[0, 0, :4] = 0.0  # culled (z <= 0 after the identity part of the motion)
    T = se3.exp(torch.randn(1, H, W, 6) * 0.03)
    # pile-up: the last 60 points are sent onto (nearly) one pixel, half of them with EXACTLY equal depth
    X0 = om.inv_project(depth, K)
    n_pile = 60
    idx = torch.arange(H * W - n_pile, H * W)
    tz = torch.where(idx % 2 == 0, torch.full((n_pile,), 3.0), 3.0 + 0.01 * (idx % 7).float())
    tu, tv = 6.5 + 0.3 * torch.sin(idx.float()), 4.5 + 0.3 * torch.cos(idx.float())
    P_t = torch.stack([(tu - 7.0) * tz / 11.0, (tv - 5.0) * tz / 10.5, tz], -1)
    Tf = T.reshape(-1, 7).clone()
    Tf[idx, :3] = P_t - X0.reshape(-1, 3)[idx]
    Tf[idx, 3:] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    T = Tf.reshape(1, H, W, 7)
    feat = torch.randn(1, C, H, W)
    for radius in (2.0, 4.0):
        got, zg = om.splat(T, depth, feat, K, radius)
        R = radius * min(H, W) / (2.0 * H)
        P = se3.act(T[0], X0[0]).reshape(-1, 3)
        ref, zr = _brute_force_splat(P, feat[0].reshape(C, -1), H, W, 11.0, 10.5, 7.0, 5.0, R)
        assert (got[0].double() - ref).abs().max() < 1e-5, (radius, (got[0].double() - ref).abs().max())
        assert (zg[0, 0].double() - zr).abs().max() < 1e-6
        # the pile-up pixel really holds more than 8 (and more than the old fixed capacities) candidates
        u = 11.0 * P[:, 0] / P[:, 2].clamp(min=1e-9) + 7.0
        v = 10.5 * P[:, 1] / P[:, 2].clamp(min=1e-9) + 5.0
        assert int((((u - 6.5) ** 2 + (v - 4.5) ** 2) < R * R).sum()) > 48


def test_identity_motion_splat_and_flow():
    """T = identity: zero induced flow; the splat of a pixel-centred grid puts point (x, y) at distance sqrt(0.5) of
    the four pixel centres around its corner -- the half-pixel shift of pytorch3d's NDC convention."""
    H, W = 6, 8
    K = torch.tensor([[5.0, 5.0, 4.0, 3.0]])
    depth = torch.full((1, H, W), 2.0)
    T = se3.identity(1, H, W)
    assert om.induced_flow2d(T, depth, K).abs().max() < 1e-6
    feat = torch.arange(H * W, dtype=torch.float32).reshape(1, 1, H, W)
    out, z = om.splat(T, depth, feat, K, 2.0)  # R = 1 px
    a = 1.0 - 0.5  # alpha of a point half a pixel away in x and y
    # interior pixel (y, x) is covered by points (y, x), (y, x+1), (y+1, x), (y+1, x+1), all at z = 2: index order
    y, x = 2, 3
    ids = [y * W + x, y * W + x + 1, (y + 1) * W + x, (y + 1) * W + x + 1]
    exp = sum(a * (1 - a) ** k * ids[k] for k in range(4))
    assert math.isclose(float(out[0, 0, y, x]), exp, rel_tol=1e-5) and float(z[0, 0, y, x]) == 2.0
