"""ORACLE (test infrastructure, not product): CPU fp32 restatement of Motion + RAFT3D.

reference model/motion/motion.py, model/motion/raft3d/*.py.

Pinned against the imported reference (tests/golden/make_golden.py): BasicEncoder,
BasicUpdateBlock + ConvGRU, CorrBlock.corr + pyramid, cvx_upsample, inv_project / project,
bilinear depth sampler.

PARITY UNPINNED (third-party CUDA ops absent from the reference tree, no reference test pins
them -- SURVEY.md section 8c): correlation lookup (lietorch_extras.corr_index_forward),
dense SE3 Gauss-Newton builder + 6x6 solve (se3_build_inplace, cholesky6x6_forward), SE3
algebra (oracle/se3.py), point splatting (pytorch3d PointsRasterizer + AlphaCompositor) and
HRNet (oracle/hrnet.py).  They are specified here from the call sites and the libraries'
published semantics and validated by invariants (tests/test_oracle_invariants.py) and by independent derivations
that share no code with this file (tests/test_independent_derivations.py).
"""
import torch
import torch.nn.functional as F

from . import se3
from .hrnet import cnet as hrnet_cnet
from .stereo import conv

MIN_DEPTH = 0.05  # reference projective_ops.py:7
EPS = 1e-5  # reference projective_ops.py:8
BF_DEFAULT = 1050 * 0.2  # reference motion.py:45


# ----------------------------------------------------------------------------- encoder
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def _res_block(sd, p, x, stride):
    """reference blocks/extractor.py:9-58 (norm_fn='instance')."""
    y = F.relu(_inorm(conv(sd, p + ".conv1", x, stride, 1)))
    y = F.relu(_inorm(conv(sd, p + ".conv2", y, 1, 1)))
    if stride != 1:
        x = _inorm(conv(sd, p + ".downsample.0", x, stride, 0))
    return F.relu(x + y)


def basic_encoder(sd, p, x):
    """reference blocks/extractor.py:119-199 (fnet: output_dim=128, instance norm)."""
    x = F.relu(_inorm(conv(sd, p + ".conv1", x, 2, 3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _res_block(sd, f"{p}.{name}.0", x, stride)
        x = _res_block(sd, f"{p}.{name}.1", x, 1)
    return conv(sd, p + ".conv2", x)


# ----------------------------------------------------------------------------- correlation
def all_pairs_corr(f1, f2):
    """reference blocks/corr.py:56-62: corr[b,y1,x1,y2,x2] = <f1, f2> / 16."""
    B, D, H, W = f1.shape
    c = torch.matmul((f1.reshape(B, D, H * W) / 4.0).transpose(1, 2), f2.reshape(B, D, H * W) / 4.0)
    return c.view(B, H, W, H, W)


def corr_pyramid(f1, f2, levels=4):
    """reference blocks/corr.py:28-45: level i = avg_pool2d(2)^i over (y2, x2)."""
    c = all_pairs_corr(f1, f2)
    B, H, W = c.shape[:3]
    c = c.reshape(B * H * W, 1, H, W)
    pyr = []
    for i in range(levels):
        pyr.append(c.view(B, H, W, c.shape[2], c.shape[3]))
        if i + 1 < levels:
            c = F.avg_pool2d(c, 2, stride=2)
    return pyr


def corr_lookup_level(vol, coords, r):
    """lietorch_extras.corr_index_forward (reference call site blocks/corr.py:17; UNPINNED).
    vol [B,h1,w1,h2,w2]; coords [B,2,h1,w1] (x,y); -> [B,(2r+1)^2,h1,w1] with channel
    i*(2r+1)+j, i = x-offset index, j = y-offset index; value = bilinear sample of the source
    pixel's own slice at (x0 - r + i, y0 - r + j), zero outside."""
    B, h1, w1, h2, w2 = vol.shape
    x0, y0 = coords[:, 0], coords[:, 1]
    fx, fy = torch.floor(x0), torch.floor(y0)
    dx, dy = x0 - fx, y0 - fy
    fx, fy = fx.long(), fy.long()
    flat = vol.reshape(B, h1, w1, h2 * w2)
    rd = 2 * r + 1

    def tap(ix, iy):
        ok = ((ix >= 0) & (ix < w2) & (iy >= 0) & (iy < h2)).to(vol.dtype)
        idx = (iy.clamp(0, h2 - 1) * w2 + ix.clamp(0, w2 - 1)).unsqueeze(-1)
        return torch.gather(flat, 3, idx).squeeze(-1) * ok

    out = []
    for i in range(rd):
        for j in range(rd):
            ix, iy = fx - r + i, fy - r + j
            v = ((1 - dx) * (1 - dy)) * tap(ix, iy) + (dx * (1 - dy)) * tap(ix + 1, iy) \
                + ((1 - dx) * dy) * tap(ix, iy + 1) + (dx * dy) * tap(ix + 1, iy + 1)
            out.append(v)
    return torch.stack(out, 1)


def corr_lookup(pyr, coords, r=3):
    """reference blocks/corr.py:47-54."""
    return torch.cat([corr_lookup_level(v, coords / 2 ** i, r) for i, v in enumerate(pyr)], 1)


# ----------------------------------------------------------------------------- projective ops
def inv_project(depth, K):
    """reference projective_ops.py:25-41.  depth [B,h,w], K [B,4] -> [B,h,w,3]."""
    h, w = depth.shape[-2:]
    fx, fy, cx, cy = K[:, None, None].unbind(-1)
    y, x = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    return torch.stack([depth * ((x - cx) / fx), depth * ((y - cy) / fy), depth], -1)


def project(X, K):
    """reference projective_ops.py:11-22."""
    Xx, Xy, Z = X.unbind(-1)
    Z = Z + EPS
    fx, fy, cx, cy = K[:, None, None].unbind(-1)
    return torch.stack([fx * (Xx / Z) + cx, fy * (Xy / Z) + cy, 1.0 / Z], -1)


def induced_flow2d(Ts, depth, K):
    """reference projective_ops.py:55-68 (first output)."""
    X0 = inv_project(depth, K)
    return project(se3.act(Ts, X0), K) - project(X0, K)


def sample_bilinear(img, coords):
    """reference sampler_ops.py:9-28: img [B,1,h,w], coords [B,h,w,2] (x,y) pixel units,
    bilinear, zero padding, align_corners=True."""
    B, _, h, w = img.shape
    x, y = coords[..., 0], coords[..., 1]
    x0, y0 = torch.floor(x), torch.floor(y)
    ax, ay = x - x0, y - y0
    x0, y0 = x0.long(), y0.long()
    flat = img.reshape(B, h * w)

    def tap(ix, iy):
        ok = ((ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)).to(img.dtype)
        idx = (iy.clamp(0, h - 1) * w + ix.clamp(0, w - 1)).reshape(B, -1)
        return torch.gather(flat, 1, idx).view_as(ix) * ok

    return (1 - ax) * (1 - ay) * tap(x0, y0) + ax * (1 - ay) * tap(x0 + 1, y0) \
        + (1 - ax) * ay * tap(x0, y0 + 1) + ax * ay * tap(x0 + 1, y0 + 1)


# ----------------------------------------------------------------------------- update block
def conv_gru(sd, p, h, *inputs):
    """reference blocks/gru.py:9-35."""
    iz = sum(i[:, 0:128] for i in inputs)
    ir = sum(i[:, 128:256] for i in inputs)
    iq = sum(i[:, 256:384] for i in inputs)
    z = torch.sigmoid(conv(sd, p + ".convz1", h, 1, 1) + conv(sd, p + ".convz2", h, 1, 4, 4) + iz)
    r = torch.sigmoid(conv(sd, p + ".convr1", h, 1, 1) + conv(sd, p + ".convr2", h, 1, 4, 4) + ir)
    rh = r * h
    q = torch.tanh(conv(sd, p + ".convq1", rh, 1, 1) + conv(sd, p + ".convq2", rh, 1, 4, 4) + iq)
    return (1 - z) * h + z * q


def motion_info(flow, twist, dz):
    """reference raft3d.py:92-94 as CALLED from :238-240 (arguments dz/twist swapped at the call
    site): channels = [flow(2), 10*log(Ts)(6), 10*dz(1)], clamped to +-50.  [B,h,w,*] -> [B,9,h,w]."""
    return torch.cat([flow, 10 * twist, 10 * dz], -1).clamp(-50.0, 50.0).permute(0, 3, 1, 2)


def update_block(sd, p, net, inp, corr, minfo):
    """reference raft3d.py:44-106."""
    mot = conv(sd, p + ".flow_enc.2", F.relu(conv(sd, p + ".flow_enc.0", minfo, 1, 3)))
    cor = F.relu(conv(sd, p + ".corr_enc.0", corr, 1, 1))
    cor = conv(sd, p + ".corr_enc.4", F.relu(conv(sd, p + ".corr_enc.2", cor, 1, 1)))
    net = conv_gru(sd, p + ".gru", net, inp, cor, mot)

    def head(name):
        return conv(sd, f"{p}.{name}.2", F.relu(conv(sd, f"{p}.{name}.0", net, 1, 1)))

    return net, head("mask"), head("ae"), head("delta"), torch.sigmoid(head("weight"))


# ----------------------------------------------------------------------------- Gauss-Newton
def se3_build(T, ae, pts, target, weight, K, radius=32, chunk=256):
    """lietorch_extras.se3_build_inplace (reference call site se3_field.py:20-21; UNPINNED).

    For every pixel i with transform T_i and every pixel j with |yi-yj|,|xi-xj| <= radius:
      a_ij = sigmoid(-|ae_i - ae_j|^2)                    (ae already / 8; cf. attention_matrix
                                                            se3_field.py:115-126)
      Y = T_i X_j;  p = (fx Yx/Yz + cx, fy Yy/Yz + cy, 1/Yz);  r = target_j - p
      J = dp/d(xi) at exp(xi) T_i X_j, xi = [tau, phi]  (3x6)
      H_i += a_ij J^T diag(w_j) J ;  b_i += a_ij J^T diag(w_j) r
    pairs with X_j.z < MIN_DEPTH or Y.z < MIN_DEPTH are skipped.
    T [B,h,w,7], ae [B,C,h,w], pts/target/weight [B,3,h,w], K [B,4]
    -> H [B,6,6,h,w], b [B,6,1,h,w]."""
    B, h, w = T.shape[:3]
    N = h * w
    Hs, bs = [], []
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    yy, xx = yy.reshape(N), xx.reshape(N)
    for b_ in range(B):
        fx, fy, cx, cy = [float(v) for v in K[b_]]
        A = ae[b_].reshape(-1, N).t()  # [N,C]
        X = pts[b_].reshape(3, N).t()  # [N,3]
        tg = target[b_].reshape(3, N).t()
        wt = weight[b_].reshape(3, N).t()
        Tb = T[b_].reshape(N, 7)
        Hb = torch.zeros(N, 6, 6)
        bb = torch.zeros(N, 6)
        for i0 in range(0, N, chunk):
            i1 = min(N, i0 + chunk)
            Ti = Tb[i0:i1, None, :]  # [n,1,7]
            Y = se3.act(Ti, X[None])  # [n,N,3]
            win = ((yy[i0:i1, None] - yy[None]).abs() <= radius) & ((xx[i0:i1, None] - xx[None]).abs() <= radius)
            ok = win & (X[None, :, 2] >= MIN_DEPTH) & (Y[..., 2] >= MIN_DEPTH)
            d2 = ((A[i0:i1, None, :] - A[None]) ** 2).sum(-1)
            a = torch.sigmoid(-d2) * ok.to(A.dtype)
            Yx, Yy, Yz = Y.unbind(-1)
            Yz = torch.where(ok, Yz, torch.ones_like(Yz))
            d = 1.0 / Yz
            o, z = torch.ones_like(d), torch.zeros_like(d)
            Jx = fx * torch.stack([d, z, -Yx * d * d, -Yx * Yy * d * d, o + Yx * Yx * d * d, -Yy * d], -1)
            Jy = fy * torch.stack([z, d, -Yy * d * d, -(o + Yy * Yy * d * d), Yx * Yy * d * d, Yx * d], -1)
            Jz = torch.stack([z, z, -d * d, -Yy * d * d, Yx * d * d, z], -1)
            p = torch.stack([fx * Yx * d + cx, fy * Yy * d + cy, d], -1)
            r = tg[None] - p  # [n,N,3]
            for k, J in enumerate((Jx, Jy, Jz)):
                wk = (a * wt[None, :, k])  # [n,N]
                Hb[i0:i1] += torch.einsum("ij,ijp,ijq->ipq", wk, J, J)
                bb[i0:i1] += torch.einsum("ij,ijp->ip", wk * r[..., k], J)
        Hs.append(Hb.view(h, w, 6, 6).permute(2, 3, 0, 1))
        bs.append(bb.view(h, w, 6, 1).permute(2, 3, 0, 1))
    return torch.stack(Hs), torch.stack(bs)


def gn_solve(H, b, lm=1e-4, ep=10.0):
    """reference se3_field.py:162-167 damping + cholesky6x6_forward (UNPINNED): solves
    (H + (lm*H + ep) o I) dx = b per pixel.  H [B,6,6,h,w], b [B,6,1,h,w] -> dx [B,h,w,6]."""
    Hm = H.permute(0, 3, 4, 1, 2).clone()
    bm = b.permute(0, 3, 4, 1, 2)
    dg = torch.diagonal(Hm, dim1=-2, dim2=-1)
    dg += lm * dg + ep
    L = torch.linalg.cholesky(Hm.double())
    return torch.cholesky_solve(bm.double(), L).float().squeeze(-1)


def gn_step(T, ae, target, weight, depth, K):
    """reference se3_field.py:150-170 (step_inplace)."""
    pts = inv_project(depth, K).permute(0, 3, 1, 2).contiguous()
    H, b = se3_build(T, ae / 8.0, pts, target, weight, K)
    dx = gn_solve(H, b)
    return se3.compose(se3.exp(dx), T)


def cvx_upsample(data, mask):
    """reference se3_field.py:173-186: data [B,h,w,D], mask [B,576,h,w] -> [B,8h,8w,D]."""
    B, h, w, D = data.shape
    m = torch.softmax(mask.view(B, 9, 8, 8, h, w), 1)
    dp = F.pad(data.permute(0, 3, 1, 2), (1, 1, 1, 1))
    out = torch.zeros(B, D, 8, 8, h, w)
    for ky in range(3):
        for kx in range(3):
            out = out + m[:, ky * 3 + kx][:, None] * dp[:, :, None, None, ky:ky + h, kx:kx + w]
    return out.permute(0, 4, 2, 5, 3, 1).reshape(B, 8 * h, 8 * w, D)


def upsample_se3(T, mask):
    """reference se3_field.py:189-192."""
    return se3.exp(cvx_upsample(se3.log(T), mask))


# ----------------------------------------------------------------------------- splatting
def splat(T, depth, feat, K, radius, points_per_pixel=8):
    """Motion.transform_and_project (reference motion.py:82-130) = pytorch3d PointsRasterizer
    (K nearest-in-z points per pixel) + AlphaCompositor (UNPINNED).

    Point n = T_n * inv_project(depth)_n projects to (u, v) = (fx X/Z + cx, fy Y/Z + cy) with
    pixel (px, py)'s centre at (px + 0.5, py + 0.5) (pytorch3d NDC convention); it covers a
    pixel when dist^2 < R^2, R = radius * min(H, W) / (2 H) pixels; alpha = 1 - dist^2 / R^2.
    Per pixel the (<= 8) nearest points in z (ties -> lower point index) are composited front
    to back: out = sum_k alpha_k prod_{m<k}(1 - alpha_m) f_k.  Points with Z <= 0 are culled.
    Returns (feat_warp [B,C,H,W], depth_warp [B,1,H,W] = z of the nearest point, 0 if none)."""
    B, H, W = depth.shape
    C = feat.shape[1]
    R = radius * min(H, W) / (2.0 * H)
    outs, zs = [], []
    span = int(R + 1.5)
    for b_ in range(B):
        P = se3.act(T[b_], inv_project(depth[b_:b_ + 1], K[b_:b_ + 1])[0]).reshape(-1, 3)
        Fm = feat[b_].reshape(C, -1).t()
        fx, fy, cx, cy = [float(v) for v in K[b_]]
        Z = P[:, 2]
        okp = Z > 0
        Zs = torch.where(okp, Z, torch.ones_like(Z))
        u = fx * P[:, 0] / Zs + cx
        v = fy * P[:, 1] / Zs + cy
        okp = okp & torch.isfinite(u) & torch.isfinite(v) & (u.abs() < 1e7) & (v.abs() < 1e7)
        bx = torch.floor(torch.where(okp, u, torch.zeros_like(u)) - 0.5).long()
        by = torch.floor(torch.where(okp, v, torch.zeros_like(v)) - 0.5).long()
        pix, pid, al = [], [], []
        n = torch.arange(P.shape[0])
        for oy in range(-span + 1, span + 1):
            for ox in range(-span + 1, span + 1):
                px, py = bx + ox, by + oy
                d2 = (u - (px.float() + 0.5)) ** 2 + (v - (py.float() + 0.5)) ** 2
                hit = okp & (d2 < R * R) & (px >= 0) & (px < W) & (py >= 0) & (py < H)
                pix.append((py * W + px)[hit])
                pid.append(n[hit])
                al.append((1.0 - d2 / (R * R))[hit])
        pix, pid, al = torch.cat(pix), torch.cat(pid), torch.cat(al)
        # sort by (pixel, z, point index)
        o = torch.argsort(pid, stable=True)
        pix, pid, al = pix[o], pid[o], al[o]
        o = torch.argsort(Z[pid], stable=True)
        pix, pid, al = pix[o], pid[o], al[o]
        o = torch.argsort(pix, stable=True)
        pix, pid, al = pix[o], pid[o], al[o]
        first = torch.ones_like(pix, dtype=torch.bool)
        first[1:] = pix[1:] != pix[:-1]
        start = torch.cummax(torch.where(first, torch.arange(len(pix)), torch.zeros_like(pix)), 0)[0]
        rank = torch.arange(len(pix)) - start
        out = torch.zeros(H * W, C)
        trans = torch.ones(H * W)
        zb = torch.zeros(H * W)
        for k in range(points_per_pixel):
            sel = rank == k
            pk, ik, ak = pix[sel], pid[sel], al[sel]
            out[pk] += (trans[pk] * ak)[:, None] * Fm[ik]
            trans[pk] = trans[pk] * (1 - ak)
            if k == 0:
                zb[pk] = Z[ik]
        outs.append(out.t().reshape(C, H, W))
        zs.append(zb.reshape(1, H, W))
    return torch.stack(outs), torch.stack(zs)


# ----------------------------------------------------------------------------- RAFT3D / Motion
def raft3d_first(sd, p, image, state):
    """reference raft3d.py:203-206."""
    state["raft_feat"] = basic_encoder(sd, p + ".fnet", image)
    state["raft_netinp"] = hrnet_cnet(sd, p + ".cnet", image)


def raft3d(sd, p, image, depth_prev, depth_curr, K, state, iters=16, trace=None):
    """reference raft3d.py:190-280 (inference branch).  Returns (Ts_up [B,H,W,7],
    flow2d_est [B,H,W,3], weight_up [B,3,H,W])."""
    B, _, H, W = image.shape
    h, w = H // 8, W // 8
    fmap_prev, net_inp = state["raft_feat"], state["raft_netinp"]
    y0, x0 = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    coords0 = torch.stack([x0, y0], -1)[None].repeat(B, 1, 1, 1)
    T = se3.identity(B, h, w)
    fmap_curr = basic_encoder(sd, p + ".fnet", image)
    pyr = corr_pyramid(fmap_prev, fmap_curr)
    net, inp = torch.tanh(net_inp[:, :128]), torch.relu(net_inp[:, 128:])
    K8 = K / 8.0
    d1 = depth_prev[:, 3::8, 3::8]
    d2 = depth_curr[:, 3::8, 3::8]
    mask = weight = None
    for it in range(iters):
        xyz = project(se3.act(T, inv_project(d1, K8)), K8)
        coords1, zinv_proj = xyz[..., :2], xyz[..., 2:]
        zinv = sample_bilinear((1.0 / d2)[:, None], coords1)
        corr = corr_lookup(pyr, coords1.permute(0, 3, 1, 2).contiguous())
        minfo = motion_info(coords1 - coords0, se3.log(T), zinv.unsqueeze(-1) - zinv_proj)
        net, mask, ae, delta, weight = update_block(sd, p + ".update_block", net, inp, corr, minfo)
        target = (xyz.permute(0, 3, 1, 2) + delta).contiguous()
        T = gn_step(T, ae, target, weight, d1, K8)
        if trace is not None:
            trace.append(dict(corr=corr, minfo=minfo, net=net, ae=ae, delta=delta, weight=weight, T=T))
    T_up = upsample_se3(T, mask)
    flow2d = induced_flow2d(T_up, depth_prev, K)
    w_up = cvx_upsample(weight.permute(0, 2, 3, 1), mask).permute(0, 3, 1, 2)
    state["raft_feat"] = fmap_curr
    state["raft_netinp"] = hrnet_cnet(sd, p + ".cnet", image)
    return T_up, flow2d, w_up


def disp_to_depth(disp, K):
    """reference motion.py:154-165."""
    scale = BF_DEFAULT / K[0, 0]
    return torch.clip(scale * K[0, 0] / (disp + 1e-5), max=BF_DEFAULT, min=0)


def motion_forward(sd, state, outputs, intrinsics, iters=16, ds=4, p="motion", trace=None):
    """reference motion.py:132-209 (mutates state / outputs)."""
    img_curr = outputs["left_img"]
    if "memory" not in state:
        raft3d_first(sd, p + ".raft3d", img_curr, state)
        return
    B = outputs["pred_disp"].shape[0]
    K = torch.tensor(intrinsics, dtype=torch.float32)[None].expand(B, -1)
    img_prev, feat_prev, disp_prev = state["memory"]
    depth_prev = disp_to_depth(disp_prev, K)
    depth_curr = disp_to_depth(outputs["pred_disp"], K).squeeze(1)
    T_up, flow2d, conf = raft3d(sd, p + ".raft3d", img_curr, depth_prev, depth_curr, K, state, iters, trace)
    outputs["Ts"], outputs["flow2d_est_induced"], outputs["weight"] = T_up, flow2d, conf
    W = depth_curr.shape[-1]
    to_proj = torch.cat([img_prev, flow2d.permute(0, 3, 1, 2), conf], 1)
    warped, depth_warp = splat(T_up, depth_prev, to_proj, K, 2.0)
    scale = BF_DEFAULT / K[0, 0]
    disp_warp = scale * K[0, 0] / (depth_warp + 1e-5)
    disp_warp = torch.where(disp_warp > W, torch.zeros_like(disp_warp), disp_warp)
    o = ds // 2 - 1
    feat_warp, _ = splat(T_up[:, o::ds, o::ds], depth_prev[:, o::ds, o::ds], feat_prev, K.float() / ds, 4.0)
    state["memory"] = [warped[:, :3], feat_warp, warped[:, 6:], disp_warp, warped[:, 3:6]]
