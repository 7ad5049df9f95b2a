"""ORACLE (test infrastructure, not product): SE(3) field algebra on [..., 7] tensors.

Restates the lietorch ``SE3`` ops the reference calls (lietorch is an un-vendored, un-pinned
dependency -- reference README.md:43; call sites raft3d.py:173,225,236, se3_field.py:157,169,
189-192, projective_ops.py:48,59, motion.py:97,196).  PARITY UNPINNED by the reference: no test
or fixture in the reference tree pins these; semantics follow lietorch's published
conventions and are validated by invariants in tests/test_oracle_se3.py:

  data   = [tx, ty, tz, qx, qy, qz, qw]   (translation, unit quaternion xyzw)
  tangent= [tau(3), phi(3)]
  exp    : q = Exp_SO3(phi), t = V(phi) tau       (V = SO3 left Jacobian)
  log    : phi = Log_SO3(q), tau = V(phi)^-1 t
  T1*T2  : q = q1 (x) q2, t = R(q1) t2 + t1
  T * X  : R(q) X + t
"""
import math

import torch

EPS = 1e-6  # lietorch common.h


def identity(*shape):
    T = torch.zeros(*shape, 7)
    T[..., 6] = 1.0
    return T


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def qrot(q, v):
    """R(q) v = v + w*uv + u x uv, uv = 2 u x v."""
    u, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * _cross(u, v)
    return v + w * uv + _cross(u, uv)


def qmul(a, b):
    ua, wa = a[..., :3], a[..., 3:4]
    ub, wb = b[..., :3], b[..., 3:4]
    u = wa * ub + wb * ua + _cross(ua, ub)
    w = wa * wb - (ua * ub).sum(-1, keepdim=True)
    return torch.cat([u, w], -1)


def act(T, X):
    return qrot(T[..., 3:], X) + T[..., :3]


def compose(T1, T2):
    return torch.cat([qrot(T1[..., 3:], T2[..., :3]) + T1[..., :3], qmul(T1[..., 3:], T2[..., 3:])], -1)


def so3_exp(phi):
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    th4 = th2 * th2
    small = th2 < EPS
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0 + th4 / 3840.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0 + th4 / 384.0, torch.cos(0.5 * ths))
    return torch.cat([imag * phi, real], -1)


def so3_log(q):
    u, w = q[..., :3], q[..., 3:4]
    n2 = (u * u).sum(-1, keepdim=True)
    n = n2.sqrt()
    small = n2 < EPS * EPS
    ns = torch.where(small, torch.ones_like(n), n)
    ws = torch.where(w.abs() < EPS, torch.full_like(w, EPS), w)
    big = torch.where(w.abs() < EPS,
                      torch.where(w > 0, math.pi / ns, -math.pi / ns),
                      2.0 * torch.atan(ns / ws) / ns)
    tiny = 2.0 / ws - (2.0 / 3.0) * n2 / (ws * ws * ws)
    return torch.where(small, tiny, big) * u


def _left_jac_apply(phi, v, inverse=False):
    """V(phi) v  or  V(phi)^-1 v  with V = I + c1 [phi]x + c2 [phi]x^2."""
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th2 < EPS
    ths = torch.where(small, torch.ones_like(th), th)
    th2s = torch.where(small, torch.ones_like(th2), th2)
    pv = _cross(phi, v)
    ppv = _cross(phi, pv)
    if not inverse:
        c1 = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / th2s)
        c2 = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (th2s * ths))
        return v + c1 * pv + c2 * ppv
    half = 0.5 * ths
    c2 = torch.where(small, torch.full_like(th2, 1.0 / 12.0),
                     (1.0 - ths * torch.cos(half) / (2.0 * torch.sin(half))) / th2s)
    return v - 0.5 * pv + c2 * ppv


def exp(xi):
    tau, phi = xi[..., :3], xi[..., 3:]
    return torch.cat([_left_jac_apply(phi, tau), so3_exp(phi)], -1)


def log(T):
    phi = so3_log(T[..., 3:])
    return torch.cat([_left_jac_apply(phi, T[..., :3], inverse=True), phi], -1)
