// SE(3) device algebra on 7-float embeddings [t(3), q_xyzw(4)] -- the lietorch SE3 conventions the
// reference relies on (call sites raft3d.py:173,225,236; se3_field.py:157,169,189-192;
// projective_ops.py:48,59; motion.py:97,196).  Mirrors oracle/se3.py operation for operation.
#pragma once
#include <hip/hip_runtime.h>

#define SE3_EPS 1e-6f

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
struct SE3T { V3 t; Q4 q; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 scale3(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// R(q) v = v + w*uv + u x uv, uv = 2 u x v
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  const V3 u = V3{q.x, q.y, q.z};
  const V3 uv = scale3(2.f, cross3(u, v));
  return add3(add3(v, scale3(q.w, uv)), cross3(u, uv));
}
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  const V3 ua = V3{a.x, a.y, a.z}, ub = V3{b.x, b.y, b.z};
  const V3 c = cross3(ua, ub);
  return Q4{a.w * b.x + b.w * a.x + c.x, a.w * b.y + b.w * a.y + c.y, a.w * b.z + b.w * a.z + c.z,
            a.w * b.w - dot3(ua, ub)};
}
__device__ __forceinline__ V3 se3_act(const SE3T& T, V3 X) { return add3(qrot(T.q, X), T.t); }
__device__ __forceinline__ SE3T se3_compose(const SE3T& A, const SE3T& B) {
  return SE3T{add3(qrot(A.q, B.t), A.t), qmul(A.q, B.q)};
}
__device__ __forceinline__ SE3T se3_load(const float* p) {
  return SE3T{V3{p[0], p[1], p[2]}, Q4{p[3], p[4], p[5], p[6]}};
}
__device__ __forceinline__ void se3_store(float* p, const SE3T& T) {
  p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z; p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w;
}

__device__ __forceinline__ Q4 so3_exp(V3 phi) {
  const float th2 = dot3(phi, phi), th = sqrtf(th2), th4 = th2 * th2;
  float imag, real;
  if (th2 < SE3_EPS) {
    imag = 0.5f - th2 / 48.f + th4 / 3840.f;
    real = 1.f - th2 / 8.f + th4 / 384.f;
  } else {
    imag = sinf(0.5f * th) / th;
    real = cosf(0.5f * th);
  }
  return Q4{imag * phi.x, imag * phi.y, imag * phi.z, real};
}

__device__ __forceinline__ V3 so3_log(Q4 q) {
  const V3 u = V3{q.x, q.y, q.z};
  const float n2 = dot3(u, u), w = q.w;
  float f;
  if (n2 < SE3_EPS * SE3_EPS) {
    const float ws = fabsf(w) < SE3_EPS ? SE3_EPS : w;
    f = 2.f / ws - (2.f / 3.f) * n2 / (ws * ws * ws);
  } else {
    const float n = sqrtf(n2);
    if (fabsf(w) < SE3_EPS) f = (w > 0.f ? 3.14159265358979323846f : -3.14159265358979323846f) / n;
    else f = 2.f * atanf(n / w) / n;
  }
  return scale3(f, u);
}

// V(phi) v (inverse = false) or V(phi)^-1 v, V = I + c1 [phi]x + c2 [phi]x^2
__device__ __forceinline__ V3 left_jac_apply(V3 phi, V3 v, bool inverse) {
  const float th2 = dot3(phi, phi), th = sqrtf(th2);
  const bool small = th2 < SE3_EPS;
  const V3 pv = cross3(phi, v), ppv = cross3(phi, pv);
  if (!inverse) {
    const float c1 = small ? 0.5f - th2 / 24.f : (1.f - cosf(th)) / th2;
    const float c2 = small ? 1.f / 6.f - th2 / 120.f : (th - sinf(th)) / (th2 * th);
    return add3(add3(v, scale3(c1, pv)), scale3(c2, ppv));
  }
  const float half = 0.5f * th;
  const float c2 = small ? 1.f / 12.f : (1.f - th * cosf(half) / (2.f * sinf(half))) / th2;
  return add3(add3(v, scale3(-0.5f, pv)), scale3(c2, ppv));
}

__device__ __forceinline__ SE3T se3_exp(V3 tau, V3 phi) {
  return SE3T{left_jac_apply(phi, tau, false), so3_exp(phi)};
}
__device__ __forceinline__ void se3_log(const SE3T& T, V3* tau, V3* phi) {
  *phi = so3_log(T.q);
  *tau = left_jac_apply(*phi, T.t, true);
}

// pinhole geometry of RAFT3D (reference model/motion/raft3d/projective_ops.py:7-52): integer pixel centres
#define MIN_DEPTH 0.05f  // projective_ops.py:7
#define PEPS 1e-5f       // projective_ops.py:8
__device__ __forceinline__ V3 inv_project(float depth, int x, int y, float fx, float fy, float cx, float cy) {
  return V3{depth * (((float)x - cx) / fx), depth * (((float)y - cy) / fy), depth};
}
__device__ __forceinline__ V3 project(V3 X, float fx, float fy, float cx, float cy) {
  const float Z = X.z + PEPS;
  return V3{fx * (X.x / Z) + cx, fy * (X.y / Z) + cy, 1.f / Z};
}
