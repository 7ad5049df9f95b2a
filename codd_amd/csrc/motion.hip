// Motion / RAFT3D non-convolution kernels (fp32, HBM / LDS / VALU bound; the only MFMA use is the
// all-pairs correlation, which is routed through the conv family as a 1x1 convolution).
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include "se3.h"


// ------------------------------------------------------------------------------------------------
// InstanceNorm2d (affine = False, eps = 1e-5, biased variance) + residual + ReLU.
// Statistics: every (b, c) plane is split over `parts` workgroups that accumulate sum(x - K) and
// sum((x - K)^2) around the pivot K = x[0] of the plane (shifted moments: no cancellation when
// |mean| >> std); the apply kernel combines the partials in fp64 in a fixed order.
// ------------------------------------------------------------------------------------------------
#define IN_MAXPARTS 32
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* __restrict__ x, int HW, int parts,
                                                             double* __restrict__ partial) {
  __shared__ double red[2][4];
  const int bc = blockIdx.y, part = blockIdx.x;
  const float* p = x + (size_t)bc * HW;
  const float K = p[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (HW + parts - 1) / parts, i0 = part * per, i1 = min(HW, i0 + per);
  float s = 0.f, q = 0.f;
  for (int i = i0 + tid; i < i1; i += 256) { const float d = p[i] - K; s += d; q += d * d; }
  double sd = (double)s, qd = (double)q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); qd += __shfl_xor(qd, o, 64); }
  if (lane == 0) { red[0][wave] = sd; red[1][wave] = qd; }
  __syncthreads();
  if (tid == 0) {
    partial[((size_t)bc * parts + part) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[((size_t)bc * parts + part) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// One workgroup normalises 2048 consecutive elements of one plane (two float4 per thread); the plane's
// statistics are combined once per workgroup (first wave, fp64 shuffle reduction) instead of once per element.
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x,
                                                             const double* __restrict__ partial, int parts,
                                                             const float* __restrict__ res, int HW, int relu,
                                                             float* __restrict__ y) {
  __shared__ float st[2];
  const int bc = blockIdx.y;
  const float* p = x + (size_t)bc * HW;
  if (threadIdx.x < 64) {
    double s = 0.0, q = 0.0;
    if ((int)threadIdx.x < parts) {
      s = partial[((size_t)bc * parts + threadIdx.x) * 2];
      q = partial[((size_t)bc * parts + threadIdx.x) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (threadIdx.x == 0) {
      const double md = s / HW;                       // mean of (x - K)
      const double var = fmax(q / HW - md * md, 0.0);  // shift invariant
      st[0] = (float)((double)p[0] + md);
      st[1] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  __syncthreads();
  const float mean = st[0], rstd = st[1];
  const size_t base = (size_t)bc * HW;
  const bool vec = ((HW & 3) == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
                   (!res || ((uintptr_t)res & 15) == 0);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = (blockIdx.x * 512 + u * 256 + threadIdx.x) * 4;
    if (i >= HW) continue;
    if (vec) {
      float4 v = *(const float4*)(x + base + i);
      v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
      if (res) { const float4 r = *(const float4*)(res + base + i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *(float4*)(y + base + i) = v;
    } else {
      for (int c = 0; c < 4 && i + c < HW; ++c) {
        float v = (x[base + i + c] - mean) * rstd;
        if (res) v += res[base + i + c];
        if (relu) v = fmaxf(v, 0.f);
        y[base + i + c] = v;
      }
    }
  }
}

// The apply pass writing the NEXT convolution's input form: one thread = the 8 channels of an octet at one pixel ->
// one split-bf16 record (hi | lo) through xs_store8, optionally the fp32 tensor as well (residual operand of the block
// output).  v = norm(x); [relu]; [+ res; [relu]] -- the second ReLU is the one a residual block applies after adding
// its input (reference blocks/extractor.py:52-58), so InstanceNorm -> ReLU -> (+ x) -> ReLU -> re-layout is one launch.
__global__ __launch_bounds__(256) void instnorm_apply_xs_kernel(const float* __restrict__ x,
                                                                const double* __restrict__ partial, int parts,
                                                                const float* __restrict__ res, int C, int H, int W,
                                                                int relu, int relu2, float* __restrict__ y,
                                                                const codd_xs_view xs) {
  __shared__ float st[8][2];
  const int HW = H * W, oct = blockIdx.y, b = blockIdx.z;
  const int bc0 = b * C + oct * 8;
  if (threadIdx.x < 64) {
    for (int c = 0; c < 8; ++c) {
      double s = 0.0, q = 0.0;
      if ((int)threadIdx.x < parts) {
        s = partial[((size_t)(bc0 + c) * parts + threadIdx.x) * 2];
        q = partial[((size_t)(bc0 + c) * parts + threadIdx.x) * 2 + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
      if (threadIdx.x == 0) {
        const double md = s / HW;
        const double var = fmax(q / HW - md * md, 0.0);
        st[c][0] = (float)((double)x[(size_t)(bc0 + c) * HW] + md);
        st[c][1] = (float)(1.0 / sqrt(var + 1e-5));
      }
    }
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  float v[8], r[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = x[(size_t)(bc0 + c) * HW + i];
  if (res) {
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = res[(size_t)(bc0 + c) * HW + i];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float t = (v[c] - st[c][0]) * st[c][1];
    if (relu) t = fmaxf(t, 0.f);
    if (res) { t += r[c]; if (relu2) t = fmaxf(t, 0.f); }
    v[c] = t;
    if (y) y[(size_t)(bc0 + c) * HW + i] = t;
  }
  xs_store8(xs, b, oct, i / W, i % W, v);
}

extern "C" int codd_instnorm_xs(const float* x, int B, int C, int H, int W, float* stats, const float* res, int relu,
                                int relu_after_res, float* y, codd_xs_view xs, void* stream) {
  const int HW = H * W;
  if (!x || !stats || B < 1 || C < 8 || (C & 7) || HW < 1 || ((uintptr_t)stats & 7) || !xs_view_ok(xs, C, H, W))
    return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int parts = cdiv(2048, B * C);
  parts = parts < 1 ? 1 : (parts > IN_MAXPARTS ? IN_MAXPARTS : parts);
  if (parts > cdiv(HW, 1024)) parts = cdiv(HW, 1024);
  double* partial = (double*)stats;
  instnorm_stats_kernel<<<dim3(parts, B * C), 256, 0, s>>>(x, HW, parts, partial);
  CODD_LAUNCH_CHECK();
  instnorm_apply_xs_kernel<<<dim3(cdiv(HW, 256), C / 8, B), 256, 0, s>>>(x, partial, parts, res, C, H, W, relu,
                                                                         relu_after_res, y, xs);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_instnorm(const float* x, int B, int C, int HW, float* stats, const float* res, int relu,
                             float* y, void* stream) {
  if (!x || !stats || !y || B < 1 || C < 1 || HW < 1 || ((uintptr_t)stats & 7)) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int parts = cdiv(2048, B * C);  // ~2048 workgroups in flight
  parts = parts < 1 ? 1 : (parts > IN_MAXPARTS ? IN_MAXPARTS : parts);
  if (parts > cdiv(HW, 1024)) parts = cdiv(HW, 1024);
  double* partial = (double*)stats;
  instnorm_stats_kernel<<<dim3(parts, B * C), 256, 0, s>>>(x, HW, parts, partial);
  CODD_LAUNCH_CHECK();
  instnorm_apply_kernel<<<dim3(cdiv(HW, 2048), B * C), 256, 0, s>>>(x, partial, parts, res, HW, relu, y);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// avg_pool2d(2) (floor sizes) -- used to build the pooled feature maps of the correlation pyramid.
// ------------------------------------------------------------------------------------------------
__global__ void avgpool2_kernel(const float* __restrict__ in, int h, int w, float* __restrict__ out, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int w2 = w >> 1, h2 = h >> 1;
  const int x = (int)(e % w2);
  long long t = e / w2;
  const int y = (int)(t % h2);
  const long long bc = t / h2;
  const float* p = in + (size_t)bc * h * w + (size_t)(2 * y) * w + 2 * x;
  out[e] = (p[0] + p[1] + p[w] + p[w + 1]) * 0.25f;
}

extern "C" int codd_avgpool2(const float* in, int BC, int h, int w, float* out, void* stream) {
  if (!in || !out || h < 2 || w < 2) return CODD_EINVAL;
  const long long total = (long long)BC * (h >> 1) * (w >> 1);
  avgpool2_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, h, w, out, total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Correlation pyramid lookup (lietorch_extras.corr_index_forward; call site blocks/corr.py:10-18,47-54).
// Workgroup = 4 waves = 16 consecutive source pixels of one pyramid level.  A wave loads the 8x8 tap
// window of one pixel (lane = ty*8 + tx), forms the 7x7 bilinear outputs with lane shuffles, and the
// [49][16] result tile is written out through LDS as 64-byte runs.
// ------------------------------------------------------------------------------------------------
struct LookupGeom {  // coords == NULL: the projected coordinates are computed here (fused geometry)
  const float *T, *d1, *d2;
  float fx, fy, cx, cy;
  float *xyz, *minfo;
  codd_xs_view cxs, mxs;  // XS mode: correlation features / motion info written as split-bf16 records instead
};
__device__ __forceinline__ void raft_geometry_pixel(const float* __restrict__ T, const float* __restrict__ d1,
                                                    const float* __restrict__ d2, int b, int pix, int h, int w,
                                                    float fx, float fy, float cx, float cy, float* __restrict__ xyz,
                                                    float* __restrict__ minfo, const codd_xs_view* mxs = nullptr);

// pixels per wave of the lookup: each pixel is a chain of dependent gathers (project -> 4 corner loads -> blend), so
// fewer pixels per wave would mean shorter chains and more workgroups -- measured: 1 per wave 18.6 us, 4 per wave 17.6 us
#ifndef LOOKUP_PPW
#define LOOKUP_PPW 4
#endif
template <bool XS>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                          const float* __restrict__ l2, const float* __restrict__ l3,
                                                          const float* __restrict__ coords, int cstride, int h, int w,
                                                          float* __restrict__ out, const LookupGeom gm) {
  constexpr int PPW = LOOKUP_PPW, NPX = 4 * PPW;  // pixels per wave / per workgroup
  __shared__ float tile[49][NPX + 1];
  const int lvl = blockIdx.y, b = blockIdx.z;
  const int N = h * w;
  const int n0 = blockIdx.x * NPX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = h >> lvl, w2 = w >> lvl;
  const float* vol = (lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3) + (size_t)b * N * h2 * w2;
  const float inv = 1.f / (float)(1 << lvl);
  const int tx = lane & 7, ty = lane >> 3;
  // three unrolled phases over the wave's PPW pixels -- all pose / depth loads, then all volume gathers, then the
  // bilinear blends -- instead of PPW dependent (load -> project -> gather -> blend) chains one after the other: the
  // kernel is a latency chain, two L2 / HBM round trips per wave instead of 2 * PPW
  float x0[PPW], y0[PPW];
  bool live[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int n = n0 + wave * PPW + q;
    live[q] = n < N;
    const int nn = live[q] ? n : N - 1;
    float px, py;
    if (coords) {
      const float* cp = coords + ((size_t)b * N + nn) * cstride;
      px = cp[0]; py = cp[1];
    } else {  // every lane recomputes the projection of its wave's pixel (reference raft3d.py:225-227)
      const int yy = nn / w, xx = nn - yy * w;
      const SE3T Ti = se3_load(gm.T + ((size_t)b * N + nn) * 7);
      const V3 pp = project(se3_act(Ti, inv_project(gm.d1[(size_t)b * N + nn], xx, yy, gm.fx, gm.fy, gm.cx, gm.cy)), gm.fx,
                            gm.fy, gm.cx, gm.cy);
      px = pp.x; py = pp.y;
    }
    x0[q] = px * inv; y0[q] = py * inv;
  }
  float v[PPW], dxq[PPW], dyq[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int nn = live[q] ? n0 + wave * PPW + q : N - 1;
    float fx = floorf(x0[q]), fy = floorf(y0[q]);
    dxq[q] = x0[q] - fx; dyq[q] = y0[q] - fy;
    // keep the int conversion defined for wild coordinates; everything is out of range then
    fx = fminf(fmaxf(fx, -16.f), (float)w2 + 16.f);
    fy = fminf(fmaxf(fy, -16.f), (float)h2 + 16.f);
    const int ix = (int)fx - 3 + tx, iy = (int)fy - 3 + ty;
    const bool in = (unsigned)ix < (unsigned)w2 && (unsigned)iy < (unsigned)h2 && x0[q] == x0[q] && y0[q] == y0[q];
    const float t = vol[(size_t)nn * h2 * w2 + (size_t)(in ? iy : 0) * w2 + (in ? ix : 0)];  // unconditional, masked
    v[q] = in ? t : 0.f;
  }
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int pi = wave * PPW + q;
    const float dx = dxq[q], dy = dyq[q];
    const float vx = __shfl_down(v[q], 1, 64), vy = __shfl_down(v[q], 8, 64), vxy = __shfl_down(v[q], 9, 64);
    if (live[q] && tx < 7 && ty < 7) {
      const float r = ((1.f - dx) * (1.f - dy)) * v[q] + (dx * (1.f - dy)) * vx + ((1.f - dx) * dy) * vy + (dx * dy) * vxy;
      tile[tx * 7 + ty][pi] = r;  // channel = i*7 + j, i = x offset, j = y offset
    }
  }
  // fused geometry: the level-0 workgroups also publish xyz and the motion-info channels of their pixels
  if (!coords && lvl == 0 && tid < NPX && n0 + tid < N)
    raft_geometry_pixel(gm.T, gm.d1, gm.d2, b, n0 + tid, h, w, gm.fx, gm.fy, gm.cx, gm.cy, gm.xyz, gm.minfo,
                        XS ? &gm.mxs : nullptr);
  __syncthreads();
  if (XS) {  // channel c = lvl*49 + ch -> slot c & 7 of record octet c >> 3
    const codd_xs_view& d = gm.cxs;
    const int c0 = lvl * 49, c1 = c0 + 49;
    const int of = (c0 + 7) >> 3, ol = c1 >> 3;  // octets [of, ol) lie inside this level: whole 16-byte records
    for (int e = tid; e < (ol - of) * NPX; e += 256) {
      const int o = of + e / NPX, pi = e % NPX, n = n0 + pi;
      if (n >= N) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = tile[o * 8 + i - c0][pi];
      xs_store8(d, b, o, n / w, n % w, v);
    }
    // the octets shared with the neighbouring levels: 2-byte stores of this level's slots
    const size_t per = (size_t)d.c8 * d.hp * d.wp;
    unsigned short* base = (unsigned short*)d.ptr + (size_t)b * CODD_TERMS_PLANES(d.terms) * per * 8;
    const int nhead = of * 8 - c0, ntail = c1 - ol * 8;
    for (int e = tid; e < (nhead + ntail) * NPX; e += 256) {
      const int q = e / NPX, pi = e % NPX, n = n0 + pi;
      if (n >= N) continue;
      const int c = q < nhead ? c0 + q : ol * 8 + (q - nhead), y = n / w, x = n - y * w;
      const float v = tile[c - c0][pi];
      const __bf16 hi = (__bf16)v;
      const size_t at = (((size_t)(d.o8 + (c >> 3)) * d.hp + (y + d.bt)) * d.wp + (x + d.bl)) * 8 + (c & 7);
      base[at] = xs_elem16(v, CODD_TERMS_IS_F16(d.terms));
      if (d.terms == 3) base[per * 8 + at] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)hi));
      if (d.terms == CODD_TERMS_SPLIT_F16) base[per * 8 + at] = xs_elem16(v - (float)(_Float16)v, true);
    }
    return;
  }
  for (int e = tid; e < 49 * NPX; e += 256) {
    const int ch = e / NPX, pi = e % NPX, n = n0 + pi;
    if (n < N) out[((size_t)b * 196 + lvl * 49 + ch) * N + n] = tile[ch][pi];
  }
}

extern "C" int codd_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                const float* coords, int cstride, int B, int h, int w, float* out, void* stream) {
  if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out || cstride < 2) return CODD_EINVAL;
  dim3 grid(cdiv(h * w, 4 * LOOKUP_PPW), 4, B);
  LookupGeom gm;
  memset(&gm, 0, sizeof(gm));
  corr_lookup_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(lvl0, lvl1, lvl2, lvl3, coords, cstride, h, w, out, gm);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_raft_geometry_lookup(const float* T, const float* depth1, const float* depth2, const float* lvl0,
                                         const float* lvl1, const float* lvl2, const float* lvl3, int B, int h, int w,
                                         float fx, float fy, float cx, float cy, float* xyz, float* minfo, float* out,
                                         void* stream) {
  if (!T || !depth1 || !depth2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || !xyz || !minfo || !out) return CODD_EINVAL;
  LookupGeom gm = {T, depth1, depth2, fx, fy, cx, cy, xyz, minfo, {}, {}};
  dim3 grid(cdiv(h * w, 4 * LOOKUP_PPW), 4, B);
  corr_lookup_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(lvl0, lvl1, lvl2, lvl3, nullptr, 0, h, w, out, gm);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_raft_geometry_lookup_xs(const float* T, const float* depth1, const float* depth2, const float* lvl0,
                                            const float* lvl1, const float* lvl2, const float* lvl3, int B, int h, int w,
                                            float fx, float fy, float cx, float cy, float* xyz, codd_xs_view minfo_xs,
                                            codd_xs_view corr_xs, void* stream) {
  if (!T || !depth1 || !depth2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || !xyz || !xs_view_ok(minfo_xs, 9, h, w) ||
      !xs_view_ok(corr_xs, 196, h, w))
    return CODD_EINVAL;
  LookupGeom gm = {T, depth1, depth2, fx, fy, cx, cy, xyz, nullptr, corr_xs, minfo_xs};
  dim3 grid(cdiv(h * w, 4 * LOOKUP_PPW), 4, B);
  corr_lookup_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(lvl0, lvl1, lvl2, lvl3, nullptr, 0, h, w, nullptr, gm);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-iteration geometry (reference raft3d.py:225-240).
// ------------------------------------------------------------------------------------------------

// full per-pixel geometry of pixel n (batch b): writes xyz[n] and the 9 motion-info channels
__device__ __forceinline__ void raft_geometry_pixel(const float* __restrict__ T, const float* __restrict__ d1,
                                                    const float* __restrict__ d2, int b, int pix, int h, int w,
                                                    float fx, float fy, float cx, float cy, float* __restrict__ xyz,
                                                    float* __restrict__ minfo, const codd_xs_view* mxs) {
  const int N = h * w, n = b * N + pix, y = pix / w, x = pix - y * w;
  const SE3T Ti = se3_load(T + (size_t)n * 7);
  const V3 X1 = se3_act(Ti, inv_project(d1[n], x, y, fx, fy, cx, cy));
  const V3 p = project(X1, fx, fy, cx, cy);
  xyz[(size_t)n * 3] = p.x; xyz[(size_t)n * 3 + 1] = p.y; xyz[(size_t)n * 3 + 2] = p.z;
  // bilinear sample of 1/depth2 at (p.x, p.y), zeros padding, align_corners = True
  float zinv = 0.f;
  if (p.x == p.x && p.y == p.y) {
    float fx0 = floorf(p.x), fy0 = floorf(p.y);
    const float ax = p.x - fx0, ay = p.y - fy0;
    fx0 = fminf(fmaxf(fx0, -4.f), (float)w + 4.f);
    fy0 = fminf(fmaxf(fy0, -4.f), (float)h + 4.f);
    const int ix = (int)fx0, iy = (int)fy0;
    const float* dp = d2 + (size_t)b * N;
    auto tap = [&](int xx, int yy) -> float {
      return ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) ? 1.f / dp[yy * w + xx] : 0.f;
    };
    zinv = (1.f - ax) * (1.f - ay) * tap(ix, iy) + ax * (1.f - ay) * tap(ix + 1, iy) +
           (1.f - ax) * ay * tap(ix, iy + 1) + ax * ay * tap(ix + 1, iy + 1);
  } else {
    zinv = p.x + p.y;  // propagate NaN like grid_sample would
  }
  V3 tau, phi;
  se3_log(Ti, &tau, &phi);
  const float vals[9] = {p.x - (float)x, p.y - (float)y, 10.f * tau.x, 10.f * tau.y, 10.f * tau.z,
                         10.f * phi.x, 10.f * phi.y, 10.f * phi.z, 10.f * (zinv - p.z)};
  if (mxs) {  // the 9 channels as split-bf16 records: octet 0 and slot 0 of octet 1 (the other 7 slots stay zero)
    float v0[8], v1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { v0[c] = fminf(fmaxf(vals[c], -50.f), 50.f); v1[c] = 0.f; }
    v1[0] = fminf(fmaxf(vals[8], -50.f), 50.f);
    xs_store8(*mxs, b, 0, y, x, v0);
    xs_store8(*mxs, b, 1, y, x, v1);
    return;
  }
  float* mp = minfo + (size_t)b * 9 * N + pix;
#pragma unroll
  for (int c = 0; c < 9; ++c) mp[(size_t)c * N] = fminf(fmaxf(vals[c], -50.f), 50.f);
}

__global__ void raft_geometry_kernel(const float* __restrict__ T, const float* __restrict__ d1,
                                     const float* __restrict__ d2, int B, int h, int w, float fx, float fy, float cx,
                                     float cy, float* __restrict__ xyz, float* __restrict__ minfo) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = h * w;
  if (n >= B * N) return;
  raft_geometry_pixel(T, d1, d2, n / N, n % N, h, w, fx, fy, cx, cy, xyz, minfo);
}

extern "C" int codd_raft_geometry(const float* T, const float* depth1, const float* depth2, int B, int h, int w,
                                  float fx, float fy, float cx, float cy, float* xyz, float* minfo, void* stream) {
  if (!T || !depth1 || !depth2 || !xyz || !minfo) return CODD_EINVAL;
  raft_geometry_kernel<<<cdiv((long long)B * h * w, 128), 128, 0, (hipStream_t)stream>>>(T, depth1, depth2, B, h, w, fx,
                                                                                        fy, cx, cy, xyz, minfo);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Dense SE3 Gauss-Newton step (reference se3_field.py:150-170):
//   a_ij = sigmoid(-|ae_i - ae_j|^2), |.|^2 expanded as |a_i|^2 + |a_j|^2 - 2 <a_i, a_j>,
//   H_i = sum_j a_ij J_ij^T W_j J_ij,  b_i = sum_j a_ij J_ij^T W_j r_ij  over the (2r+1)^2 window.
// Three launches:
//   prep   packs everything a neighbour j contributes -- ae_j/8 (32), X_j (3), target_j (3), weight_j (3),
//          |ae_j/8|^2 -- into one 176-byte record.
//   build  lane = pixel i of an 8x8 tile; all 64 lanes of a wave visit the same j, so the record is read with
//          SCALAR loads (wave-uniform address -> s_load_dwordx16, SGPR pairs feed v_pk_fma_f32 directly): no LDS,
//          no broadcast traffic.  The kernel is VALU-issue bound; its loop is written on float2 values so that the
//          32-term dot product is 16 packed FMAs and the 49 + 13 normal-equation updates are 23 + 8 (measured on
//          gfx950 at 4 waves/SIMD: v_fma_f32 4.2 cycles per wave instruction, v_pk_fma_f32 5.2 -- tools/ubench/
//          valu_rate.hip; 111 VALU instructions per neighbour against 197 for the scalar form).
//          Work decomposition: a tile's clipped neighbourhood (1600 ... 5184 neighbours at 72x120, r = 32) is a
//          row-major list cut into G = nj / q4 equal pieces, one per workgroup, and each piece into 4, one per
//          wave: every wave of the launch carries the same number of pairs, and the ~8000 waves are dispatched
//          dynamically over the 4096 wave slots.  The 4 waves of a workgroup add their sums through LDS (fixed
//          order) and write ONE [27][64] partial.
//   solve  one workgroup per tile: 14 waves add the tile's G partials in index order (deterministic), wave 0
//          damps, solves (Cholesky, fp64) and retracts its 64 pixels.
// ------------------------------------------------------------------------------------------------
#define GN_AE 32
#define GN_JS 44  // floats per neighbour record (16-byte aligned)
#define GN_WAVES 4
#define GN_SOLVE_WAVES 14  // 27 sums over 14 waves: <= 2 each
// the pair builder's second record image (se3_gn_build3_kernel): [b][y][x / 2][k = 0..11][x & 1], k = X (3), target in
// normalised image coordinates ((u - cx) / fx, (v - cy) / fy) and inverse depth, weights (wx fx^2, wy fy^2, wz), |a|^2;
// the phantom partner of the last pixel of an odd-width row is written as zeros (depth 0 = masked, finite)
static __device__ __forceinline__ void gn_geo2_store(float* __restrict__ geo2, int b, int h, int w, int yj, int xj, V3 X,
                                                      float tx, float ty, float tz, float wx, float wy, float wz,
                                                      float fx, float fy, float cx, float cy) {
  if (!geo2) return;
  const int wp2 = (w + 1) >> 1;
  float* gp = geo2 + (((size_t)b * h + yj) * wp2 + (xj >> 1)) * 24 + (xj & 1);
  const float v[9] = {X.x, X.y, X.z, (tx - cx) / fx, (ty - cy) / fy, tz, wx * fx * fx, wy * fy * fy, wz};
#pragma unroll
  for (int k = 0; k < 9; ++k) gp[2 * k] = v[k];
  gp[20] = 0.f; gp[22] = 0.f;
  if (xj == w - 1 && !(xj & 1)) {
#pragma unroll
    for (int k = 0; k < 12; ++k) gp[2 * k + 1] = 0.f;
  }
}
static __device__ __forceinline__ void gn_geo2_store_a2(float* __restrict__ geo2, int b, int h, int w, int yj, int xj, float a2) {
  if (!geo2) return;
  const int wp2 = (w + 1) >> 1;
  geo2[(((size_t)b * h + yj) * wp2 + (xj >> 1)) * 24 + (xj & 1) + 18] = a2;
}
__global__ void se3_gn_prep_kernel(const float* __restrict__ ae, int ae_c, const float* __restrict__ xyz,
                                   const float* __restrict__ delta, const float* __restrict__ wgt,
                                   const float* __restrict__ d1, int h, int w, float fx, float fy, float cx, float cy,
                                   float* __restrict__ jd, int* __restrict__ cnt, int ntiles, float* __restrict__ geo2) {
  const int N = h * w;
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j < ntiles) cnt[b * ntiles + j] = 0;  // arrival counters of the builder launched next (ntiles <= N)
  if (j >= N) return;
  const int yj = j / w, xj = j - yj * w;
  const float* aeb = ae + (size_t)b * ae_c * N;
  float* rp = jd + ((size_t)b * N + j) * GN_JS;
  float av[GN_AE];
#pragma unroll
  for (int c = 0; c < GN_AE; ++c) av[c] = aeb[(size_t)min(c, ae_c - 1) * N + j];
  float a2 = 0.f;
#pragma unroll
  for (int c = 0; c < GN_AE; ++c) {
    const float v = av[c] * (c < ae_c ? 0.125f : 0.f);
    rp[c] = v;
    a2 += v * v;
  }
  const V3 X = inv_project(d1[(size_t)b * N + j], xj, yj, fx, fy, cx, cy);
  const float* xb = xyz + ((size_t)b * N + j) * 3;
  const float* db = delta + (size_t)b * 3 * N + j;
  const float* wb = wgt + (size_t)b * 3 * N + j;
  rp[32] = X.x; rp[33] = X.y; rp[34] = X.z;
  rp[35] = xb[0] + db[0]; rp[36] = xb[1] + db[N]; rp[37] = xb[2] + db[2 * N];
  rp[38] = wb[0]; rp[39] = wb[N]; rp[40] = wb[2 * N];
  rp[41] = a2; rp[42] = 0.f; rp[43] = 0.f;
  gn_geo2_store(geo2, b, h, w, yj, xj, X, rp[35], rp[36], rp[37], rp[38], rp[39], rp[40], fx, fy, cx, cy);
  gn_geo2_store_a2(geo2, b, h, w, yj, xj, a2);
}

// ------------------------------------------------------------------------------------------------
// The three 1x1 heads (ae 256->32, delta 256->3, weight 256->3 + sigmoid; reference raft3d.py:59-61,100-104) fused
// with the record packing above: the 768 hidden channels arrive as split-bf16 records (written by the 3x3 head
// convolution), x = hi + lo is rebuilt in fp32 and multiplied with the fp32 weights -- the heads' outputs never
// exist as tensors (except weight, which the caller up-samples after the last update).
// ------------------------------------------------------------------------------------------------
// One wave = 16 pixels, three 16-row MFMA tiles (v_mfma_f32_16x16x32_bf16, rows = head outputs, columns = pixels):
// tiles 0/1 = the 32 ae rows over hidden channels 0..255 (8 k-steps), tile 2 = [delta 3 rows | weight 3 rows | 0]
// over channels 256..767 (16 k-steps, zero weights where a row does not read a channel).  A 16-byte record (8 channels
// of one pixel) IS the B operand of lane (pixel = lane % 16, k-group = lane / 16), so the hidden channels go from
// global memory straight into the MFMA; the weights are pre-packed on the host as A operands
// [32 (tile, k-step) blocks][plane hi|lo][lane][8 bf16] (64 KB, L2-resident).  Same 3-term split arithmetic as the
// convolution kernel (conv_bf16_kernel.h).  Lane (g = lane / 16) ends up with rows 4g..4g+3 of every tile for its
// pixel = 4 consecutive floats of the record.
// Workgroup = TWO waves over the same 16 pixels: wave 0 owns the two ae tiles (hidden channels 0..255: 8 k-steps),
// wave 1 the delta | weight tile (channels 256..767: 16 k-steps) -- every accumulator chain is the single-wave kernel's
// own (same order, same bits), the launch has twice the waves and two thirds of the longest dependent chain.
__global__ __launch_bounds__(128) void gn_heads_prep_kernel(const codd_xs_view hs, const uint4* __restrict__ Wp,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ xyz, const float* __restrict__ d1,
                                                          int h, int w, float fx, float fy, float cx, float cy,
                                                          float* __restrict__ jd, float* __restrict__ wout,
                                                          int* __restrict__ cnt, int ntiles, float* __restrict__ geo2) {
  const int N = h * w;
  const int lane = threadIdx.x & 63, px = lane & 15, g = lane >> 4, b = blockIdx.y;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0)  // arrival counters of the builder launched next
    for (int i = blockIdx.x; i < ntiles; i += gridDim.x) cnt[b * ntiles + i] = 0;
  const int n = blockIdx.x * 16 + px;
  const bool ok = n < N;
  const int j = ok ? n : N - 1;
  const int yj = j / w, xj = j - yj * w;
  const size_t per = (size_t)hs.c8 * hs.hp * hs.wp, ostride = (size_t)hs.hp * hs.wp;
  const bool three = CODD_TERMS_PLANES(hs.terms) == 2, f16 = CODD_TERMS_IS_F16(hs.terms);  // (f16: head_w holds fp16 A operands too)
  const uint4* src = (const uint4*)hs.ptr + (size_t)b * (three ? 2 : 1) * per +
                     ((size_t)(hs.o8 + g) * hs.hp + (yj + hs.bt)) * hs.wp + (xj + hs.bl);
  const uint4* wl = Wp + lane;
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // k-steps 0..7: channels 0..255 -> tiles 0, 1;  k-steps 8..23: channels 256..767 -> tile 2.  Three phases of 8
  // k-steps; the loads of a phase are all in flight before the MFMAs of the previous one are issued (the compiler's
  // own order is load-wait-MFMA per k-step, 24 exposed latencies) -- sched_barriers pin the phases.
  uint4 xh[2][8], xl[2][8], wh[2][16], wlo[2][16];
#define HEADS_LOAD(P, BUF)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                     \
      xh[BUF][s] = src[(size_t)4 * (8 * (P) + s) * ostride];                                            \
      if (three) xl[BUF][s] = src[per + (size_t)4 * (8 * (P) + s) * ostride];                           \
    }                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < ((P) == 0 ? 16 : 8); ++q) {                                   \
      const uint4* wb = wl + (size_t)((P) == 0 ? q : 8 + 8 * (P) + q) * 128;                            \
      wh[BUF][q] = wb[0];                                                                               \
      if (three) wlo[BUF][q] = wb[64];                                                                  \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
#define HEADS_MMA(T, WQ, BUF, S)                                                                        \
  {                                                                                                     \
    const codd_bf16x8 ah = __builtin_bit_cast(codd_bf16x8, wh[BUF][WQ]);                                \
    const codd_bf16x8 bh = __builtin_bit_cast(codd_bf16x8, xh[BUF][S]);                                 \
    if (three) {                                                                                        \
      const codd_bf16x8 al = __builtin_bit_cast(codd_bf16x8, wlo[BUF][WQ]);                             \
      const codd_bf16x8 bl = __builtin_bit_cast(codd_bf16x8, xl[BUF][S]);                               \
      if (f16) {                                                                                        \
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(codd_f16x8, al),             \
                                                        __builtin_bit_cast(codd_f16x8, bh), acc[T], 0, 0, 0); \
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(codd_f16x8, ah),             \
                                                        __builtin_bit_cast(codd_f16x8, bl), acc[T], 0, 0, 0); \
      } else {                                                                                          \
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[T], 0, 0, 0);                      \
        acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[T], 0, 0, 0);                      \
      }                                                                                                 \
    }                                                                                                   \
    if (f16)                                                                                            \
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(codd_f16x8, ah),               \
                                                      __builtin_bit_cast(codd_f16x8, bh), acc[T], 0, 0, 0); \
    else                                                                                                \
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[T], 0, 0, 0);                        \
  }
  float* rp = jd + ((size_t)b * N + j) * GN_JS;
  if (role == 1) {
    HEADS_LOAD(1, 1)
    HEADS_LOAD(2, 0)
#pragma unroll
    for (int s = 0; s < 8; ++s) HEADS_MMA(2, s, 1, s)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) HEADS_MMA(2, s, 0, s)
    // tile 2: g = 0 holds (delta0, delta1, delta2, weight0), g = 1 holds (weight1, weight2, 0, 0)
    const float w1 = __shfl_down(acc[2][0], 16, 64), w2 = __shfl_down(acc[2][1], 16, 64);
    if (g == 0 && ok) {
      const V3 X = inv_project(d1[(size_t)b * N + j], xj, yj, fx, fy, cx, cy);
      const float* xb = xyz + ((size_t)b * N + j) * 3;
      const float dl0 = acc[2][0] + bias[32], dl1 = acc[2][1] + bias[33], dl2 = acc[2][2] + bias[34];
      float wv[3] = {acc[2][3] + bias[35], w1 + bias[36], w2 + bias[37]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        wv[r] = 1.f / (1.f + expf(-wv[r]));
        wout[((size_t)b * 3 + r) * N + j] = wv[r];
      }
      *(f32x4*)(rp + 32) = f32x4{X.x, X.y, X.z, xb[0] + dl0};
      *(f32x4*)(rp + 36) = f32x4{xb[1] + dl1, xb[2] + dl2, wv[0], wv[1]};
      rp[40] = wv[2];  // (rp[41] = |a|^2: wave 0)
      *(float2*)(rp + 42) = float2{0.f, 0.f};
      gn_geo2_store(geo2, b, h, w, yj, xj, X, xb[0] + dl0, xb[1] + dl1, xb[2] + dl2, wv[0], wv[1], wv[2], fx, fy, cx, cy);
    }
    return;
  }
  HEADS_LOAD(0, 0)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    HEADS_MMA(0, s, 0, s)
    HEADS_MMA(1, 8 + s, 0, s)
  }
#undef HEADS_LOAD
#undef HEADS_MMA
  // ae rows 16 t + 4 g + r -> record floats [16 t + 4 g, +4)
  float a2 = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (acc[t][r] + bias[16 * t + 4 * g + r]) * 0.125f;
      a2 += v[r] * v[r];
    }
    if (ok) *(f32x4*)(rp + 16 * t + 4 * g) = v;
  }
  a2 += __shfl_xor(a2, 16, 64);
  a2 += __shfl_xor(a2, 32, 64);
  if (g == 0 && ok) {
    rp[41] = a2;
    gn_geo2_store_a2(geo2, b, h, w, yj, xj, a2);
  }
}

typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2f GN_PK(v2f a, v2f b, v2f c) {  // -> v_pk_fma_f32: 2 fp32 FMAs per lane and issue slot
  return __builtin_elementwise_fma(a, b, c);
}

// Workgroups a tile's nj neighbours are split into: ~q4 neighbours per workgroup (4 waves), so that every wave of
// the launch carries the same number of (i, j) pairs whatever the clipping of its tile's neighbourhood.
static __host__ __device__ __forceinline__ int gn_groups(int nj, int q4, int gmax) {
  const int g = (nj + q4 / 2) / q4;
  return g < 1 ? 1 : (g > gmax ? gmax : g);
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Pair builder (round 4): TWO neighbours per loop step, every operation of the geometry a packed fp32 instruction over
// the pair (v_pk_fma / v_pk_mul / v_pk_add_f32: lane half = neighbour parity), and the normal equations in the factored
// form of the Jacobian (reference se3_field.py:115-147; lietorch's left perturbation):
//     J = A [I | -[Y]x],  A = d(u, v, 1/z)/dY = d * A^,  A^ = [fx 0 -fx xn; 0 fy -fy yn; 0 0 -d],  Y = (xn, yn, 1) / d
//     S^ = a A^T W A^  (5 non-zeros: S00 S11 S02 S12 S22),   g^ = a A^T W r   (fx, fy, cx, cy folded into the records)
//     H_tt = d^2 S^,   H_tr = d N,   H_rr = Q^T N,   N = S^ Q,   Q = -[(xn, yn, 1)]x;   b_t = d g^,   b_r = Q^T g^
// i.e. 80 packed instructions per PAIR for the geometry where the J-entry form above issues 76 (31 of them packed) per
// NEIGHBOUR.  The pair's per-neighbour scalars (X, target, weights, |a|^2) come from a second record image
// ``geo2`` [b][y][x / 2][12][2] (both prep kernels write it) so that every packed source is an aligned SGPR pair; the
// embeddings stay in the 176-byte records.  Sums are kept per parity (2 x 26 accumulators) and added at the end.
// ------------------------------------------------------------------------------------------------
#define GN_G2 24  // floats per neighbour PAIR in geo2: [k = 0..11][parity]: X(3) target(3) weight(3) |a|^2 0 0
__global__ __launch_bounds__(64 * GN_WAVES) void se3_gn_build3_kernel(
    const float* __restrict__ T, const float* __restrict__ jd, const float* __restrict__ geo2, int h, int w, float fx,
    float fy, float cx, float cy, int radius, int tiles_x, int ntiles, int q4, int gmax, float* __restrict__ part) {
  __shared__ float red[27][64];
  const int N = h * w, wp2 = (w + 1) >> 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int tx0 = (tile % tiles_x) * 8, ty0 = (tile / tiles_x) * 8;
  const int ylo = max(ty0 - radius, 0), yhi = min(ty0 + 7 + radius, h - 1);
  const int xlo = max(tx0 - radius, 0), xhi = min(tx0 + 7 + radius, w - 1);
  const int ncols = xhi - xlo + 1, nj = (yhi - ylo + 1) * ncols;
  const int G = gn_groups(nj, q4, gmax);
  if (g >= G) return;
  const int slot = g * GN_WAVES + wave, nslots = G * GN_WAVES;
  const int s0 = (int)((long long)nj * slot / nslots), s1 = (int)((long long)nj * (slot + 1) / nslots);
  const int ys = ylo + s0 / ncols, xs = xlo + s0 % ncols;
  const int ye = ylo + (s1 - 1) / ncols, xe = xlo + (s1 - 1) % ncols;

  const int xi = tx0 + (lane & 7), yi = ty0 + (lane >> 3);
  const bool vi = xi < w && yi < h;
  const int i = vi ? yi * w + xi : 0;
  const int xim = xi - radius, yim = yi - radius;
  const unsigned twor = 2u * (unsigned)radius;
  const float* rec = jd + (size_t)b * N * GN_JS;
  const float* aip = rec + (size_t)i * GN_JS;
  const SE3T Ti = se3_load(T + ((size_t)b * N + i) * 7);
  const V3 c0 = qrot(Ti.q, V3{1.f, 0.f, 0.f}), c1 = qrot(Ti.q, V3{0.f, 1.f, 0.f}), c2 = qrot(Ti.q, V3{0.f, 0.f, 1.f});
  v2f ai[GN_AE / 2];
#pragma unroll
  for (int c = 0; c < GN_AE / 2; ++c) ai[c] = *(const v2f*)(aip + 2 * c);
  const float ai2 = aip[41];
  const v2f Z = {0.f, 0.f};
  v2f H00 = Z, H11 = Z, H02 = Z, H12 = Z, H22 = Z, H03 = Z, H04 = Z, H05 = Z, H13 = Z, H14 = Z, H15 = Z, H23 = Z, H24 = Z,
      H25 = Z, H33 = Z, H34 = Z, H35 = Z, H44 = Z, H45 = Z, H55 = Z, b0 = Z, b1 = Z, b2 = Z, b3 = Z, b4 = Z, b5 = Z;
#define BC(s) ((v2f){(s), (s)})

  // Walk over the wave's pairs, row by row.  The records of a pair occupy 84 SGPRs (2 x 32 embedding values + 10 geometry
  // pairs): five of the six 16-register tuples a wave has, so the compiler requests them in three or four separately
  // awaited portions, and they cannot be requested a step ahead -- every attempt (next pair's records loaded into the
  // registers the step has just finished with; scalar-cache warm-up loads; the embeddings through the wave's own LDS
  // region with broadcast ds_read_b128) ended in SGPR spills through v_readlane / v_writelane, in vector loads, or at 2
  // waves per SIMD and slower than this form (ROCm 7.2; DESIGN finding 45).
  if (s1 > s0)
  for (int yj = ys; yj <= ye; ++yj) {
    const bool rowin = vi && (unsigned)(yj - yim) <= twor;
    const float* rrow = rec + (size_t)yj * w * GN_JS;
    const v2f* grow = (const v2f*)(geo2 + ((size_t)b * h + yj) * wp2 * GN_G2);
    const int xa = yj == ys ? xs : xlo, xb = yj == ye ? xe : xhi;
    for (int xp = xa >> 1; xp <= (xb >> 1); ++xp) {
      const int x0 = 2 * xp, x1 = x0 + 1;
      const bool own0 = x0 >= xa, own1 = x1 <= xb;  // (x0 <= xb and x1 >= xa always hold)
      const float4* e0 = (const float4*)(rrow + (size_t)x0 * GN_JS);
      const float4* e1 = (const float4*)(rrow + (size_t)min(x1, w - 1) * GN_JS);
      const v2f* gp = grow + (size_t)xp * (GN_G2 / 2);
      v2f G2[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) G2[k] = gp[k];
      float4 r0[8], r1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r0[q] = e0[q];
#pragma unroll
      for (int q = 0; q < 8; ++q) r1[q] = e1[q];
      const v2f Xx = G2[0], Xy = G2[1], Xz = G2[2];
      const v2f Yz = GN_PK(BC(c0.z), Xx, GN_PK(BC(c1.z), Xy, GN_PK(BC(c2.z), Xz, BC(Ti.t.z))));
      const bool in0 = rowin & own0 & ((unsigned)(x0 - xim) <= twor) & (Xz.x >= MIN_DEPTH) & (Yz.x >= MIN_DEPTH);
      const bool in1 = rowin & own1 & ((unsigned)(x1 - xim) <= twor) & (Xz.y >= MIN_DEPTH) & (Yz.y >= MIN_DEPTH);
      v2f p0 = Z, p1 = Z, q0 = Z, q1 = Z;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        p0 = GN_PK(ai[2 * q], (v2f){r0[q].x, r0[q].y}, p0);
        p1 = GN_PK(ai[2 * q + 1], (v2f){r0[q].z, r0[q].w}, p1);
        q0 = GN_PK(ai[2 * q], (v2f){r1[q].x, r1[q].y}, q0);
        q1 = GN_PK(ai[2 * q + 1], (v2f){r1[q].z, r1[q].w}, q1);
      }
      p0 += p1;
      q0 += q1;
      const v2f dot = {p0.x + p0.y, q0.x + q0.y};
      const v2f e2 = GN_PK(BC(-2.f), dot, BC(ai2) + G2[9]);
      const float ax = __builtin_amdgcn_rcpf(1.f + __expf(fmaxf(e2.x, 0.f)));
      const float ay = __builtin_amdgcn_rcpf(1.f + __expf(fmaxf(e2.y, 0.f)));
      const v2f a = {in0 ? ax : 0.f, in1 ? ay : 0.f};  // sigmoid(-d2), masked
      if (__ballot(a.x > 1e-9f || a.y > 1e-9f) == 0ull) continue;  
      const v2f Yx = GN_PK(BC(c0.x), Xx, GN_PK(BC(c1.x), Xy, GN_PK(BC(c2.x), Xz, BC(Ti.t.x))));
      const v2f Yy = GN_PK(BC(c0.y), Xx, GN_PK(BC(c1.y), Xy, GN_PK(BC(c2.y), Xz, BC(Ti.t.y))));
      const v2f d = {__builtin_amdgcn_rcpf(fmaxf(Yz.x, MIN_DEPTH)), __builtin_amdgcn_rcpf(fmaxf(Yz.y, MIN_DEPTH))};
      const v2f xn = Yx * d, yn = Yy * d;
      const v2f rx = G2[3] - xn, ry = G2[4] - yn, rz = G2[5] - d;  // (rx, ry in units of fx, fy: folded into S00, S11)
      const v2f S00 = a * G2[6], S11 = a * G2[7], t = (a * G2[8]) * d;
      const v2f S02 = -(S00 * xn), S12 = -(S11 * yn);
      const v2f S22 = GN_PK(t, d, -GN_PK(S12, yn, S02 * xn));
      const v2f g0 = S00 * rx, g1 = S11 * ry;
      const v2f g2 = -GN_PK(t, rz, GN_PK(g1, yn, g0 * xn));
      const v2f dd = d * d;
      H00 = GN_PK(dd, S00, H00); H11 = GN_PK(dd, S11, H11); H02 = GN_PK(dd, S02, H02); H12 = GN_PK(dd, S12, H12);
      H22 = GN_PK(dd, S22, H22);
      const v2f N00 = yn * S02, N01 = GN_PK(-xn, S02, S00), N02 = -(yn * S00);
      const v2f N10 = GN_PK(yn, S12, -S11), N11 = -(xn * S12), N12 = xn * S11;
      const v2f N20 = GN_PK(yn, S22, -S12), N21 = GN_PK(-xn, S22, S02), N22 = GN_PK(xn, S12, -N00);
      H03 = GN_PK(d, N00, H03); H04 = GN_PK(d, N01, H04); H05 = GN_PK(d, N02, H05);
      H13 = GN_PK(d, N10, H13); H14 = GN_PK(d, N11, H14); H15 = GN_PK(d, N12, H15);
      H23 = GN_PK(d, N20, H23); H24 = GN_PK(d, N21, H24); H25 = GN_PK(d, N22, H25);
      H33 = GN_PK(yn, N20, H33) - N10; H34 = GN_PK(yn, N21, H34) - N11; H35 = GN_PK(yn, N22, H35) - N12;
      H44 = GN_PK(-xn, N21, H44) + N01; H45 = GN_PK(-xn, N22, H45) + N02;
      H55 = GN_PK(xn, N12, GN_PK(-yn, N02, H55));
      b0 = GN_PK(d, g0, b0); b1 = GN_PK(d, g1, b1); b2 = GN_PK(d, g2, b2);
      b3 = GN_PK(yn, g2, b3) - g1; b4 = GN_PK(-xn, g2, b4) + g0; b5 = GN_PK(xn, g1, GN_PK(-yn, g0, b5));
    }
  }
#undef BC
#define S2(v) ((v).x + (v).y)
  const float Hs[27] = {S2(H00), 0.f, S2(H02), S2(H03), S2(H04), S2(H05), S2(H11), S2(H12), S2(H13), S2(H14), S2(H15),
                        S2(H22), S2(H23), S2(H24), S2(H25), S2(H33), S2(H34), S2(H35), S2(H44), S2(H45), S2(H55),
                        S2(b0), S2(b1), S2(b2), S2(b3), S2(b4), S2(b5)};
#undef S2
  float* pp = part + (((size_t)b * ntiles + tile) * gmax + g) * 27 * 64 + lane;
#pragma unroll
  for (int w_ = 0; w_ < GN_WAVES; ++w_) {  // (one [27][64] image, waves add in index order)
    if (wave == w_) {
#pragma unroll
      for (int k = 0; k < 27; ++k) red[k][lane] = w_ == 0 ? Hs[k] : red[k][lane] + Hs[k];
    }
    __syncthreads();
  }
  for (int k = wave; k < 27; k += GN_WAVES) pp[k * 64] = red[k][lane];
}

// The pair builder with the pixel's own embedding in LDS (round 5, CODD_GN_AILDS=1): same instructions on the same values in
// the same order -- same bits -- but 32 registers fewer live across the step, so that four waves fit a SIMD (the scalar-load
// waits are 37 % of a wave's life: finding 57) without the 16 KB of affinities per workgroup that the two-pass builder costs
// the co-scheduled z|r convolution (8 KB per workgroup here, shared with the reduction image).
__global__ __launch_bounds__(64 * GN_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void se3_gn_build5_kernel(
    const float* __restrict__ T, const float* __restrict__ jd, const float* __restrict__ geo2, int h, int w, float fx,
    float fy, float cx, float cy, int radius, int tiles_x, int ntiles, int q4, int gmax, float* __restrict__ part) {
  // the pixel's own embedding lives in LDS ([quad q][lane] float4: conflict-free 16-byte reads), NOT in 32 registers that
  // are live across the whole step; the partial-sum image of the reduction aliases it (the loop is over by then)
  __shared__ float4 aild[8][64];
  float (*red)[64] = (float (*)[64])&aild[0][0];  // [27][64] floats = 6.9 KB of the 8 KB
  const int N = h * w, wp2 = (w + 1) >> 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int tx0 = (tile % tiles_x) * 8, ty0 = (tile / tiles_x) * 8;
  const int ylo = max(ty0 - radius, 0), yhi = min(ty0 + 7 + radius, h - 1);
  const int xlo = max(tx0 - radius, 0), xhi = min(tx0 + 7 + radius, w - 1);
  const int ncols = xhi - xlo + 1, nj = (yhi - ylo + 1) * ncols;
  const int G = gn_groups(nj, q4, gmax);
  if (g >= G) return;
  const int slot = g * GN_WAVES + wave, nslots = G * GN_WAVES;
  const int s0 = (int)((long long)nj * slot / nslots), s1 = (int)((long long)nj * (slot + 1) / nslots);
  const int ys = ylo + s0 / ncols, xs = xlo + s0 % ncols;
  const int ye = ylo + (s1 - 1) / ncols, xe = xlo + (s1 - 1) % ncols;

  const int xi = tx0 + (lane & 7), yi = ty0 + (lane >> 3);
  const bool vi = xi < w && yi < h;
  const int i = vi ? yi * w + xi : 0;
  const int xim = xi - radius, yim = yi - radius;
  const unsigned twor = 2u * (unsigned)radius;
  const float* rec = jd + (size_t)b * N * GN_JS;
  const float* aip = rec + (size_t)i * GN_JS;
  const SE3T Ti = se3_load(T + ((size_t)b * N + i) * 7);
  const V3 c0 = qrot(Ti.q, V3{1.f, 0.f, 0.f}), c1 = qrot(Ti.q, V3{0.f, 1.f, 0.f}), c2 = qrot(Ti.q, V3{0.f, 0.f, 1.f});
  for (int q = wave; q < 8; q += GN_WAVES) aild[q][lane] = *(const float4*)(aip + 4 * q);  // (every wave of the tile: same pixels)
  __syncthreads();
  const float ai2 = aip[41];
  const v2f Z = {0.f, 0.f};
  v2f H00 = Z, H11 = Z, H02 = Z, H12 = Z, H22 = Z, H03 = Z, H04 = Z, H05 = Z, H13 = Z, H14 = Z, H15 = Z, H23 = Z, H24 = Z,
      H25 = Z, H33 = Z, H34 = Z, H35 = Z, H44 = Z, H45 = Z, H55 = Z, b0 = Z, b1 = Z, b2 = Z, b3 = Z, b4 = Z, b5 = Z;
#define BC(s) ((v2f){(s), (s)})

  // Walk over the wave's pairs, row by row.  The records of a pair occupy 84 SGPRs (2 x 32 embedding values + 10 geometry
  // pairs): five of the six 16-register tuples a wave has, so the compiler requests them in three or four separately
  // awaited portions, and they cannot be requested a step ahead -- every attempt (next pair's records loaded into the
  // registers the step has just finished with; scalar-cache warm-up loads; the embeddings through the wave's own LDS
  // region with broadcast ds_read_b128) ended in SGPR spills through v_readlane / v_writelane, in vector loads, or at 2
  // waves per SIMD and slower than this form (ROCm 7.2; DESIGN finding 45).
  if (s1 > s0)
  for (int yj = ys; yj <= ye; ++yj) {
    const bool rowin = vi && (unsigned)(yj - yim) <= twor;
    const float* rrow = rec + (size_t)yj * w * GN_JS;
    const v2f* grow = (const v2f*)(geo2 + ((size_t)b * h + yj) * wp2 * GN_G2);
    const int xa = yj == ys ? xs : xlo, xb = yj == ye ? xe : xhi;
    for (int xp = xa >> 1; xp <= (xb >> 1); ++xp) {
      const int x0 = 2 * xp, x1 = x0 + 1;
      const bool own0 = x0 >= xa, own1 = x1 <= xb;  // (x0 <= xb and x1 >= xa always hold)
      const float4* e0 = (const float4*)(rrow + (size_t)x0 * GN_JS);
      const float4* e1 = (const float4*)(rrow + (size_t)min(x1, w - 1) * GN_JS);
      const v2f* gp = grow + (size_t)xp * (GN_G2 / 2);
      v2f G2[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) G2[k] = gp[k];
      float4 r0[8], r1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r0[q] = e0[q];
#pragma unroll
      for (int q = 0; q < 8; ++q) r1[q] = e1[q];
      const v2f Xx = G2[0], Xy = G2[1], Xz = G2[2];
      const v2f Yz = GN_PK(BC(c0.z), Xx, GN_PK(BC(c1.z), Xy, GN_PK(BC(c2.z), Xz, BC(Ti.t.z))));
      const bool in0 = rowin & own0 & ((unsigned)(x0 - xim) <= twor) & (Xz.x >= MIN_DEPTH) & (Yz.x >= MIN_DEPTH);
      const bool in1 = rowin & own1 & ((unsigned)(x1 - xim) <= twor) & (Xz.y >= MIN_DEPTH) & (Yz.y >= MIN_DEPTH);
      v2f p0 = Z, p1 = Z, q0 = Z, q1 = Z;
      float4 a4 = aild[0][lane], a5 = aild[1][lane];  // two quads in flight, the next one requested before this one is used
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 nx = aild[q + 2 < 8 ? q + 2 : 7][lane];
        const v2f alo = {a4.x, a4.y}, ahi = {a4.z, a4.w};
        p0 = GN_PK(alo, (v2f){r0[q].x, r0[q].y}, p0);
        p1 = GN_PK(ahi, (v2f){r0[q].z, r0[q].w}, p1);
        q0 = GN_PK(alo, (v2f){r1[q].x, r1[q].y}, q0);
        q1 = GN_PK(ahi, (v2f){r1[q].z, r1[q].w}, q1);
        __builtin_amdgcn_sched_barrier(0);  // (keeps the compiler from hoisting all eight reads: 32 registers again)
        a4 = a5; a5 = nx;
      }
      p0 += p1;
      q0 += q1;
      const v2f dot = {p0.x + p0.y, q0.x + q0.y};
      const v2f e2 = GN_PK(BC(-2.f), dot, BC(ai2) + G2[9]);
      const float ax = __builtin_amdgcn_rcpf(1.f + __expf(fmaxf(e2.x, 0.f)));
      const float ay = __builtin_amdgcn_rcpf(1.f + __expf(fmaxf(e2.y, 0.f)));
      const v2f a = {in0 ? ax : 0.f, in1 ? ay : 0.f};  // sigmoid(-d2), masked
      if (__ballot(a.x > 1e-9f || a.y > 1e-9f) == 0ull) continue;  
      const v2f Yx = GN_PK(BC(c0.x), Xx, GN_PK(BC(c1.x), Xy, GN_PK(BC(c2.x), Xz, BC(Ti.t.x))));
      const v2f Yy = GN_PK(BC(c0.y), Xx, GN_PK(BC(c1.y), Xy, GN_PK(BC(c2.y), Xz, BC(Ti.t.y))));
      const v2f d = {__builtin_amdgcn_rcpf(fmaxf(Yz.x, MIN_DEPTH)), __builtin_amdgcn_rcpf(fmaxf(Yz.y, MIN_DEPTH))};
      const v2f xn = Yx * d, yn = Yy * d;
      const v2f rx = G2[3] - xn, ry = G2[4] - yn, rz = G2[5] - d;  // (rx, ry in units of fx, fy: folded into S00, S11)
      const v2f S00 = a * G2[6], S11 = a * G2[7], t = (a * G2[8]) * d;
      const v2f S02 = -(S00 * xn), S12 = -(S11 * yn);
      const v2f S22 = GN_PK(t, d, -GN_PK(S12, yn, S02 * xn));
      const v2f g0 = S00 * rx, g1 = S11 * ry;
      const v2f g2 = -GN_PK(t, rz, GN_PK(g1, yn, g0 * xn));
      const v2f dd = d * d;
      H00 = GN_PK(dd, S00, H00); H11 = GN_PK(dd, S11, H11); H02 = GN_PK(dd, S02, H02); H12 = GN_PK(dd, S12, H12);
      H22 = GN_PK(dd, S22, H22);
      const v2f N00 = yn * S02, N01 = GN_PK(-xn, S02, S00), N02 = -(yn * S00);
      const v2f N10 = GN_PK(yn, S12, -S11), N11 = -(xn * S12), N12 = xn * S11;
      const v2f N20 = GN_PK(yn, S22, -S12), N21 = GN_PK(-xn, S22, S02), N22 = GN_PK(xn, S12, -N00);
      H03 = GN_PK(d, N00, H03); H04 = GN_PK(d, N01, H04); H05 = GN_PK(d, N02, H05);
      H13 = GN_PK(d, N10, H13); H14 = GN_PK(d, N11, H14); H15 = GN_PK(d, N12, H15);
      H23 = GN_PK(d, N20, H23); H24 = GN_PK(d, N21, H24); H25 = GN_PK(d, N22, H25);
      H33 = GN_PK(yn, N20, H33) - N10; H34 = GN_PK(yn, N21, H34) - N11; H35 = GN_PK(yn, N22, H35) - N12;
      H44 = GN_PK(-xn, N21, H44) + N01; H45 = GN_PK(-xn, N22, H45) + N02;
      H55 = GN_PK(xn, N12, GN_PK(-yn, N02, H55));
      b0 = GN_PK(d, g0, b0); b1 = GN_PK(d, g1, b1); b2 = GN_PK(d, g2, b2);
      b3 = GN_PK(yn, g2, b3) - g1; b4 = GN_PK(-xn, g2, b4) + g0; b5 = GN_PK(xn, g1, GN_PK(-yn, g0, b5));
    }
  }
#undef BC
#define S2(v) ((v).x + (v).y)
  const float Hs[27] = {S2(H00), 0.f, S2(H02), S2(H03), S2(H04), S2(H05), S2(H11), S2(H12), S2(H13), S2(H14), S2(H15),
                        S2(H22), S2(H23), S2(H24), S2(H25), S2(H33), S2(H34), S2(H35), S2(H44), S2(H45), S2(H55),
                        S2(b0), S2(b1), S2(b2), S2(b3), S2(b4), S2(b5)};
#undef S2
  float* pp = part + (((size_t)b * ntiles + tile) * gmax + g) * 27 * 64 + lane;
  __syncthreads();  // every wave is done with the embedding image before the reduction re-uses its LDS
#pragma unroll
  for (int w_ = 0; w_ < GN_WAVES; ++w_) {  // (one [27][64] image, waves add in index order)
    if (wave == w_) {
#pragma unroll
      for (int k = 0; k < 27; ++k) red[k][lane] = w_ == 0 ? Hs[k] : red[k][lane] + Hs[k];
    }
    __syncthreads();
  }
  for (int k = wave; k < 27; k += GN_WAVES) pp[k * 64] = red[k][lane];
}


// Damp, solve (Cholesky, fp64) and retract the 64 pixels of one tile from its summed normal equations sums[27][64]
// (one wave; lane = pixel).
static __device__ __forceinline__ void gn_solve_tile(float* __restrict__ T, const float (*sums)[64], int lane, int b,
                                                     int tx0, int ty0, int h, int w, float lm, float ep) {
  const int xi = tx0 + (lane & 7), yi = ty0 + (lane >> 3);
  if (xi >= w || yi >= h) return;
  const int N = h * w, i = yi * w + xi;
  float Hf[21], bf[6];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float s = sums[k][lane];
    if (k < 21) Hf[k] = s; else bf[k - 21] = s;
  }
  // damping H_pp += lm*H_pp + ep (fp32, as the reference), then Cholesky solve in fp64
  double L[6][6];
  {
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int q = p; q < 6; ++q) {
        float v = Hf[k++];
        if (p == q) v = v + (lm * v + ep);
        L[q][p] = (double)v;  // lower triangle holds A
      }
  }
  bool ok = true;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      double s = L[p][q];
#pragma unroll
      for (int r = 0; r < q; ++r) s -= L[p][r] * L[q][r];
      if (p == q) { if (!(s > 0.0)) { ok = false; s = 1.0; } L[p][p] = sqrt(s); }
      else L[p][q] = s / L[q][q];
    }
  }
  double yv[6], xv[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    double s = (double)bf[p];
#pragma unroll
    for (int r = 0; r < p; ++r) s -= L[p][r] * yv[r];
    yv[p] = s / L[p][p];
  }
#pragma unroll
  for (int p = 5; p >= 0; --p) {
    double s = yv[p];
#pragma unroll
    for (int r = p + 1; r < 6; ++r) s -= L[r][p] * xv[r];
    xv[p] = s / L[p][p];
  }
  if (ok) {
    float* tp = T + ((size_t)b * N + i) * 7;
    const SE3T Ti = se3_load(tp);
    const SE3T dT = se3_exp(V3{(float)xv[0], (float)xv[1], (float)xv[2]}, V3{(float)xv[3], (float)xv[4], (float)xv[5]});
    se3_store(tp, se3_compose(dT, Ti));
  }
}

// One workgroup per tile: 14 waves add the tile's G partials (fixed order -> deterministic), wave 0 damps, solves
// (Cholesky, fp64) and retracts its 64 pixels.  (The separate-launch form: CODD_GN_FUSED_SOLVE=0.)
__global__ __launch_bounds__(64 * GN_SOLVE_WAVES) void se3_gn_solve_kernel(
    float* __restrict__ T, const float* __restrict__ part, int h, int w, int radius, int tiles_x, int ntiles, int q4,
    int gmax, float lm, float ep) {
  __shared__ float sums[27][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int tx0 = (tile % tiles_x) * 8, ty0 = (tile / tiles_x) * 8;
  {
    const int ylo = max(ty0 - radius, 0), yhi = min(ty0 + 7 + radius, h - 1);
    const int xlo = max(tx0 - radius, 0), xhi = min(tx0 + 7 + radius, w - 1);
    const int G = gn_groups((yhi - ylo + 1) * (xhi - xlo + 1), q4, gmax);
    const float* p = part + (((size_t)b * ntiles + tile) * gmax) * 27 * 64 + lane;
    for (int k = wave; k < 27; k += GN_SOLVE_WAVES) {
      float s = 0.f;
      int g = 0;
      for (; g + 4 <= G; g += 4) {  // four loads in flight, added in index order
        const float v0 = p[((size_t)g * 27 + k) * 64], v1 = p[((size_t)(g + 1) * 27 + k) * 64];
        const float v2 = p[((size_t)(g + 2) * 27 + k) * 64], v3 = p[((size_t)(g + 3) * 27 + k) * 64];
        s = (((s + v0) + v1) + v2) + v3;
      }
      for (; g < G; ++g) s += p[((size_t)g * 27 + k) * 64];
      sums[k][lane] = s;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  gn_solve_tile(T, sums, lane, b, tx0, ty0, h, w, lm, ep);
}

// Neighbours per workgroup (4 waves): small enough that the launch is several dispatch rounds of equal-sized
// waves (dynamic balance over the 256 CUs), large enough that a wave's set-up (its 32 + 12 per-pixel registers) and
// the per-workgroup partial (6.9 KB) stay in the noise.  Default 192 (a pair slot is lost at each end of a row segment:
// 103.5 us at 128, 100.3 at 192, 99.4 at 256, 104.7 at 384); CODD_OPT_GN_Q4 selects another grouping -- another fp32
// summation order of the same normal equations, which the parity tests use as a re-association probe.
static inline int gn_q4() {
  const int q = codd_opt(CODD_OPT_GN_Q4);
  return q < 16 ? 16 : q;
}
static inline int gn_gmax(int radius) {
  const int NC = 8 + 2 * radius;
  return gn_groups(NC * NC, gn_q4(), 1 << 20);
}

// floats of scratch in front of the pair builder's geo2 image: [partials | neighbour records | per-tile counters]
static inline size_t gn_geo2_offset(int B, int h, int w, int radius) {
  const int ntiles = cdiv(w, 8) * cdiv(h, 8);
  const size_t n = (size_t)B * ntiles * gn_gmax(radius) * 27 * 64 + (size_t)B * h * w * GN_JS + (size_t)B * ntiles;
  return (n + 3) & ~(size_t)3;
}
extern "C" long long codd_se3_gn_scratch(int B, int h, int w, int radius) {
  // (+ 2 pairs: the pair builder's warm-up loads read one pair past the step's own)
  return (long long)gn_geo2_offset(B, h, w, radius) + ((long long)B * h * ((w + 1) / 2) + 2) * GN_G2;
}
static inline int* gn_counters(float* Hb, int B, int h, int w, int radius) {
  const int ntiles = cdiv(w, 8) * cdiv(h, 8);
  return (int*)(Hb + (size_t)B * ntiles * gn_gmax(radius) * 27 * 64 + (size_t)B * h * w * GN_JS);
}

static int gn_build_solve(float* T, int B, int h, int w, float fx, float fy, float cx, float cy, int radius, float lm,
                          float ep, float* Hb, hipStream_t s) {
  const int tiles_x = cdiv(w, 8), ntiles = tiles_x * cdiv(h, 8);
  const int q4 = gn_q4(), gmax = gn_gmax(radius);
  float* part = Hb;
  const float* jd = Hb + (size_t)B * ntiles * gmax * 27 * 64;
  // CODD_OPT_GN_BUILDER: 5 (default) = the pixel's own embedding in LDS (125 VGPRs), 3 = in registers (157 VGPRs): the same
  // instructions on the same values in the same order (bit-identical; tests/test_gpu_motion_ops.py)
  if (codd_opt(CODD_OPT_GN_BUILDER) == 3)
    se3_gn_build3_kernel<<<dim3(ntiles, gmax, B), 64 * GN_WAVES, 0, s>>>(T, jd, Hb + gn_geo2_offset(B, h, w, radius), h, w, fx, fy,
                                                                        cx, cy, radius, tiles_x, ntiles, q4, gmax, part);
  else
    se3_gn_build5_kernel<<<dim3(ntiles, gmax, B), 64 * GN_WAVES, 0, s>>>(T, jd, Hb + gn_geo2_offset(B, h, w, radius), h, w, fx, fy,
                                                                        cx, cy, radius, tiles_x, ntiles, q4, gmax, part);
  CODD_LAUNCH_CHECK();
  se3_gn_solve_kernel<<<dim3(ntiles, B), 64 * GN_SOLVE_WAVES, 0, s>>>(T, part, h, w, radius, tiles_x, ntiles, q4, gmax, lm, ep);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_se3_gn_step(float* T, const float* ae, int ae_c, const float* xyz, const float* delta,
                                const float* weight, const float* depth1, int B, int h, int w, float fx, float fy,
                                float cx, float cy, int radius, float lm, float ep, float* Hb, void* stream) {
  if (!T || !ae || !xyz || !delta || !weight || !depth1 || !Hb || ae_c < 1 || ae_c > GN_AE || radius < 0)
    return CODD_EINVAL;
  const int ntiles = cdiv(w, 8) * cdiv(h, 8);
  float* jd = Hb + (size_t)B * ntiles * gn_gmax(radius) * 27 * 64;
  hipStream_t s = (hipStream_t)stream;
  se3_gn_prep_kernel<<<dim3(cdiv(h * w, 128), B), 128, 0, s>>>(ae, ae_c, xyz, delta, weight, depth1, h, w, fx, fy, cx,
                                                             cy, jd, gn_counters(Hb, B, h, w, radius), ntiles,
                                                             Hb + gn_geo2_offset(B, h, w, radius));
  CODD_LAUNCH_CHECK();
  return gn_build_solve(T, B, h, w, fx, fy, cx, cy, radius, lm, ep, Hb, s);
}

extern "C" int codd_se3_gn_step_heads(float* T, codd_xs_view hidden, const void* head_w, const float* head_b,
                                      const float* xyz, const float* depth1, int B, int h, int w, float fx, float fy,
                                      float cx, float cy, int radius, float lm, float ep, float* weight_out, float* Hb,
                                      void* stream) {
  if (!T || !head_w || ((uintptr_t)head_w & 15) || !head_b || !xyz || !depth1 || !weight_out || !Hb || radius < 0 ||
      !xs_view_ok(hidden, 768, h, w))
    return CODD_EINVAL;
  const int ntiles = cdiv(w, 8) * cdiv(h, 8);
  float* jd = Hb + (size_t)B * ntiles * gn_gmax(radius) * 27 * 64;
  hipStream_t s = (hipStream_t)stream;
  gn_heads_prep_kernel<<<dim3(cdiv(h * w, 16), B), 128, 0, s>>>(hidden, (const uint4*)head_w, head_b, xyz, depth1, h, w, fx, fy, cx, cy,
                                                               jd, weight_out, gn_counters(Hb, B, h, w, radius), ntiles,
                                                               Hb + gn_geo2_offset(B, h, w, radius));
  CODD_LAUNCH_CHECK();
  return gn_build_solve(T, B, h, w, fx, fy, cx, cy, radius, lm, ep, Hb, s);
}

// ------------------------------------------------------------------------------------------------
// Convex 8x up-sampling (reference se3_field.py:173-192).  Workgroup = one coarse row segment of 64
// pixels; thread = coarse pixel, waves split the 64 sub-pixels (i, j).  mask reads are coalesced
// along x; the 3x3 neighbourhood of the (log-)data lives in registers.
// ------------------------------------------------------------------------------------------------
template <int MODE, int DIM>
__global__ __launch_bounds__(256) void cvx_upsample_kernel(const float* __restrict__ data,
                                                           const float* __restrict__ mask, int h, int w,
                                                           float* __restrict__ out) {
  // workgroup = 64 coarse pixels of one row x a quarter of the 64 sub-pixels (4 per wave): 4x the workgroups of a
  // one-row-segment-per-workgroup grid (144 at 72x120: 39 us for the 19.9 MB mask), the 9 neighbours' data (for the
  // SE3 mode their logarithms) are recomputed per quarter
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int wave = (threadIdx.x >> 6) + 4 * (blockIdx.z & 3);
  const int y = blockIdx.y, b = blockIdx.z >> 2;
  if (x >= w) return;
  const int N = h * w;
  constexpr int D = (MODE == 1) ? 6 : DIM;
  float nb[9][D];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool in = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
    if (MODE == 1) {
      V3 tau = V3{0, 0, 0}, phi = V3{0, 0, 0};
      if (in) { const SE3T Tn = se3_load(data + ((size_t)b * N + yy * w + xx) * 7); se3_log(Tn, &tau, &phi); }
      nb[k][0] = tau.x; nb[k][1] = tau.y; nb[k][2] = tau.z; nb[k][3] = phi.x; nb[k][4] = phi.y; nb[k][5] = phi.z;
    } else if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < D; ++c) nb[k][c] = in ? data[((size_t)b * N + yy * w + xx) * D + c] : 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < D; ++c) nb[k][c] = in ? data[((size_t)b * D + c) * N + yy * w + xx] : 0.f;
    }
  }
  const float* mb = mask + (size_t)b * 576 * N + (size_t)y * w + x;
  const int H8 = 8 * h, W8 = 8 * w;
  for (int s = wave * 4; s < wave * 4 + 4; ++s) {
    float m[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = mb[(size_t)(k * 64 + s) * N]; mx = fmaxf(mx, m[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float wk = m[k] / den;
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] += wk * nb[k][c];
    }
    const int oy = 8 * y + (s >> 3), ox = 8 * x + (s & 7);
    if (MODE == 1) {
      const SE3T To = se3_exp(V3{acc[0], acc[1], acc[2]}, V3{acc[3], acc[4], acc[5]});
      se3_store(out + ((size_t)b * H8 * W8 + (size_t)oy * W8 + ox) * 7, To);
    } else if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < D; ++c) out[((size_t)b * H8 * W8 + (size_t)oy * W8 + ox) * D + c] = acc[c];
    } else {
#pragma unroll
      for (int c = 0; c < D; ++c) out[((size_t)b * D + c) * H8 * W8 + (size_t)oy * W8 + ox] = acc[c];
    }
  }
}

// upsample_se3 (mode 1) and the convex up-sampling of the 3-channel confidence (mode 2) of the SAME mask in one pass
// (reference raft3d.py:267-273 applies both after the last update): the 19.9 MB mask and its soft-max are read / formed once.
// Stores (round 6): a thread owns one COARSE pixel, so its results for sub-pixel (i, j) lie 8 output pixels = 224 bytes
// (SE3 records) / 32 bytes (confidence planes) from its neighbour lane's -- as direct stores every lane wrote its own
// partial 32-byte sector: 60 MB of HBM writes for 22 MB of results (profiles/r05_pmc_counters.md).  The workgroup's results
// (2 output rows x 512 output pixels) are staged in LDS and leave as whole rows: 14 336 contiguous bytes per SE3 row,
// 2 048 per confidence row, 16-byte stores.  Same arithmetic, same bits.
#define CVX_TS 57  // floats per coarse pixel of a staged SE3 row: 8 sub-pixels x 7 + 1 pad (odd stride: <= 2 lanes per bank)
#define CVX_WS 9   // floats per coarse pixel of a staged confidence row: 8 + 1 pad
__global__ __launch_bounds__(256) void cvx_upsample_se3w_kernel(const float* __restrict__ T, const float* __restrict__ wgt,
                                                                const float* __restrict__ mask, int h, int w,
                                                                float* __restrict__ Tout, float* __restrict__ wout) {
  __shared__ float sT[2][64 * CVX_TS];
  __shared__ float sW[3][2][64 * CVX_WS];
  const int xl = threadIdx.x & 63, x0 = blockIdx.x * 64, x = x0 + xl;
  const int q = blockIdx.z & 3;  // quarter of the 64 sub-pixels: output rows 8 y + 2 q, + 1
  const int wave = (threadIdx.x >> 6) + 4 * q;
  const int y = blockIdx.y, b = blockIdx.z >> 2;
  const int N = h * w;
  const int H8 = 8 * h, W8 = 8 * w;
  if (x < w) {
    float nb[9][9];  // 6 twist components | 3 confidence channels of the 9 neighbours
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      const bool in = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
      V3 tau = V3{0, 0, 0}, phi = V3{0, 0, 0};
      if (in) { const SE3T Tn = se3_load(T + ((size_t)b * N + yy * w + xx) * 7); se3_log(Tn, &tau, &phi); }
      nb[k][0] = tau.x; nb[k][1] = tau.y; nb[k][2] = tau.z; nb[k][3] = phi.x; nb[k][4] = phi.y; nb[k][5] = phi.z;
#pragma unroll
      for (int c = 0; c < 3; ++c) nb[k][6 + c] = in ? wgt[((size_t)b * 3 + c) * N + yy * w + xx] : 0.f;
    }
    const float* mb = mask + (size_t)b * 576 * N + (size_t)y * w + x;
    for (int s = wave * 4; s < wave * 4 + 4; ++s) {
      float m[9], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; ++k) { m[k] = mb[(size_t)(k * 64 + s) * N]; mx = fmaxf(mx, m[k]); }
      float den = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
      float acc[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) acc[c] = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float wk = m[k] / den;
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] += wk * nb[k][c];
      }
      const int row = (s >> 3) - 2 * q, j = s & 7;
      const SE3T To = se3_exp(V3{acc[0], acc[1], acc[2]}, V3{acc[3], acc[4], acc[5]});
      se3_store(&sT[row][xl * CVX_TS + j * 7], To);
#pragma unroll
      for (int c = 0; c < 3; ++c) sW[c][row][xl * CVX_WS + j] = acc[6 + c];
    }
  }
  __syncthreads();
  const int nv = min(64, w - x0);  // coarse pixels of this segment
  const int oy0 = 8 * y + 2 * q;
  // SE3 records: a row segment is nv * 56 contiguous floats (16-byte aligned: 224 bytes per coarse pixel)
  for (int i = threadIdx.x; i < 2 * nv * 14; i += 256) {
    const int row = i / (nv * 14), q4 = i - row * (nv * 14), p = q4 / 14, k = q4 - p * 14;
    const float* src = &sT[row][p * CVX_TS + 4 * k];
    *(f32x4*)(Tout + ((size_t)b * H8 * W8 + (size_t)(oy0 + row) * W8 + 8 * x0) * 7 + 4 * q4) = f32x4{src[0], src[1], src[2], src[3]};
  }
  // confidence planes: nv * 8 contiguous floats per (channel, row)
  for (int i = threadIdx.x; i < 6 * nv * 2; i += 256) {
    const int cr = i / (nv * 2), q4 = i - cr * (nv * 2), c = cr >> 1, row = cr & 1, p = q4 >> 1, k = q4 & 1;
    const float* src = &sW[c][row][p * CVX_WS + 4 * k];
    *(f32x4*)(wout + ((size_t)b * 3 + c) * H8 * W8 + (size_t)(oy0 + row) * W8 + 8 * x0 + 4 * q4) = f32x4{src[0], src[1], src[2], src[3]};
  }
}

extern "C" int codd_cvx_upsample_se3_weight(const float* T, const float* weight, const float* mask, int B, int h, int w,
                                            float* T_out, float* weight_out, void* stream) {
  if (!T || !weight || !mask || !T_out || !weight_out || B < 1 || h < 1 || w < 1) return CODD_EINVAL;
  cvx_upsample_se3w_kernel<<<dim3(cdiv(w, 64), h, 4 * B), 256, 0, (hipStream_t)stream>>>(T, weight, mask, h, w, T_out,
                                                                                         weight_out);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_cvx_upsample(const float* data, const float* mask, int B, int h, int w, int dim, int mode,
                                 float* out, void* stream) {
  if (!data || !mask || !out) return CODD_EINVAL;
  dim3 grid(cdiv(w, 64), h, 4 * B);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) cvx_upsample_kernel<1, 6><<<grid, 256, 0, s>>>(data, mask, h, w, out);
  else if (mode == 0 && dim == 6) cvx_upsample_kernel<0, 6><<<grid, 256, 0, s>>>(data, mask, h, w, out);
  else if (mode == 0 && dim == 3) cvx_upsample_kernel<0, 3><<<grid, 256, 0, s>>>(data, mask, h, w, out);
  else if (mode == 0 && dim == 2) cvx_upsample_kernel<0, 2><<<grid, 256, 0, s>>>(data, mask, h, w, out);
  else if (mode == 2 && dim == 3) cvx_upsample_kernel<2, 3><<<grid, 256, 0, s>>>(data, mask, h, w, out);
  else return CODD_EUNSUPPORTED;
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// disparity -> depth (reference motion.py:154-165)
// ------------------------------------------------------------------------------------------------
__global__ void disp_to_depth_kernel(const float* __restrict__ disp, long long n, float bf, float* __restrict__ depth) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float v = bf / (disp[e] + 1e-5f);
  depth[e] = fminf(fmaxf(v, 0.f), 210.f);
}

extern "C" int codd_disp_to_depth(const float* disp, long long n, float bf, float* depth, void* stream) {
  if (!disp || !depth || n < 1) return CODD_EINVAL;
  disp_to_depth_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(disp, n, bf, depth);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// out[b][y][x] = in[b][oy + step y][ox + step x]: the 1/8-resolution depth samples of RAFT3D (`depth[:, 3::8, 3::8]`,
// reference raft3d.py:213-216) as a kernel of this library (a strided torch copy inside the captured frame otherwise)
__global__ void subsample_kernel(const float* __restrict__ in, int H, int W, int oy, int ox, int step, int h, int w,
                                 float* __restrict__ out, long long n) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int x = (int)(e % w);
  const long long t = e / w;
  const int y = (int)(t % h);
  const long long b = t / h;
  out[e] = in[(b * H + oy + (long long)step * y) * W + ox + step * x];
}

extern "C" int codd_subsample(const float* in, int B, int H, int W, int oy, int ox, int step, float* out, void* stream) {
  if (!in || !out || B < 1 || H < 1 || W < 1 || step < 1 || oy < 0 || ox < 0 || oy >= H || ox >= W) return CODD_EINVAL;
  const int h = (H - oy + step - 1) / step, w = (W - ox + step - 1) / step;
  const long long n = (long long)B * h * w;
  subsample_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, H, W, oy, ox, step, h, w, out, n);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Forward splat (Motion.transform_and_project, reference motion.py:82-130).
//   count   (per source point): project, store (u, v, z); count the point on every covered pixel;
//   reserve (per pixel): claim exactly cnt[pixel] list slots from a global cursor;
//   fill    (per source point): write the point id into its pixels' lists;
//   gather  (per output pixel): read ALL of the pixel's candidates back, keep the 8 nearest by (z, id) and
//           composite front to back.
// The lists are exact-size (a point covers at most (2*ceil(R)+1)^2 pixel centres, which bounds the buffer), so no
// candidate is ever dropped however many points pile up on a pixel, and the result does not depend on the order in
// which the atomics resolve: the slot ORDER inside a list and the list's POSITION in the buffer are arbitrary, the
// (z, id) selection in pass 4 is not.
// Pixel centres sit at +0.5 (pytorch3d NDC convention); R = radius * min(H,W) / (2H) pixels.
// ------------------------------------------------------------------------------------------------
struct SplatP {
  const float* T; const float* depth; int HT, WT, oy, ox, ds;
  const float* featA; int CA; const float* featB; int CB; int with_flow;
  int H, W; float fx, fy, cx, cy, R; float bf;
  float* out; float* zout;
  int* cnt; int* off; int* cur; int* cursor; int* list;
  float4* uvz;  // per source point: projected (u, v, z, valid) written by the count pass
};

__device__ __forceinline__ bool splat_point(const SplatP& p, int b, int n, float* u, float* v, float* z,
                                            V3* flow) {
  const int py = n / p.W, px = n - py * p.W;
  const size_t src = (size_t)b * p.HT * p.WT + (size_t)(p.oy + p.ds * py) * p.WT + (p.ox + p.ds * px);
  const SE3T Ti = se3_load(p.T + src * 7);
  const V3 X0 = inv_project(p.depth[src], px, py, p.fx, p.fy, p.cx, p.cy);
  const V3 X1 = se3_act(Ti, X0);
  *z = X1.z;
  if (!(X1.z > 0.f)) return false;
  *u = p.fx * X1.x / X1.z + p.cx;
  *v = p.fy * X1.y / X1.z + p.cy;
  if (flow) {
    const V3 a = project(X1, p.fx, p.fy, p.cx, p.cy), c = project(X0, p.fx, p.fy, p.cx, p.cy);
    *flow = V3{a.x - c.x, a.y - c.y, a.z - c.z};
  }
  return fabsf(*u) < 1e7f && fabsf(*v) < 1e7f;  // also rejects NaN / inf
}

// visits the pixels covered by the point (u, v): f(global pixel index)
template <typename F>
__device__ __forceinline__ void splat_cover(const SplatP& p, int b, float u, float v, F f) {
  const int span = (int)(p.R + 1.5f);
  const int bx = (int)floorf(u - 0.5f), by = (int)floorf(v - 0.5f);
  const float R2 = p.R * p.R;
  for (int oy = -span + 1; oy <= span; ++oy) {
    const int yy = by + oy;
    if ((unsigned)yy >= (unsigned)p.H) continue;
    for (int ox = -span + 1; ox <= span; ++ox) {
      const int xx = bx + ox;
      if ((unsigned)xx >= (unsigned)p.W) continue;
      const float du = u - ((float)xx + 0.5f), dv = v - ((float)yy + 0.5f);
      if (!(du * du + dv * dv < R2)) continue;
      f((size_t)b * p.H * p.W + (size_t)yy * p.W + xx);
    }
  }
}

__global__ void splat_count_kernel(const SplatP p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= p.H * p.W) return;
  float u = 0.f, v = 0.f, z = 0.f;
  const bool ok = splat_point(p, b, n, &u, &v, &z, nullptr);
  p.uvz[(size_t)b * p.H * p.W + n] = make_float4(u, v, z, ok ? 1.f : 0.f);
  if (!ok) return;
  splat_cover(p, b, u, v, [&](size_t pix) { atomicAdd(&p.cnt[pix], 1); });
}

__global__ __launch_bounds__(1024) void splat_reserve_kernel(const SplatP p, long long npix) {
  // one atomic per 4096 pixels: requests to the single cursor serialise in L2 at ~6 ns each (one per thread: 56 us
  // at 960x576, one per wave: 55 us) -- so a workgroup scans 4 counts per thread (shuffles inside a wave, LDS across
  // its 16 waves) and thread 0 claims the workgroup's total
  __shared__ int wsum[16];
  __shared__ int sbase;
  const long long e0 = ((long long)blockIdx.x * 1024 + threadIdx.x) * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = e0 + k < npix ? p.cnt[e0 + k] : 0;
  const int t = (c[0] + c[1]) + (c[2] + c[3]);
  int incl = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int q = 0; q < 16; ++q) {
      const int v = wsum[q];
      wsum[q] = tot;
      tot += v;
    }
    sbase = tot > 0 ? atomicAdd(p.cursor, tot) : 0;
  }
  __syncthreads();
  int o = sbase + wsum[wave] + incl - t;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (e0 + k < npix) {
      p.off[e0 + k] = o;
      p.cur[e0 + k] = 0;
      o += c[k];
    }
}

__global__ void splat_fill_kernel(const SplatP p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= p.H * p.W) return;
  const float4 q = p.uvz[(size_t)b * p.H * p.W + n];
  if (q.w == 0.f) return;
  splat_cover(p, b, q.x, q.y, [&](size_t pix) { p.list[p.off[pix] + atomicAdd(&p.cur[pix], 1)] = n; });
}

__global__ void splat_gather_kernel(const SplatP p) {
  // XCD-contiguous walk (common.h codd_xcd_item): a 256-pixel row segment's candidates are the points that landed within
  // R of it -- the same uvz records, pose / depth and feature lines the segments of the rows above and below read.
  // With the dispatcher's block -> XCD b % 8 placement those neighbours sit on eight different L2s (round 4: 75 MB per
  // launch against 22 MB algorithmic); every XCD now walks one contiguous band of rows.
  const int pix = codd_xcd_item(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  const int HW = p.H * p.W;
  if (pix >= HW) return;
  const int py = pix / p.W, px = pix - py * p.W;
  const size_t gp = (size_t)b * HW + pix;
  const int cnt = p.cnt[gp];
  const int* lp = p.list + p.off[gp];
  // sorted (z, id) top-8 kept in REGISTERS: every array index below is a compile-time constant after unrolling
  // (a run-time index would push the arrays to scratch memory); empty slots hold (+inf, INT_MAX)
  float kz[8], ka[8];
  int kid[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { kz[q] = INFINITY; ka[q] = 0.f; kid[q] = 0x7fffffff; }
  int nk = 0;
  const float R2 = p.R * p.R;
  for (int s = 0; s < cnt; ++s) {
    const int n = lp[s];
    // 16 bytes written by pass 1 instead of re-loading T (28 B) + depth and redoing the SE3 action per candidate
    const float4 q4 = p.uvz[(size_t)b * HW + n];
    const float du = q4.x - ((float)px + 0.5f), dv = q4.y - ((float)py + 0.5f);
    float cz = q4.z, ca = 1.f - (du * du + dv * dv) / R2;
    int cn = n;
    // insertion by carrying the displaced element down the list
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (cz < kz[q] || (cz == kz[q] && cn < kid[q])) {
        const float tz = kz[q], ta = ka[q]; const int tn = kid[q];
        kz[q] = cz; ka[q] = ca; kid[q] = cn;
        cz = tz; ca = ta; cn = tn;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) nk += kid[q] != 0x7fffffff ? 1 : 0;  // occupied slots (they are a prefix of the list)
  const int C = p.CA + (p.with_flow ? 3 : 0) + p.CB;
  float wk[8];
  float tr = 1.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { wk[k] = k < nk ? tr * ka[k] : 0.f; tr *= (1.f - ka[k]); if (k >= nk) kid[k] = 0; }
  float* op = p.out + (size_t)b * C * HW + pix;
  for (int c = 0; c < p.CA; ++c) {
    const float* fp = p.featA + ((size_t)b * p.CA + c) * HW;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nk) acc += wk[k] * fp[kid[k]];
    op[(size_t)c * HW] = acc;
  }
  int co = p.CA;
  if (p.with_flow) {
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nk) {
        float u, v, z; V3 fl;
        splat_point(p, b, kid[k], &u, &v, &z, &fl);
        f0 += wk[k] * fl.x; f1 += wk[k] * fl.y; f2 += wk[k] * fl.z;
      }
    }
    op[(size_t)co * HW] = f0; op[(size_t)(co + 1) * HW] = f1; op[(size_t)(co + 2) * HW] = f2;
    co += 3;
  }
  for (int c = 0; c < p.CB; ++c) {
    const float* fp = p.featB + ((size_t)b * p.CB + c) * HW;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nk) acc += wk[k] * fp[kid[k]];
    op[(size_t)(co + c) * HW] = acc;
  }
  if (p.zout) {
    const float zn = nk > 0 ? fmaxf(kz[0], 0.f) : 0.f;
    float o = zn;
    if (p.bf > 0.f) { o = p.bf / (zn + 1e-5f); if (o > (float)p.W) o = 0.f; }
    p.zout[gp] = o;
  }
}

__global__ void zero_int_kernel(int* p, long long n) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = 0;
}

static inline float splat_radius_px(float radius, int H, int W) { return radius * (float)(H < W ? H : W) / (2.f * (float)H); }
static inline long long splat_cover_bound(float R) {  // pixel centres inside an open disc of radius R: <= (2 ceil(R) + 1)^2
  const long long s = 2 * (long long)ceilf(R) + 1;
  return s * s;
}

/* ints of `scratch` codd_splat needs: [cnt | off | cur] (3 B H W) + cursor (4, padding) + candidate lists
 * (B H W * cover bound) + one float4 (u, v, z, valid) per source point */
extern "C" long long codd_splat_scratch(int B, int H, int W, float radius) {
  if (B < 1 || H < 1 || W < 1 || !(radius > 0.f)) return -1;
  const long long n = (long long)B * H * W;
  return ((3 * n + 4 + 3) & ~3LL) + ((n * splat_cover_bound(splat_radius_px(radius, H, W)) + 3) & ~3LL) + 4 * n;
}

extern "C" int codd_splat(const float* T, const float* depth, int HT, int WT, int oy, int ox, int ds,
                          const float* featA, int CA, const float* featB, int CB, int with_flow, int B, int H, int W,
                          float fx, float fy, float cx, float cy, float radius, float bf, float* out, float* zout,
                          int* scratch, void* stream) {
  if (!T || !depth || !out || !scratch || ((uintptr_t)scratch & 15) || !(radius > 0.f) || CA < 0 || CB < 0 ||
      (CA > 0 && !featA) || (CB > 0 && !featB))
    return CODD_EINVAL;
  if (oy + ds * (H - 1) >= HT || ox + ds * (W - 1) >= WT) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SplatP p;
  p.T = T; p.depth = depth; p.HT = HT; p.WT = WT; p.oy = oy; p.ox = ox; p.ds = ds;
  p.featA = featA; p.CA = CA; p.featB = featB; p.CB = CB; p.with_flow = with_flow;
  p.H = H; p.W = W; p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy;
  p.R = splat_radius_px(radius, H, W);
  p.bf = bf; p.out = out; p.zout = zout;
  const long long n = (long long)B * H * W;
  const long long nlist = (n * splat_cover_bound(p.R) + 3) & ~3LL;
  if (nlist > 0x7fffffffLL) return CODD_EUNSUPPORTED;  // list offsets are 32-bit
  const long long nhead = (3 * n + 4 + 3) & ~3LL;  // counters + cursor, padded to 16 bytes
  p.cnt = scratch; p.off = scratch + n; p.cur = scratch + 2 * n; p.cursor = scratch + 3 * n;
  p.list = scratch + nhead;
  p.uvz = (float4*)(scratch + nhead + nlist);  // 16-byte aligned: scratch is, nhead and nlist are multiples of 4 ints
  // zero the counters + cursor with a kernel (a plain kernel node under graph capture; hipMemsetAsync becomes a
  // memset node whose ordering against neighbouring kernel nodes is not relied upon)
  zero_int_kernel<<<cdiv(3 * n + 4, 256), 256, 0, s>>>(scratch, 3 * n + 4);
  CODD_LAUNCH_CHECK();
  dim3 grid(cdiv(H * W, 256), B);
  splat_count_kernel<<<grid, 256, 0, s>>>(p);
  CODD_LAUNCH_CHECK();
  splat_reserve_kernel<<<cdiv(n, 4096), 1024, 0, s>>>(p, n);
  CODD_LAUNCH_CHECK();
  splat_fill_kernel<<<grid, 256, 0, s>>>(p);
  CODD_LAUNCH_CHECK();
  splat_gather_kernel<<<grid, 256, 0, s>>>(p);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// induced flow (reference projective_ops.py:55-68), full resolution: out [B,H,W,3]
__global__ void induced_flow_kernel(const float* __restrict__ T, const float* __restrict__ depth, int H, int W,
                                    float fx, float fy, float cx, float cy, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (n >= H * W) return;
  const int y = n / W, x = n - y * W;
  const size_t g = (size_t)b * H * W + n;
  const V3 X0 = inv_project(depth[g], x, y, fx, fy, cx, cy);
  const V3 X1 = se3_act(se3_load(T + g * 7), X0);
  const V3 a = project(X1, fx, fy, cx, cy), c = project(X0, fx, fy, cx, cy);
  out[g * 3] = a.x - c.x; out[g * 3 + 1] = a.y - c.y; out[g * 3 + 2] = a.z - c.z;
}

extern "C" int codd_induced_flow(const float* T, const float* depth, int B, int H, int W, float fx, float fy, float cx,
                                 float cy, float* out, void* stream) {
  if (!T || !depth || !out) return CODD_EINVAL;
  dim3 grid(cdiv(H * W, 256), B);
  induced_flow_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(T, depth, H, W, fx, fy, cx, cy, out);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// ConvGRU gate fusions (reference blocks/gru.py:17-34).  The six gate convolutions run as
// independent launches (concurrently, on forked streams); these two kernels apply
//   z|r = sigmoid(conv1 + conv2 + (inp + cor + mot)[0:256]),  rh = r * h
//   q   = tanh   (conv1 + conv2 + (inp + cor + mot)[256:384]), h' = (1 - z) h + z q
// with the three input streams inp / cor / mot [B,384,hw] summed in the reference's order (cor = mot = NULL: ``inp``
// already holds their sum -- the merged 1x1 encoder head of BasicUpdateBlock.run writes inp + cor + mot).
// ------------------------------------------------------------------------------------------------
__global__ void gru_gate_zr_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                   const float* __restrict__ inp, const float* __restrict__ cor,
                                   const float* __restrict__ mot, const float* __restrict__ h, int hw,
                                   float* __restrict__ zr, float* __restrict__ rh, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;  // total = B*256*hw
  const long long b = e / (256LL * hw), r = e - b * 256LL * hw;
  const long long i3 = b * 384LL * hw + r;
  const float isum = cor ? (inp[i3] + cor[i3]) + mot[i3] : inp[i3];
  const float v = 1.f / (1.f + expf(-((t1[e] + t2[e]) + isum)));
  zr[e] = v;
  if (r >= 128LL * hw) { const long long hi = b * 128LL * hw + (r - 128LL * hw); rh[hi] = v * h[hi]; }
}
__global__ void gru_gate_q_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                  const float* __restrict__ inp, const float* __restrict__ cor,
                                  const float* __restrict__ mot, const float* __restrict__ zr,
                                  const float* __restrict__ h, int hw, float* __restrict__ ho, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;  // total = B*128*hw
  const long long b = e / (128LL * hw), r = e - b * 128LL * hw;
  const long long i3 = b * 384LL * hw + 256LL * hw + r;
  const float isum = cor ? (inp[i3] + cor[i3]) + mot[i3] : inp[i3];
  const float q = tanhf((t1[e] + t2[e]) + isum);
  const float z = zr[b * 256LL * hw + r];
  ho[e] = (1.f - z) * h[e] + z * q;
}
extern "C" int codd_gru_gate_zr(const float* t1, const float* t2, const float* inp, const float* cor,
                                const float* mot, const float* h, int B, int hw, float* zr, float* rh, void* stream) {
  if (!t1 || !t2 || !inp || (!cor != !mot) || !h || !zr || !rh) return CODD_EINVAL;
  const long long total = (long long)B * 256 * hw;
  gru_gate_zr_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(t1, t2, inp, cor, mot, h, hw, zr, rh, total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
extern "C" int codd_gru_gate_q(const float* t1, const float* t2, const float* inp, const float* cor, const float* mot,
                               const float* zr, const float* h, int B, int hw, float* hout, void* stream) {
  if (!t1 || !t2 || !inp || (!cor != !mot) || !zr || !h || !hout) return CODD_EINVAL;
  const long long total = (long long)B * 128 * hw;
  gru_gate_q_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(t1, t2, inp, cor, mot, zr, h, hw, hout, total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// The gates writing split-bf16 records: one thread = 8 consecutive channels of one pixel (16-byte record stores;
// the fp32 loads of a wave are 8 coalesced rows of 64 pixels).  Arithmetic identical to the kernels above.
__global__ void gru_gate_zr_xs_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                      const float* __restrict__ inp, const float* __restrict__ cor,
                                      const float* __restrict__ mot, const float* __restrict__ h, int B, int H, int W,
                                      float* __restrict__ z, const codd_xs_view rh) {
  const int hw = H * W;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * 16 * hw) return;
  const int pix = (int)(e % hw), oct = (int)((e / hw) % 16), b = (int)(e / (16LL * hw));
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = oct * 8 + i;
    const size_t ez = ((size_t)b * 256 + c) * hw + pix, er = ez + (size_t)128 * hw;
    const size_t iz = ((size_t)b * 384 + c) * hw + pix, ir = iz + (size_t)128 * hw;
    const size_t ih = ((size_t)b * 128 + c) * hw + pix;
    z[ih] = 1.f / (1.f + expf(-((t1[ez] + t2[ez]) + (cor ? (inp[iz] + cor[iz]) + mot[iz] : inp[iz]))));
    const float r = 1.f / (1.f + expf(-((t1[er] + t2[er]) + (cor ? (inp[ir] + cor[ir]) + mot[ir] : inp[ir]))));
    v[i] = r * h[ih];
  }
  xs_store8(rh, b, oct, pix / W, pix % W, v);
}
__global__ void gru_gate_q_xs_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                     const float* __restrict__ inp, const float* __restrict__ cor,
                                     const float* __restrict__ mot, const float* __restrict__ z,
                                     const float* __restrict__ h, int B, int H, int W, float* __restrict__ ho,
                                     const codd_xs_view hx) {
  const int hw = H * W;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * 16 * hw) return;
  const int pix = (int)(e % hw), oct = (int)((e / hw) % 16), b = (int)(e / (16LL * hw));
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = oct * 8 + i;
    const size_t ih = ((size_t)b * 128 + c) * hw + pix;
    const size_t i3 = ((size_t)b * 384 + 256 + c) * hw + pix;
    const float q = tanhf((t1[ih] + t2[ih]) + (cor ? (inp[i3] + cor[i3]) + mot[i3] : inp[i3]));
    const float zz = z[ih];
    v[i] = (1.f - zz) * h[ih] + zz * q;
    ho[ih] = v[i];
  }
  xs_store8(hx, b, oct, pix / W, pix % W, v);
}
extern "C" int codd_gru_gate_zr_xs(const float* t1, const float* t2, const float* inp, const float* cor,
                                   const float* mot, const float* h, int B, int H, int W, float* z, codd_xs_view rh_xs,
                                   void* stream) {
  if (!t1 || !t2 || !inp || (!cor != !mot) || !h || !z || !xs_view_ok(rh_xs, 128, H, W)) return CODD_EINVAL;
  const long long total = (long long)B * 16 * H * W;
  gru_gate_zr_xs_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(t1, t2, inp, cor, mot, h, B, H, W, z, rh_xs);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
extern "C" int codd_gru_gate_q_xs(const float* t1, const float* t2, const float* inp, const float* cor, const float* mot,
                                  const float* z, const float* h, int B, int H, int W, float* hout, codd_xs_view h_xs,
                                  void* stream) {
  if (!t1 || !t2 || !inp || (!cor != !mot) || !z || !h || !hout || !xs_view_ok(h_xs, 128, H, W)) return CODD_EINVAL;
  const long long total = (long long)B * 16 * H * W;
  gru_gate_q_xs_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(t1, t2, inp, cor, mot, z, h, B, H, W, hout, h_xs);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// net = tanh(x[:, :128]), inp = relu(x[:, 128:512])  (reference raft3d.py:183-186)
__global__ void context_split_kernel(const float* __restrict__ x, int hw, float* __restrict__ net,
                                     float* __restrict__ inp, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long b = e / (512LL * hw), r = e - b * 512LL * hw;
  const float v = x[e];
  if (r < 128LL * hw) net[b * 128LL * hw + r] = tanhf(v);
  else inp[b * 384LL * hw + (r - 128LL * hw)] = fmaxf(v, 0.f);
}
extern "C" int codd_context_split(const float* x, int B, int hw, float* net, float* inp, void* stream) {
  if (!x || !net || !inp) return CODD_EINVAL;
  const long long total = (long long)B * 512 * hw;
  context_split_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x, hw, net, inp, total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// SE3 identity field
__global__ void se3_identity_kernel(float* T, long long n) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  T[e] = (e % 7 == 6) ? 1.f : 0.f;
}
extern "C" int codd_se3_identity(float* T, long long npix, void* stream) {
  if (!T) return CODD_EINVAL;
  se3_identity_kernel<<<cdiv(npix * 7, 256), 256, 0, (hipStream_t)stream>>>(T, npix * 7);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
