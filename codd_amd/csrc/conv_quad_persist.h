#pragma once
// Persistent variant of the quad-layout convolution kernel (conv_quad_kernel.h) for layers whose WHOLE weight tensor is
// one LDS chunk and one channel group (cin <= 32, cout <= 16 * MB: HITNet's 16- / 32-channel layers at 1/2 and full
// resolution, 40 % of the stereo network's time).  conv_quad_kernel re-stages the weights (36.9 KB at 32 -> 32 3x3) for
// every 64-256-pixel tile next to a 8-28 KB input tile and exposes one global-load latency per workgroup; here a
// workgroup
//   * loads the weights ONCE,
//   * walks tiles  t = blockIdx.x, + gridDim.x, ...  (grid = the resident workgroups of the chip),
//   * fetches tile t + 1's input into registers while the MFMAs of tile t run and the epilogue of tile t stores
//     (issue early / commit late across TILES instead of across chunks).
// Same arithmetic, same operand layouts, same epilogue as conv_quad_kernel (bit-identical results).
#include "conv_quad_kernel.h"

template <int NPB, int MB, int WREG, int QREG>
__global__ __launch_bounds__(256) void conv_quad_persist_kernel(const ConvK k, const int ntiles) {
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* il = smem + k.wchunk;
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  constexpr int XB = NPB >= 2 ? 2 : 1;
  constexpr int RPW = NPB / XB;
  const int hwin = p.Hin * p.Win, hwout = p.Hout * p.Wout;
  const int tiles_per_img = k.tiles_x * k.tiles_y;

  // per-thread quad units: (channel quad, row, float4 column) -- independent of the tile
  const int nq = (p.ck >> 2) * k.upc;
  int q_lds[QREG], q_y[QREG], q_x[QREG], q_c[QREG];
#pragma unroll
  for (int r = 0; r < QREG; ++r) {
    const int u = tid + r * NT;
    q_c[r] = -1; q_lds[r] = 0; q_y[r] = 0; q_x[r] = 0;
    if (u < nq) {
      const int cq = u / k.upc, rem = u - cq * k.upc;
      const int y = rem / k.twp4, x4 = rem - y * k.twp4;
      q_c[r] = 4 * cq; q_y[r] = y; q_x[r] = 4 * x4;
      q_lds[r] = ((cq * k.thi + y) * k.twp + 4 * x4) * 4;
    }
  }
  float4 ireg[QREG][4];
#define QP_ISSUE(T)                                                                                        \
  {                                                                                                        \
    const int b_ = (T) / tiles_per_img, tt_ = (T) - b_ * tiles_per_img;                                    \
    const int ty_ = tt_ / k.tiles_x, tx_ = tt_ - ty_ * k.tiles_x;                                          \
    const int gy0_ = ty_ * k.th * p.sy - p.pad_t, gxs_ = tx_ * k.tw * p.sx - p.pad_l - k.xoff;             \
    _Pragma("unroll") for (int r = 0; r < QREG; ++r) {                                                     \
      const int gy = gy0_ + q_y[r], gx = gxs_ + q_x[r];                                                    \
      const bool in_ = q_c[r] >= 0 && (unsigned)gy < (unsigned)p.Hin && gx >= 0 && gx + 3 < p.Win;         \
      const int go_ = gy * p.Win + gx;                                                                     \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
        const int cg = q_c[r] + c;                                                                         \
        if (in_ && cg < k.cin) {                                                                           \
          const float* s_ =                                                                                \
              (cg < p.C0 ? view_ptr(p.in0, b_, cg, hwin) : view_ptr(p.in1, b_, cg - p.C0, hwin)) + go_;    \
          v = *(const float4*)s_;                                                                          \
        }                                                                                                  \
        ireg[r][c] = v;                                                                                    \
      }                                                                                                    \
    }                                                                                                      \
  }

  int pbase[NPB];
#pragma unroll
  for (int a = 0; a < NPB; ++a) {
    const int prow = wave * RPW + a / XB, pcol = (a % XB) * 16 + j;
    pbase[a] = prow * p.sy * k.twp + pcol * p.sx + k.xoff;
  }
  const int wq = 16 * MB, iq = k.thi * k.twp, ngr = p.ck >> 4;

  int t = blockIdx.x;
  if (t >= ntiles) return;
  {  // the layer's weights: once per workgroup
    const float4* src = (const float4*)p.wpacked;
    float4* dst = (float4*)wl;
    const int wchunk4 = k.wchunk >> 2;
    float4 wreg[WREG];
#pragma unroll
    for (int r = 0; r < WREG; ++r) {
      const int e = tid + r * NT;
      wreg[r] = src[e < wchunk4 ? e : wchunk4 - 1];
    }
    QP_ISSUE(t);
#pragma unroll
    for (int r = 0; r < WREG; ++r) {
      const int e = tid + r * NT;
      if (e < wchunk4) dst[e] = wreg[r];
    }
  }
  while (true) {
    __syncthreads();  // every wave is done reading the previous tile's input image
#pragma unroll
    for (int r = 0; r < QREG; ++r)
      if (q_c[r] >= 0) {
        float4* d_ = (float4*)(il + q_lds[r]);  // 4 pixels x (4 channels innermost): register transpose
        d_[0] = make_float4(ireg[r][0].x, ireg[r][1].x, ireg[r][2].x, ireg[r][3].x);
        d_[1] = make_float4(ireg[r][0].y, ireg[r][1].y, ireg[r][2].y, ireg[r][3].y);
        d_[2] = make_float4(ireg[r][0].z, ireg[r][1].z, ireg[r][2].z, ireg[r][3].z);
        d_[3] = make_float4(ireg[r][0].w, ireg[r][1].w, ireg[r][2].w, ireg[r][3].w);
      }
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < ntiles) QP_ISSUE(tn);  // next tile's input: in flight during this tile's MFMAs and stores

    f32x4 acc[NPB][MB];
#pragma unroll
    for (int a = 0; a < NPB; ++a)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4* w4 = (const float4*)wl;
    const float4* i4 = (const float4*)il;
    for (int ky = 0; ky < p.kh; ++ky) {
      for (int kx = 0; kx < p.kw; ++kx) {
        const float4* wp = w4 + ((size_t)(ky * p.kw + kx) * (p.ck >> 2) + g) * wq + j;
        const float4* ip = i4 + (size_t)g * iq + ky * p.dil_y * k.twp + kx * p.dil_x;
        for (int gr = 0; gr < ngr; ++gr) {
          float4 av[MB], bv[NPB];
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = wp[m * 16];
#pragma unroll
          for (int a = 0; a < NPB; ++a) bv[a] = ip[pbase[a]];
          wp += 4 * wq;
          ip += 4 * iq;
#pragma unroll
          for (int a = 0; a < NPB; ++a)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].x, bv[a].x, acc[a][m], 0, 0, 0);
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].y, bv[a].y, acc[a][m], 0, 0, 0);
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].z, bv[a].z, acc[a][m], 0, 0, 0);
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].w, bv[a].w, acc[a][m], 0, 0, 0);
            }
        }
      }
    }

    // epilogue of tile t (as conv_quad_kernel; one channel group: cog = 0)
    const int b = t / tiles_per_img, tt = t - b * tiles_per_img;
    const int ty = tt / k.tiles_x, tx = tt - ty * k.tiles_x;
#pragma unroll
    for (int a = 0; a < NPB; ++a) {
      const int oy = ty * k.th + wave * RPW + a / XB;
      const int ox = tx * k.tw + (a % XB) * 16 + j;
      if (oy >= p.Hout || ox >= p.Wout) continue;
      const int pix = oy * p.Wout + ox;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + 4 * g + r;
          if (co >= k.cout_eff) continue;
          float v = acc[a][m][r];
          if (p.store_mode == 0) {
            if (p.bias) v += p.bias[co];
            if (p.res1.ptr) v += view_ptr(p.res1, b, co, hwout)[pix];
            if (p.res2.ptr) v += view_ptr(p.res2, b, co, hwout)[pix];
            v = act_apply(v, p.act, co);
            if (p.post.ptr) v += view_ptr(p.post, b, co, hwout)[pix];
            p.out[((size_t)b * p.out_ctot + p.out_coff + co) * (size_t)hwout + pix] = v;
          } else {
            const int q = co / p.Cout, c = co - q * p.Cout;
            if (p.bias) v += p.bias[c];
            v = act_apply(v, p.act, c);
            const int W2 = 2 * p.Wout;
            p.out[((size_t)b * p.out_ctot + p.out_coff + c) * (size_t)(4 * hwout) +
                  (size_t)(2 * oy + (q >> 1)) * W2 + 2 * ox + (q & 1)] = v;
          }
        }
      }
    }
    if (tn >= ntiles) break;
    t = tn;
  }
#undef QP_ISSUE
}

// instantiations X(NPB, MB, WREG, QREG): 4x16 / 4x32 / 8x32 tiles x 16 / 32 output channels
#define CONVQP_ALL(X) X(1, 1, 8, 2) X(1, 2, 16, 2) X(2, 1, 8, 2) X(2, 2, 16, 2) X(4, 1, 8, 4) X(4, 2, 16, 4)
#define CONVQP_DECLARE(NPB, MB, WREG, QREG) extern template __global__ void conv_quad_persist_kernel<NPB, MB, WREG, QREG>(const ConvK, const int);
#define CONVQP_DEFINE(NPB, MB, WREG, QREG) template __global__ void conv_quad_persist_kernel<NPB, MB, WREG, QREG>(const ConvK, const int);
