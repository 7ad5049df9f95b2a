#pragma once
// "Quad" variant of the convolution kernel: operands are stored in LDS with FOUR input channels
// innermost, so that one ds_read_b128 per lane feeds four k-steps of v_mfma_f32_16x16x4_f32
// (tools/ubench/mfma_lds.hip: 139 vs 126 TFLOP/s for the LDS-fed loop at 2 waves/SIMD).
//
// k mapping inside a group of 16 input channels: k-step s (0..3), lane group g (0..3) <-> channel 4g + s, i.e.
// lane (g, j) reads the 16 bytes "channels 4g..4g+3" once and uses component s in step s -- for both operands:
//   weights  LDS image  [tap][cq = c/4][co (16*MB)][4]   (= the packed global layout, straight float4 copy)
//   input    LDS image  [cq][y][x][4]                    (transposed while staging: a thread loads the same
//                                                          float4 of pixels from 4 channels and writes 4 pixels)
// 16 consecutive lanes read 16 consecutive 16-byte slots (256 B = every bank once): no padding needed.
// Restrictions (checked on the host): ck = 16 or 32, 16-byte aligned rows (vec_ok).  With an x-stride of 2 the
// 16 lanes of an operand read are 32 bytes apart (2-way bank conflict, accepted).
#include "conv_kernel.h"

// body of the kernel for workgroup ``bid`` of the launch described by ``k`` (shared by the single- and the multi-job
// entry points; always inlined with a compile-time-known ``k`` operand so that the kernel arguments stay scalar loads)
template <int NW, int NPB, int MB, int WREG, int QREG>
__device__ __forceinline__ void conv_quad_body(const ConvK& k, int bid, float* smem) {
  constexpr int NT = NW * 64;
  float* wl = smem;
  float* il = smem + k.wchunk;
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  constexpr int XB = NPB >= 2 ? 2 : 1;
  constexpr int RPW = NPB / XB;

  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y; bid /= k.tiles_y;
  const int cog = bid % k.ncog;
  const int b = bid / k.ncog;

  const int hwin = p.Hin * p.Win;
  const int gy0 = ty * k.th * p.sy - p.pad_t;
  const int gxs = tx * k.tw * p.sx - p.pad_l - k.xoff;

  // ---- per-thread quad units: (channel quad, row, float4 column) ----------------------------------
  const int ncq = p.ck >> 2;            // channel quads per chunk
  const int nq = ncq * k.upc;           // quad units per chunk
  int q_lds[QREG], q_g[QREG], q_c[QREG];  // LDS float offset of pixel 0 / global offset in a channel / first channel
  bool q_in[QREG];
#pragma unroll
  for (int r = 0; r < QREG; ++r) {
    const int u = tid + r * NT;
    q_c[r] = -1; q_in[r] = false; q_lds[r] = 0; q_g[r] = 0;
    if (u < nq) {
      const int cq = u / k.upc, rem = u - cq * k.upc;
      const int y = rem / k.twp4, x4 = rem - y * k.twp4;
      const int gy = gy0 + y, gx = gxs + 4 * x4;
      q_c[r] = 4 * cq;
      q_in[r] = (unsigned)gy < (unsigned)p.Hin && gx >= 0 && gx + 3 < p.Win;  // vec_ok: all four pixels or none
      q_lds[r] = ((cq * k.thi + y) * k.twp + 4 * x4) * 4;
      q_g[r] = gy * p.Win + gx;
    }
  }
  const int wchunk4 = k.wchunk >> 2;
  float4 wreg[WREG], ireg[QREG][4];

#define QUAD_ISSUE(CH)                                                                                    \
  {                                                                                                       \
    const float4* src_ = (const float4*)(p.wpacked + ((size_t)(cog * k.nchunks + (CH))) * k.wchunk);      \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      wreg[r] = e < wchunk4 ? src_[e] : make_float4(0.f, 0.f, 0.f, 0.f);                                  \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < QREG; ++r) {                                                    \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                     \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
        const int cg = (CH) * p.ck + q_c[r] + c;                                                          \
        if (q_c[r] >= 0 && q_in[r] && cg < k.cin) {                                                       \
          const float* s_ =                                                                               \
              (cg < p.C0 ? view_ptr(p.in0, b, cg, hwin) : view_ptr(p.in1, b, cg - p.C0, hwin)) + q_g[r];  \
          v = *(const float4*)s_;                                                                         \
        }                                                                                                 \
        ireg[r][c] = v;                                                                                   \
      }                                                                                                   \
    }                                                                                                     \
  }
#define QUAD_COMMIT()                                                                                     \
  {                                                                                                       \
    float4* dst_ = (float4*)wl;                                                                           \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      if (e < wchunk4) dst_[e] = wreg[r];                                                                 \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < QREG; ++r) if (q_c[r] >= 0) {                                   \
      float4* d_ = (float4*)(il + q_lds[r]); /* 4 pixels x (4 channels innermost): register transpose */  \
      d_[0] = make_float4(ireg[r][0].x, ireg[r][1].x, ireg[r][2].x, ireg[r][3].x);                        \
      d_[1] = make_float4(ireg[r][0].y, ireg[r][1].y, ireg[r][2].y, ireg[r][3].y);                        \
      d_[2] = make_float4(ireg[r][0].z, ireg[r][1].z, ireg[r][2].z, ireg[r][3].z);                        \
      d_[3] = make_float4(ireg[r][0].w, ireg[r][1].w, ireg[r][2].w, ireg[r][3].w);                        \
    }                                                                                                     \
  }

  f32x4 acc[NPB][MB];
#pragma unroll
  for (int a = 0; a < NPB; ++a)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pbase[NPB];  // float4 index of the pixel inside a channel-quad plane (tap-independent part)
#pragma unroll
  for (int a = 0; a < NPB; ++a) {
    const int prow = wave * RPW + a / XB, pcol = (a % XB) * 16 + j;
    pbase[a] = prow * p.sy * k.twp + pcol * p.sx + k.xoff;
  }
  const int wq = 16 * MB;           // float4 per (tap, channel quad) of the weights
  const int iq = k.thi * k.twp;     // float4 per channel quad of the input tile
  const int ngr = p.ck >> 4;        // groups of 16 channels per chunk

  QUAD_ISSUE(0);
  for (int ch = 0; ch < k.nchunks; ++ch) {
    __syncthreads();
    QUAD_COMMIT();
    __syncthreads();
    if (ch + 1 < k.nchunks) QUAD_ISSUE(ch + 1);
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
  }
#undef QUAD_ISSUE
#undef QUAD_COMMIT

  // epilogue (identical to conv_mfma_kernel)
  const int hwout = p.Hout * p.Wout;
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
        const int co = (cog * MB + m) * 16 + 4 * g + r;
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
}

template <int NW, int NPB, int MB, int WREG, int QREG>
__global__ __launch_bounds__(NW * 64) void conv_quad_kernel(const ConvK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_quad_body<NW, NPB, MB, WREG, QREG>(k, k.xcd ? conv_xcd_item(blockIdx.x, gridDim.x) : blockIdx.x, smem);
}

// Several INDEPENDENT convolutions (different tensors, shapes and channel counts; same tile / register class) as one
// launch: workgroups [start[j], start[j + 1]) run job j.  For the launch-bound small layers of HRNet's resolution
// branches (mmseg HRModule: the branches of a module do not depend on each other): one ~10 us launch instead of 2-4.
constexpr int CONVQ_MULTI_MAX = 4;
struct ConvKN {
  ConvK k[CONVQ_MULTI_MAX];
  int start[CONVQ_MULTI_MAX + 1];
};
template <int NW, int NPB, int MB, int WREG, int QREG>
__global__ __launch_bounds__(NW * 64) void conv_quad_multi_kernel(const ConvKN kn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  if (bid < kn.start[1]) conv_quad_body<NW, NPB, MB, WREG, QREG>(kn.k[0], bid, smem);
  else if (bid < kn.start[2]) conv_quad_body<NW, NPB, MB, WREG, QREG>(kn.k[1], bid - kn.start[1], smem);
  else if (bid < kn.start[3]) conv_quad_body<NW, NPB, MB, WREG, QREG>(kn.k[2], bid - kn.start[2], smem);
  else conv_quad_body<NW, NPB, MB, WREG, QREG>(kn.k[3], bid - kn.start[3], smem);
}

// instantiations (X(NW, NPB, MB, WREG, QREG)): 4-wave workgroups with 4x16 / 4x32 / 8x32 tiles, 9-wave with 4x16
#define CONVQ_GROUP_A(X) X(4, 1, 2, 8, 1) X(4, 1, 2, 16, 2) X(4, 1, 4, 12, 1) X(4, 1, 4, 16, 2) X(4, 1, 1, 8, 2)
#define CONVQ_GROUP_B(X) X(4, 2, 2, 8, 2) X(4, 2, 2, 16, 2) X(4, 2, 4, 12, 2) X(4, 2, 4, 16, 2) X(4, 2, 1, 8, 2)
#define CONVQ_GROUP_C(X) X(9, 1, 2, 8, 1) X(9, 1, 4, 8, 1) X(9, 1, 4, 16, 1) X(9, 1, 1, 8, 1) X(4, 4, 1, 8, 4) X(4, 4, 2, 8, 4)
#define CONVQ_ALL(X) CONVQ_GROUP_A(X) CONVQ_GROUP_B(X) CONVQ_GROUP_C(X)
#define CONVQ_DECLARE(NW, NPB, MB, WREG, QREG) extern template __global__ void conv_quad_kernel<NW, NPB, MB, WREG, QREG>(const ConvK);
#define CONVQ_DEFINE(NW, NPB, MB, WREG, QREG) template __global__ void conv_quad_kernel<NW, NPB, MB, WREG, QREG>(const ConvK);
// the multi-job kernel is instantiated for the small-layer class only: 4 x 16 tiles, 16 or 32 output channels per group
#define CONVQ_MULTI(X) X(4, 1, 1, 8, 2) X(4, 1, 2, 16, 2)
#define CONVQM_DECLARE(NW, NPB, MB, WREG, QREG) extern template __global__ void conv_quad_multi_kernel<NW, NPB, MB, WREG, QREG>(const ConvKN);
#define CONVQM_DEFINE(NW, NPB, MB, WREG, QREG) template __global__ void conv_quad_multi_kernel<NW, NPB, MB, WREG, QREG>(const ConvKN);
