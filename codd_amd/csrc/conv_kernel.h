#pragma once
// Kernel template of the convolution family; instantiated in conv_inst_*.hip (one translation unit per
// group so that the instantiations compile in parallel) and launched from conv.hip.
#include "common.h"
#include <stdint.h>
#include <string.h>

// Convolution family: implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain).
//
// GEMM view:  D[co, pixel] = sum_k W[co, k] * X[k, pixel],  k = (tap, ci).
//   MFMA A operand (16 x 4)  = weights   : lane l holds W[co = l&15][ci = c4 + (l>>4)]
//   MFMA B operand (4 x 16)  = im2col    : lane l holds X[ci = c4 + (l>>4)][pixel = l&15]
//   MFMA D (16 x 16)         : lane l holds D[co = 4*(l>>4) + r][pixel = l&15], r = 0..3
// so the 16 lanes l&15 address 16 consecutive output pixels of one row: loads and stores are
// 64-byte contiguous per channel in NCHW.
//
// Workgroup = NW waves (4 by default).  Output tile = TH x TW pixels, every wave owns NPB 16-pixel blocks and
// all MB 16-channel blocks of the workgroup's channel group.  The input halo tile of CK
// channels and the matching weight chunk are staged in LDS per chunk; channel stride of the
// LDS image is padded to 16 (mod 32) floats so that the two ci-groups of a 32-lane half hit
// disjoint banks.

struct ConvK {
  codd_conv_params p;
  int cin, nchunks, ntaps;
  int th, tw, thi, twi, chs;
  int wrow, wchunk;
  int tiles_x, tiles_y, ncog;
  int cout_eff;
  int twp;     // LDS row stride (floats, multiple of 4) of the input tile
  int xoff;    // column of the tile's first input pixel inside the 4-aligned LDS row
  int twp4;    // float4 units per LDS row
  int upc;     // float4 units per channel = thi * twp4
  int nunits;  // ck * upc
  int vec_ok;  // 16-byte global loads allowed (Win % 4 == 0 and 16-byte aligned bases)
  int xcd;     // 1: workgroup b (which the dispatcher places on XCD b % 8) takes work item conv_xcd_item(b): every XCD
               // walks ONE contiguous range of tiles, so the halo rows neighbouring tiles share are re-read from that
               // XCD's own L2 instead of being fetched by up to 8 L2s
};

// work item of workgroup ``bid`` under the XCD-contiguous mapping: XCD x = bid % 8 owns items [start_x, start_x + n_x)
__device__ __forceinline__ int conv_xcd_item(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
}

__device__ __forceinline__ const float* view_ptr(const codd_view& v, int b, int c, int hw) {
  return v.ptr + ((size_t)b * v.ctot + v.coff + c) * (size_t)hw;
}

// Staging: a chunk (CK input channels of the halo tile + the matching packed weights) is fetched
// with 16-byte global loads into REGISTERS right before the MFMA phase of the previous chunk and
// written to LDS after it (issue early / write late), so that HBM/L2 latency overlaps the matrix
// pipe.  The (channel, row, float4-column) decomposition of a thread's units does not depend on
// the chunk and is computed once.
template <int NW, int NPB, int MB, int WREG, int IREG>
__global__ __launch_bounds__(NW * 64) void conv_mfma_kernel(const ConvK k) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* il = smem + k.wchunk;
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  constexpr int XB = NPB >= 2 ? 2 : 1;  // 16-pixel blocks along x in the tile
  constexpr int RPW = NPB / XB;         // tile rows per wave

  int bid = k.xcd ? conv_xcd_item(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y; bid /= k.tiles_y;
  const int cog = bid % k.ncog;
  const int b = bid / k.ncog;

  const int hwin = p.Hin * p.Win;
  const int gy0 = ty * k.th * p.sy - p.pad_t;
  const int gxs = tx * k.tw * p.sx - p.pad_l - k.xoff;  // 4-aligned start column (may be negative)

  // ---- per-thread staging metadata ---------------------------------------------------------------
  int u_lds[IREG], u_g[IREG], u_c[IREG];
  unsigned u_m[IREG];
#pragma unroll
  for (int r = 0; r < IREG; ++r) {
    const int u = tid + r * NT;
    u_c[r] = -1; u_m[r] = 0; u_lds[r] = 0; u_g[r] = 0;
    if (u < k.nunits) {
      const int c = u / k.upc, rem = u - c * k.upc;
      const int y = rem / k.twp4, x4 = rem - y * k.twp4;
      const int gy = gy0 + y, gx = gxs + 4 * x4;
      unsigned m = 0;
      if ((unsigned)gy < (unsigned)p.Hin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) m |= ((unsigned)(gx + q) < (unsigned)p.Win) ? (1u << q) : 0u;
      }
      u_c[r] = c; u_m[r] = m;
      u_lds[r] = c * k.chs + y * k.twp + 4 * x4;
      u_g[r] = gy * p.Win + gx;
    }
  }
  const int wchunk4 = k.wchunk >> 2;
  float4 wreg[WREG], ireg[IREG];

#define CONV_ISSUE(CH)                                                                                    \
  {                                                                                                       \
    const float4* src_ = (const float4*)(p.wpacked + ((size_t)(cog * k.nchunks + (CH))) * k.wchunk);      \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      wreg[r] = e < wchunk4 ? src_[e] : make_float4(0.f, 0.f, 0.f, 0.f);                                  \
    }                                                                                                     \
    const int c0_ = (CH) * p.ck;                                                                          \
    _Pragma("unroll") for (int r = 0; r < IREG; ++r) {                                                    \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
      const int cg = c0_ + u_c[r];                                                                        \
      if (u_c[r] >= 0 && cg < k.cin && u_m[r]) {                                                          \
        const float* s_ =                                                                                 \
            (cg < p.C0 ? view_ptr(p.in0, b, cg, hwin) : view_ptr(p.in1, b, cg - p.C0, hwin)) + u_g[r];    \
        if (u_m[r] == 0xFu && k.vec_ok) {                                                                 \
          v = *(const float4*)s_;                                                                         \
        } else {                                                                                          \
          if (u_m[r] & 1u) v.x = s_[0];                                                                   \
          if (u_m[r] & 2u) v.y = s_[1];                                                                   \
          if (u_m[r] & 4u) v.z = s_[2];                                                                   \
          if (u_m[r] & 8u) v.w = s_[3];                                                                   \
        }                                                                                                 \
      }                                                                                                   \
      ireg[r] = v;                                                                                        \
    }                                                                                                     \
  }
#define CONV_COMMIT()                                                                                     \
  {                                                                                                       \
    float4* dst_ = (float4*)wl;                                                                           \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      if (e < wchunk4) dst_[e] = wreg[r];                                                                 \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < IREG; ++r) if (u_c[r] >= 0) *(float4*)(il + u_lds[r]) = ireg[r]; \
  }

  f32x4 acc[NPB][MB];
#pragma unroll
  for (int a = 0; a < NPB; ++a)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per pixel-block LDS base offset (row / col part that does not depend on the tap)
  int pbase[NPB];
#pragma unroll
  for (int a = 0; a < NPB; ++a) {
    const int prow = wave * RPW + a / XB, pcol = (a % XB) * 16 + j;
    pbase[a] = prow * p.sy * k.twp + pcol * p.sx + k.xoff;
  }

  CONV_ISSUE(0);
  for (int ch = 0; ch < k.nchunks; ++ch) {
    __syncthreads();  // every wave is done reading the previous chunk
    CONV_COMMIT();
    __syncthreads();
    if (ch + 1 < k.nchunks) CONV_ISSUE(ch + 1);
    for (int ky = 0; ky < p.kh; ++ky) {
      for (int kx = 0; kx < p.kw; ++kx) {
        const float* wp = wl + (ky * p.kw + kx) * p.ck * k.wrow + j + g * k.wrow;
        const float* ip = il + ky * p.dil_y * k.twp + kx * p.dil_x + g * k.chs;
        const int wstep = 4 * k.wrow, istep = 4 * k.chs;
        // 4 k-steps per trip: the 4*(MB+NPB) LDS reads are issued ahead of the 4*MB*NPB MFMAs
#pragma unroll 4
        for (int c4 = 0; c4 < p.ck; c4 += 4) {
          float av[MB], bv[NPB];
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = wp[m * 16];
#pragma unroll
          for (int a = 0; a < NPB; ++a) bv[a] = ip[pbase[a]];
          wp += wstep;
          ip += istep;
#pragma unroll
          for (int a = 0; a < NPB; ++a)
#pragma unroll
            for (int m = 0; m < MB; ++m)
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[a], acc[a][m], 0, 0, 0);
        }
      }
    }
  }

  // epilogue
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
        } else {  // ConvTranspose2d k=2 s=2: co = (a2*2+b2)*Cout + c
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

// ---- instantiation lists (X(NW, NPB, MB, WREG, IREG)) -----------------------------------------------
#define CONV_REGS_NW4(X, NPB, MB) X(4, NPB, MB, 4, 4) X(4, NPB, MB, 4, 8) X(4, NPB, MB, 12, 4) X(4, NPB, MB, 12, 8) X(4, NPB, MB, 16, 8)
#define CONV_REGS_NWX(X, NW, MB) X(NW, 1, MB, 8, 4) X(NW, 1, MB, 16, 4)
#define CONV_GROUP_0(X) CONV_REGS_NW4(X, 1, 1)
#define CONV_GROUP_1(X) CONV_REGS_NW4(X, 1, 2)
#define CONV_GROUP_2(X) CONV_REGS_NW4(X, 1, 4)
#define CONV_GROUP_3(X) CONV_REGS_NW4(X, 2, 1) CONV_REGS_NW4(X, 2, 2)
#define CONV_GROUP_4(X) CONV_REGS_NW4(X, 2, 4) CONV_REGS_NW4(X, 4, 1)
#define CONV_GROUP_5(X) CONV_REGS_NW4(X, 4, 2)
#define CONV_GROUP_6(X) CONV_REGS_NW4(X, 4, 4)
#define CONV_GROUP_7(X) CONV_REGS_NWX(X, 9, 1) CONV_REGS_NWX(X, 9, 2) CONV_REGS_NWX(X, 9, 4)
#define CONV_GROUP_8(X) CONV_REGS_NWX(X, 2, 1) CONV_REGS_NWX(X, 2, 2) CONV_REGS_NWX(X, 2, 4)
#define CONV_GROUP_9(X) CONV_REGS_NWX(X, 8, 1) CONV_REGS_NWX(X, 8, 2) CONV_REGS_NWX(X, 8, 4)
#define CONV_ALL_GROUPS(X) CONV_GROUP_0(X) CONV_GROUP_1(X) CONV_GROUP_2(X) CONV_GROUP_3(X) CONV_GROUP_4(X) CONV_GROUP_5(X) CONV_GROUP_6(X) CONV_GROUP_7(X) CONV_GROUP_8(X) CONV_GROUP_9(X)
#define CONV_DECLARE(NW, NPB, MB, WREG, IREG) extern template __global__ void conv_mfma_kernel<NW, NPB, MB, WREG, IREG>(const ConvK);
#define CONV_DEFINE(NW, NPB, MB, WREG, IREG) template __global__ void conv_mfma_kernel<NW, NPB, MB, WREG, IREG>(const ConvK);
