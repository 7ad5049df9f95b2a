#pragma once
// Split-bf16 convolution kernel: implicit GEMM on v_mfma_f32_16x16x32_bf16 (16x the rate of the f32-input MFMA).
//
// Numerics.  TERMS = 3 ("split-bf16", the fp32-grade path): every fp32 operand x is split into two bf16 numbers
//   hi = bf16_rne(x), lo = bf16_rne(x - hi)          (|x - hi - lo| <= 2^-18 |x|),
// and a product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi -- three bf16 MFMAs whose products are exact
// in the fp32 accumulator; the dropped terms (a_lo*b_lo and the two split residuals) are <= 3 * 2^-18 |a b|, i.e.
// about 16 mantissa bits per product against fp32's 24 (TF32, the reference's own CUDA conv arithmetic under
// torch >= 1.12 defaults, keeps 10).  TERMS = 1 is the plain bf16-operand / fp32-accumulate path of
// BASELINE.json configs[4].
//
// GEMM view (same roles as conv_kernel.h):  D[co, pixel] = sum_k W[co, k] X[k, pixel],  k = (tap, ci).
//   A operand (16 x 32) = weights : lane (j = l&15, g = l>>4) holds W[co = j][entry g][8 channels]
//   B operand (32 x 16) = im2col  : lane (j, g) holds X[entry g][8 channels][pixel j]
//   D (16 x 16)                   : lane holds D[co = 4g + r][pixel = j], r = 0..3
// A k-step consumes four ENTRIES; an entry is (tap, channel octet) in the order e = tap * noct + oct.  The LDS byte
// offset of every entry is kept in a small table (built once per workgroup), so any (kh, kw, dilation, stride, ck)
// runs through the same loop; entries beyond ntaps * noct point at entry 0 and carry zero weights.
//
// LDS images (16-byte slots = 8 bf16, one ds_read_b128 per operand fragment; hi and lo planes back to back):
//   weights [plane][k-step][g][co (16 * mb)][8]        = the packed global layout, copied 16 bytes at a time
//   input   [plane][octet][y][x][8]                    fp32 NCHW -> (hi, lo) converted while staging; the octet
//                                                       stride is a multiple of 256 B so that the four 16-lane
//                                                       groups of a ds_read_b128 never meet on a bank
// Workgroup = PGW x CGW waves: wave (pg, cg) owns pixel units [pg*A, pg*A + A) (a unit = 16 consecutive pixels of
// one tile row; the tile has th rows x xb units) and the B 16-channel blocks [cg*B, cg*B + B) of the workgroup's
// 16*mb output channels (mb = B * CGW): A*B accumulator tiles per wave, (A + B) fragment reads per plane and k-step.
#include "conv_kernel.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvB {
  codd_conv_params p;
  int cin, nchunks, ntaps, noct;  // noct = ck / 8 channel octets per chunk
  int nk;                         // k-steps per chunk = ceil(ntaps * noct / 4)
  int th, tw, thi, twp, twp4, xoff;
  int pu, xb;                     // pixel units per tile (th * xb), units per tile row
  int upo;                        // staging units (float4 columns) per octet = thi * twp4
  int nunits;                     // noct * upo
  int os16;                       // 16-byte slots per octet plane of the input image (multiple of 16)
  int iplane16;                   // slots per precision plane of the input image = noct * os16
  int wplane16;                   // slots per precision plane of the weight image = nk * 4 * nco
  int nco;                        // output channels per workgroup = 16 * mb
  int wslots;                     // slots of one (channel group, chunk) weight image = planes * wplane16
  int tiles_x, tiles_y, ncog, cout_eff;
  int vec_ok;
};

__device__ __forceinline__ void split8(const float* v, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)v[i];
    h[i] = hh;
    l[i] = (__bf16)(v[i] - (float)hh);
  }
}

template <int PGW, int CGW, int A, int B, int TERMS, int WREG, int QREG>
__global__ __launch_bounds__(PGW * CGW * 64) void conv_bf16_kernel(const ConvB k) {
  constexpr int NT = PGW * CGW * 64;
  constexpr int NPL = TERMS == 1 ? 1 : 2;  // precision planes
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* wl = smem4;                              // weight image
  uint4* il = smem4 + k.wslots;                   // input image
  int* etab = (int*)(il + NPL * k.iplane16);      // entry table: slot offset of (k-step, g) inside an input plane
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int cgi = wave % CGW, pgi = wave / CGW;

  int bid = blockIdx.x;
  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y; bid /= k.tiles_y;
  const int cog = bid % k.ncog;
  const int b = bid / k.ncog;

  const int hwin = p.Hin * p.Win;
  const int gy0 = ty * k.th * p.sy - p.pad_t;
  const int gxs = tx * k.tw * p.sx - p.pad_l - k.xoff;  // 4-aligned start column (may be negative)

  // ---- entry table (once) ----------------------------------------------------------------------------
  for (int e = tid; e < k.nk * 4; e += NT) {
    const int tap = e / k.noct, oct = e - tap * k.noct;
    int off = 0;
    if (tap < k.ntaps) {
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      off = oct * k.os16 + ky * p.dil_y * k.twp + kx * p.dil_x;
    }
    etab[e] = off;
  }

  // ---- per-thread staging units: (octet, row, float4 column) -------------------------------------------
  int q_lds[QREG], q_g[QREG], q_c[QREG];
  unsigned q_m[QREG];
#pragma unroll
  for (int r = 0; r < QREG; ++r) {
    const int u = tid + r * NT;
    q_c[r] = -1; q_m[r] = 0; q_lds[r] = 0; q_g[r] = 0;
    if (u < k.nunits) {
      const int oct = u / k.upo, rem = u - oct * k.upo;
      const int y = rem / k.twp4, x4 = rem - y * k.twp4;
      const int gy = gy0 + y, gx = gxs + 4 * x4;
      unsigned m = 0;
      if ((unsigned)gy < (unsigned)p.Hin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) m |= ((unsigned)(gx + q) < (unsigned)p.Win) ? (1u << q) : 0u;
      }
      q_c[r] = 8 * oct; q_m[r] = m;
      q_lds[r] = oct * k.os16 + y * k.twp + 4 * x4;
      q_g[r] = gy * p.Win + gx;
    }
  }
  uint4 wreg[WREG];
  float4 ireg[QREG][8];

#define BF_ISSUE(CH)                                                                                      \
  {                                                                                                       \
    const uint4* src_ = (const uint4*)p.wpacked + ((size_t)(cog * k.nchunks + (CH))) * k.wslots;          \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      wreg[r] = e < k.wslots ? src_[e] : make_uint4(0u, 0u, 0u, 0u);                                      \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < QREG; ++r) {                                                    \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                     \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
        const int cg = (CH) * p.ck + q_c[r] + c;                                                          \
        if (q_c[r] >= 0 && q_m[r] && cg < k.cin) {                                                        \
          const float* s_ =                                                                               \
              (cg < p.C0 ? view_ptr(p.in0, b, cg, hwin) : view_ptr(p.in1, b, cg - p.C0, hwin)) + q_g[r];  \
          if (q_m[r] == 0xFu && k.vec_ok) {                                                               \
            v = *(const float4*)s_;                                                                       \
          } else {                                                                                        \
            if (q_m[r] & 1u) v.x = s_[0];                                                                 \
            if (q_m[r] & 2u) v.y = s_[1];                                                                 \
            if (q_m[r] & 4u) v.z = s_[2];                                                                 \
            if (q_m[r] & 8u) v.w = s_[3];                                                                 \
          }                                                                                               \
        }                                                                                                 \
        ireg[r][c] = v;                                                                                   \
      }                                                                                                   \
    }                                                                                                     \
  }
#define BF_COMMIT()                                                                                       \
  {                                                                                                       \
    _Pragma("unroll") for (int r = 0; r < WREG; ++r) {                                                    \
      const int e = tid + r * NT;                                                                         \
      if (e < k.wslots) wl[e] = wreg[r];                                                                  \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < QREG; ++r) if (q_c[r] >= 0) {                                   \
      uint4* d_ = il + q_lds[r]; /* 4 pixels x 8 channels: register transpose + precision split */        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
        float v_[8];                                                                                      \
        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                     \
          v_[c] = q == 0 ? ireg[r][c].x : q == 1 ? ireg[r][c].y : q == 2 ? ireg[r][c].z : ireg[r][c].w;   \
        bf16x8 h_, l_;                                                                                    \
        split8(v_, h_, l_);                                                                               \
        d_[q] = __builtin_bit_cast(uint4, h_);                                                            \
        if (NPL == 2) d_[k.iplane16 + q] = __builtin_bit_cast(uint4, l_);                                 \
      }                                                                                                   \
    }                                                                                                     \
  }

  f32x4 acc[A][B];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int m = 0; m < B; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // slot of the lane's pixel inside an octet plane, per pixel unit (units past the tile are clamped: their
  // results are never stored)
  int pbase[A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    int u = pgi * A + a;
    if (u >= k.pu) u = k.pu - 1;
    const int prow = u / k.xb, pcol = (u - prow * k.xb) * 16 + j;
    pbase[a] = prow * p.sy * k.twp + pcol * p.sx + k.xoff;
  }
  const uint4* wlane = wl + (size_t)g * k.nco + cgi * B * 16 + j;  // + kstep * 4 * nco + m * 16
  const int wstep = 4 * k.nco;

  BF_ISSUE(0);
  for (int ch = 0; ch < k.nchunks; ++ch) {
    __syncthreads();  // every wave is done reading the previous chunk (and the entry table is written)
    BF_COMMIT();
    __syncthreads();
    if (ch + 1 < k.nchunks) BF_ISSUE(ch + 1);
    const uint4* wp = wlane;
    for (int ks = 0; ks < k.nk; ++ks) {
      const int eo = etab[ks * 4 + g];
      bf16x8 ah[B], al[B], bh[A], bl[A];
#pragma unroll
      for (int m = 0; m < B; ++m) {
        ah[m] = __builtin_bit_cast(bf16x8, wp[m * 16]);
        if (TERMS == 3) al[m] = __builtin_bit_cast(bf16x8, wp[k.wplane16 + m * 16]);
      }
#pragma unroll
      for (int a = 0; a < A; ++a) {
        bh[a] = __builtin_bit_cast(bf16x8, il[eo + pbase[a]]);
        if (TERMS == 3) bl[a] = __builtin_bit_cast(bf16x8, il[k.iplane16 + eo + pbase[a]]);
      }
      wp += wstep;
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int m = 0; m < B; ++m) {
          if (TERMS == 3) {  // small terms first
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[a], acc[a][m], 0, 0, 0);
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[a], acc[a][m], 0, 0, 0);
          }
          acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[a], acc[a][m], 0, 0, 0);
        }
    }
  }
#undef BF_ISSUE
#undef BF_COMMIT

  // epilogue (as conv_mfma_kernel)
  const int hwout = p.Hout * p.Wout;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int u = pgi * A + a;
    if (u >= k.pu) continue;
    const int prow = u / k.xb;
    const int oy = ty * k.th + prow;
    const int ox = tx * k.tw + (u - prow * k.xb) * 16 + j;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    const int pix = oy * p.Wout + ox;
#pragma unroll
    for (int m = 0; m < B; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (cog * CGW * B + cgi * B + m) * 16 + 4 * g + r;
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

// ---- instantiation list: X(PGW, CGW, A, B, WREG, QREG), each for TERMS = 1 and 3 ----------------------------
//   (2,2,5,2): 9/10-unit tiles x 64 channels (the 72x120 GRU maps: 256 workgroups of 144 px x 64 co)
//   (4,1,4,4) / (4,1,4,2) / (4,1,4,1): 16-unit tiles (8 x 32 px) x 64 / 32 / 16 channels
//   (4,1,2,2) / (4,1,2,1): 8-unit tiles (4 x 32 or 8 x 16 px) x 32 / 16 channels (small maps)
//   (4,1,8,1): 32-unit tiles (16 x 32 px) x 16 channels (full-resolution 16-channel layers)
#define CONVB_GROUP_A(X) X(2, 2, 5, 2, 10, 1) X(2, 2, 5, 2, 20, 2)
#define CONVB_GROUP_B(X) X(4, 1, 4, 4, 10, 2) X(4, 1, 4, 4, 20, 3)
#define CONVB_GROUP_C(X) X(4, 1, 4, 2, 6, 2) X(4, 1, 4, 2, 12, 3)
#define CONVB_GROUP_D(X) X(4, 1, 4, 1, 4, 2) X(4, 1, 8, 1, 4, 4)
#define CONVB_GROUP_E(X) X(4, 1, 2, 2, 6, 1) X(4, 1, 2, 2, 12, 2) X(4, 1, 2, 1, 4, 1) X(4, 1, 2, 1, 8, 2)
#define CONVB_ALL(X) CONVB_GROUP_A(X) CONVB_GROUP_B(X) CONVB_GROUP_C(X) CONVB_GROUP_D(X) CONVB_GROUP_E(X)
#define CONVB_DECLARE(PGW, CGW, A, B, WREG, QREG)                                                     \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, WREG, QREG>(const ConvB);      \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, WREG, QREG>(const ConvB);
#define CONVB_DEFINE(PGW, CGW, A, B, WREG, QREG)                                                      \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, WREG, QREG>(const ConvB);             \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, WREG, QREG>(const ConvB);
