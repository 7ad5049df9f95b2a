// Host side of the split-bf16 convolution kernel (conv_bf16_kernel.h): weight packing, launch geometry, dispatch.
#include "conv_bf16_kernel.h"
#include <stdio.h>
#include <stdlib.h>

CONVB_ALL(CONVB_DECLARE)

static inline int nk_of(int ntaps, int ck) { return cdiv((long long)ntaps * (ck >> 3), 4); }

// packed image: [cog][chunk][plane (hi, lo)][k-step][g][co (16*mb)][8 bf16]; entry e = kstep*4 + g = tap*noct + oct
__global__ void conv_pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int Cin,
                                      int ntaps, int mb, int ck, int nk, int nchunks, int planes, long long total,
                                      long long co_stride, long long ci_stride, float scale, int f16) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int nco = 16 * mb, noct = ck >> 3;
  const int c8 = (int)(e & 7);
  long long t = e >> 3;
  const int col = (int)(t % nco); t /= nco;
  const int ent = (int)(t % (nk * 4)); t /= nk * 4;
  const int plane = (int)(t % planes); t /= planes;
  const int chunk = (int)(t % nchunks);
  const int cog = (int)(t / nchunks);
  const int tap = ent / noct, oct = ent - tap * noct;
  const int co = cog * nco + col, ci = chunk * ck + oct * 8 + c8;
  float v = 0.f;
  if (tap < ntaps && co < Cout && ci < Cin) v = w[(size_t)co * co_stride + (size_t)ci * ci_stride + tap] * scale;
  if (f16) {  // CODD_TERMS_F16 / _SPLIT_F16: IEEE fp16 (hi | lo) plane(s)
    const _Float16 h = (_Float16)v;
    wp[e] = __builtin_bit_cast(unsigned short, plane == 0 ? h : (_Float16)(v - (float)h));
    return;
  }
  const __bf16 hi = (__bf16)v;
  const __bf16 r = plane == 0 ? hi : (__bf16)(v - (float)hi);
  wp[e] = __builtin_bit_cast(unsigned short, r);
}

extern "C" long long codd_conv2d_packed_bytes_bf16(int Cout, int Cin, int kh, int kw, int mb, int ck, int terms) {
  if (mb < 1 || ck < 8 || (ck & 7) || !CODD_TERMS_OK(terms)) return -1;
  const long long ncog = cdiv(Cout, 16 * mb), nchunks = cdiv(Cin, ck);
  return ncog * nchunks * CODD_TERMS_PLANES(terms) * (long long)nk_of(kh * kw, ck) * 4 * 16 * mb * 16;
}

extern "C" int codd_conv2d_pack_weights_bf16(const float* w, void* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                             int ck, int terms, long long co_stride, long long ci_stride, float scale,
                                             void* stream) {
  const long long bytes = codd_conv2d_packed_bytes_bf16(Cout, Cin, kh, kw, mb, ck, terms);
  if (bytes <= 0 || !w || !wpacked) return CODD_EINVAL;
  const long long total = bytes / 2;
  conv_pack_bf16_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(
      w, (unsigned short*)wpacked, Cout, Cin, kh * kw, mb, ck, nk_of(kh * kw, ck), cdiv(Cin, ck), CODD_TERMS_PLANES(terms),
      total, co_stride, ci_stride, scale, CODD_TERMS_IS_F16(terms));
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

template <int PGW, int CGW, int A, int B, int TERMS, int OUTF, int KS>
static int launch_b(const ConvB& k, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_bf16_kernel<PGW, CGW, A, B, TERMS, OUTF, KS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_bf16_kernel<PGW, CGW, A, B, TERMS, OUTF, KS><<<grid, (PGW * CGW * KS + CONVB_NWP) * 64, lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

/* layout 2 of codd_conv2d */
int codd_conv2d_bf16(const codd_conv_params* pp, void* stream, int dry_run) {
  ConvB k;
  size_t lds;
  long long grid;
  const int rc = convb_geometry(pp, k, lds, grid, !dry_run);
  if (rc != CODD_OK) return rc;
  const codd_conv_params& p = k.p;
  const int a = cdiv(k.pu, p.pgw), bb = p.mb / p.cgw;
  hipStream_t s = (hipStream_t)stream;
  const int ks = p.ksplit == 2 ? 2 : 1;
#define X(PGW, CGW, A, B, KS)                                                                    \
  if (p.pgw == PGW && p.cgw == CGW && a == A && bb == B && ks == KS)                             \
    return dry_run ? CODD_OK                                                                     \
           : (p.xso || p.gate) ? (p.terms == 3   ? launch_b<PGW, CGW, A, B, 3, 1, KS>(k, lds, (int)grid, s)   \
                                  : p.terms == 48 ? launch_b<PGW, CGW, A, B, 48, 1, KS>(k, lds, (int)grid, s)  \
                                  : p.terms == 16 ? launch_b<PGW, CGW, A, B, 16, 1, KS>(k, lds, (int)grid, s)  \
                                                  : launch_b<PGW, CGW, A, B, 1, 1, KS>(k, lds, (int)grid, s))  \
                               : (p.terms == 3   ? launch_b<PGW, CGW, A, B, 3, 0, KS>(k, lds, (int)grid, s)   \
                                  : p.terms == 48 ? launch_b<PGW, CGW, A, B, 48, 0, KS>(k, lds, (int)grid, s)  \
                                  : p.terms == 16 ? launch_b<PGW, CGW, A, B, 16, 0, KS>(k, lds, (int)grid, s)  \
                                                  : launch_b<PGW, CGW, A, B, 1, 0, KS>(k, lds, (int)grid, s));
  CONVB_ALL(X)
#undef X
  return CODD_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------
// fp32 NCHW (one or two channel-slice views = the conv's concatenated input) -> split-bf16 records
//   xs[b][plane][octet][yp][xp][8],  pixel (y, x) of the image at (yp, xp) = (y + bt, x + bl), zeros elsewhere and
//   in the channels past C0 + C1.  One thread = one record position: 8 coalesced dword loads, 1-2 16-byte stores.
// ------------------------------------------------------------------------------------------------------------
__global__ void split_bf16_kernel(codd_view in0, codd_view in1, int C0, int C1, int B, int H, int W, int bt, int bl,
                                  int c8, int hp, int wp, int planes, uint4* __restrict__ xs, int f16) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)c8 * hp * wp;
  if (e >= (long long)B * per) return;
  const int xp = (int)(e % wp);
  long long t = e / wp;
  const int yp = (int)(t % hp); t /= hp;
  const int oct = (int)(t % c8);
  const int b = (int)(t / c8);
  const int y = yp - bt, x = xp - bl;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
    const int hw = H * W, pix = y * W + x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = oct * 8 + i;
      if (c < C0) v[i] = view_ptr(in0, b, c, hw)[pix];
      else if (c < C0 + C1) v[i] = view_ptr(in1, b, c - C0, hw)[pix];
    }
  }
  bf16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)v[i];
    h[i] = hh;
    l[i] = (__bf16)(v[i] - (float)hh);
  }
  uint4* dst = xs + (size_t)b * planes * per + ((size_t)oct * hp + yp) * wp + xp;
  if (f16) {  // CODD_TERMS_F16 / _SPLIT_F16: IEEE fp16 records (hi | lo)
    f16x8 q, ql;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = (_Float16)v[i];
      ql[i] = (_Float16)(v[i] - (float)q[i]);
    }
    dst[0] = __builtin_bit_cast(uint4, q);
    if (planes == 2) dst[per] = __builtin_bit_cast(uint4, ql);
    return;
  }
  dst[0] = __builtin_bit_cast(uint4, h);
  if (planes == 2) dst[per] = __builtin_bit_cast(uint4, l);
}

extern "C" long long codd_split_bf16_bytes(int B, int c8, int hp, int wp, int terms) {
  if (B < 1 || c8 < 1 || hp < 1 || wp < 1 || !CODD_TERMS_OK(terms)) return -1;
  return (long long)B * CODD_TERMS_PLANES(terms) * c8 * hp * wp * 16;
}

extern "C" int codd_split_bf16(codd_view in0, int C0, codd_view in1, int C1, int B, int H, int W, int bt, int bl,
                               int c8, int hp, int wp, int terms, void* xs, void* stream) {
  if (!in0.ptr || C0 < 1 || C1 < 0 || (C1 > 0 && !in1.ptr) || !xs || bt < 0 || bl < 0 || hp < bt + H || wp < bl + W ||
      8 * c8 < C0 + C1 || codd_split_bf16_bytes(B, c8, hp, wp, terms) <= 0)
    return CODD_EINVAL;
  const long long total = (long long)B * c8 * hp * wp;
  split_bf16_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in0, in1, C0, C1, B, H, W, bt, bl, c8, hp, wp,
                                                                       CODD_TERMS_PLANES(terms), (uint4*)xs, CODD_TERMS_IS_F16(terms));
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
