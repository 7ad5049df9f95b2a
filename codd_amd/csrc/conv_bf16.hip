// Host side of the split-bf16 convolution kernel (conv_bf16_kernel.h): weight packing, launch geometry, dispatch.
#include "conv_bf16_kernel.h"
#include <stdio.h>
#include <stdlib.h>

CONVB_ALL(CONVB_DECLARE)

static inline int nk_of(int ntaps, int ck) { return cdiv((long long)ntaps * (ck >> 3), 4); }

// packed image: [cog][chunk][plane (hi, lo)][k-step][g][co (16*mb)][8 bf16]; entry e = kstep*4 + g = tap*noct + oct
__global__ void conv_pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int Cin,
                                      int ntaps, int mb, int ck, int nk, int nchunks, int planes, long long total,
                                      long long co_stride, long long ci_stride, float scale) {
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
  const __bf16 hi = (__bf16)v;
  const __bf16 r = plane == 0 ? hi : (__bf16)(v - (float)hi);
  wp[e] = __builtin_bit_cast(unsigned short, r);
}

extern "C" long long codd_conv2d_packed_bytes_bf16(int Cout, int Cin, int kh, int kw, int mb, int ck, int terms) {
  if (mb < 1 || ck < 8 || (ck & 7) || !(terms == 1 || terms == 3)) return -1;
  const long long ncog = cdiv(Cout, 16 * mb), nchunks = cdiv(Cin, ck);
  return ncog * nchunks * (terms == 3 ? 2 : 1) * (long long)nk_of(kh * kw, ck) * 4 * 16 * mb * 16;
}

extern "C" int codd_conv2d_pack_weights_bf16(const float* w, void* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                             int ck, int terms, long long co_stride, long long ci_stride, float scale,
                                             void* stream) {
  const long long bytes = codd_conv2d_packed_bytes_bf16(Cout, Cin, kh, kw, mb, ck, terms);
  if (bytes <= 0 || !w || !wpacked) return CODD_EINVAL;
  const long long total = bytes / 2;
  conv_pack_bf16_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(
      w, (unsigned short*)wpacked, Cout, Cin, kh * kw, mb, ck, nk_of(kh * kw, ck), cdiv(Cin, ck), terms == 3 ? 2 : 1,
      total, co_stride, ci_stride, scale);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

template <int PGW, int CGW, int A, int B, int TERMS, int WREG, int QREG>
static int launch_b(const ConvB& k, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_bf16_kernel<PGW, CGW, A, B, TERMS, WREG, QREG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_bf16_kernel<PGW, CGW, A, B, TERMS, WREG, QREG><<<grid, PGW * CGW * 64, lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

/* layout 2 of codd_conv2d.  Field use: nw = tile rows, npb = 16-pixel units per tile row (1 or 2), mb = 16-channel
 * blocks per workgroup, ck = channels per chunk (multiple of 8), pgw x cgw = wave grid, terms = 1 | 3. */
int codd_conv2d_bf16(const codd_conv_params* pp, void* stream) {
  ConvB k;
  k.p = *pp;
  const codd_conv_params& p = k.p;
  if (p.ck < 8 || (p.ck & 7) || !(p.terms == 1 || p.terms == 3) || p.nw < 1 || p.npb < 1 || p.npb > 2 ||
      p.pgw < 1 || p.cgw < 1 || p.mb < 1 || p.mb % p.cgw)
    return CODD_EINVAL;
  const int planes = p.terms == 3 ? 2 : 1;
  k.cin = p.C0 + p.C1;
  k.ntaps = p.kh * p.kw;
  k.noct = p.ck >> 3;
  k.nchunks = cdiv(k.cin, p.ck);
  k.nk = nk_of(k.ntaps, p.ck);
  k.th = p.nw; k.xb = p.npb; k.pu = k.th * k.xb; k.tw = 16 * k.xb;
  k.thi = (k.th - 1) * p.sy + (p.kh - 1) * p.dil_y + 1;
  const int twi = (k.tw - 1) * p.sx + (p.kw - 1) * p.dil_x + 1;
  k.xoff = (4 - (p.pad_l % 4)) % 4;
  k.twp = ((k.xoff + twi + 3) / 4) * 4;
  k.twp4 = k.twp / 4;
  k.upo = k.thi * k.twp4;
  k.nunits = k.noct * k.upo;
  k.os16 = ((k.thi * k.twp + 15) / 16) * 16;
  k.iplane16 = k.noct * k.os16;
  k.nco = 16 * p.mb;
  k.wplane16 = k.nk * 4 * k.nco;
  k.wslots = planes * k.wplane16;
  k.tiles_x = cdiv(p.Wout, k.tw);
  k.tiles_y = cdiv(p.Hout, k.th);
  k.cout_eff = p.store_mode ? 4 * p.Cout : p.Cout;
  k.ncog = cdiv(k.cout_eff, k.nco);
  const size_t hwb = (size_t)p.Hin * p.Win * sizeof(float);
  k.vec_ok = (p.Win % 4 == 0) && ((uintptr_t)p.in0.ptr % 16 == 0) && (hwb % 16 == 0) &&
             (p.C1 == 0 || (uintptr_t)p.in1.ptr % 16 == 0);
  const size_t lds = ((size_t)k.wslots + (size_t)planes * k.iplane16) * 16 + (size_t)k.nk * 16;
  if (lds > 160 * 1024) return CODD_EUNSUPPORTED;
  const long long grid = (long long)k.tiles_x * k.tiles_y * k.ncog * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return CODD_EINVAL;
  const int nt = p.pgw * p.cgw * 64;
  const int a = cdiv(k.pu, p.pgw), bb = p.mb / p.cgw;
  const int wr = cdiv(k.wslots, nt), qr = cdiv(k.nunits, nt);
  hipStream_t s = (hipStream_t)stream;
#define X(PGW, CGW, A, B, WREG, QREG)                                                                      \
  if (p.pgw == PGW && p.cgw == CGW && a == A && bb == B && wr <= WREG && qr <= QREG)                       \
    return p.terms == 3 ? launch_b<PGW, CGW, A, B, 3, WREG, QREG>(k, lds, (int)grid, s)                    \
                        : launch_b<PGW, CGW, A, B, 1, WREG, QREG>(k, lds, (int)grid, s);
  CONVB_ALL(X)
#undef X
  return CODD_EUNSUPPORTED;
}
