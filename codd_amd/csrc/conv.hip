// Convolution family: implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain).
//
// GEMM view:  D[co, pixel] = sum_k W[co, k] * X[k, pixel],  k = (tap, ci).
//   MFMA A operand (16 x 4)  = weights   : lane l holds W[co = l&15][ci = c4 + (l>>4)]
//   MFMA B operand (4 x 16)  = im2col    : lane l holds X[ci = c4 + (l>>4)][pixel = l&15]
//   MFMA D (16 x 16)         : lane l holds D[co = 4*(l>>4) + r][pixel = l&15], r = 0..3
// so the 16 lanes l&15 address 16 consecutive output pixels of one row: loads and stores are
// 64-byte contiguous per channel in NCHW.
//
// Workgroup = 4 waves.  Output tile = TH x TW pixels, every wave owns NPB 16-pixel blocks and
// all MB 16-channel blocks of the workgroup's channel group.  The input halo tile of CK
// channels and the matching weight chunk are staged in LDS per chunk; channel stride of the
// LDS image is padded to 16 (mod 32) floats so that the two ci-groups of a 32-lane half hit
// disjoint banks.
#include "common.h"
#include <string.h>

struct ConvK {
  codd_conv_params p;
  int cin, nchunks, ntaps;
  int th, tw, thi, twi, chs;
  int wrow, wchunk;
  int tiles_x, tiles_y, ncog;
  int cout_eff;
  int step_c, step_y, step_x;  // decomposition of 256 in (c, y, x) units of the LDS tile
};

__device__ __forceinline__ const float* view_ptr(const codd_view& v, int b, int c, int hw) {
  return v.ptr + ((size_t)b * v.ctot + v.coff + c) * (size_t)hw;
}

template <int NPB, int MB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* il = smem + k.wchunk;
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  constexpr int XB = NPB >= 2 ? 2 : 1;  // 16-pixel blocks along x in the tile
  constexpr int RPW = NPB / XB;         // tile rows per wave

  int bid = blockIdx.x;
  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y; bid /= k.tiles_y;
  const int cog = bid % k.ncog;
  const int b = bid / k.ncog;

  const int hwin = p.Hin * p.Win;
  const int gy0 = ty * k.th * p.sy - p.pad_t;
  const int gx0 = tx * k.tw * p.sx - p.pad_l;

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
    pbase[a] = prow * p.sy * k.twi + pcol * p.sx;
  }

  const int per = k.thi * k.twi;
  const int tot = p.ck * per;
  for (int ch = 0; ch < k.nchunks; ++ch) {
    __syncthreads();
    {  // weights: straight copy of the packed chunk
      const float4* src = (const float4*)(p.wpacked + ((size_t)(cog * k.nchunks + ch)) * k.wchunk);
      float4* dst = (float4*)wl;
      for (int e = tid; e < (k.wchunk >> 2); e += 256) dst[e] = src[e];
    }
    {  // input halo tile, zero filled outside the image / beyond Cin
      const int c0 = ch * p.ck;
      int c = tid / per, r = tid - c * per;
      int y = r / k.twi, x = r - y * k.twi;
      for (int e = tid; e < tot; e += 256) {
        const int cg = c0 + c, gy = gy0 + y, gx = gx0 + x;
        float v = 0.f;
        if (cg < k.cin && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win) {
          const float* s = cg < p.C0 ? view_ptr(p.in0, b, cg, hwin) : view_ptr(p.in1, b, cg - p.C0, hwin);
          v = s[gy * p.Win + gx];
        }
        il[c * k.chs + y * k.twi + x] = v;
        x += k.step_x;
        if (x >= k.twi) { x -= k.twi; ++y; }
        y += k.step_y;
        if (y >= k.thi) { y -= k.thi; ++c; }
        c += k.step_c;
      }
    }
    __syncthreads();
    for (int ky = 0; ky < p.kh; ++ky) {
      for (int kx = 0; kx < p.kw; ++kx) {
        const float* wt = wl + (ky * p.kw + kx) * p.ck * k.wrow + j;
        const float* it = il + ky * p.dil_y * k.twi + kx * p.dil_x;
        for (int c4 = 0; c4 < p.ck; c4 += 4) {
          const int c = c4 + g;
          float av[MB], bv[NPB];
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = wt[c * k.wrow + m * 16];
#pragma unroll
          for (int a = 0; a < NPB; ++a) bv[a] = it[c * k.chs + pbase[a]];
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

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int ntaps,
                                 int mb, int ck, int wrow, int nchunks, long long total, long long co_stride,
                                 long long ci_stride, float scale) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int col = (int)(e % wrow);
  long long t = e / wrow;
  int c = (int)(t % ck); t /= ck;
  int tap = (int)(t % ntaps); t /= ntaps;
  int chunk = (int)(t % nchunks);
  int cog = (int)(t / nchunks);
  int co = cog * 16 * mb + col, ci = chunk * ck + c;
  float v = 0.f;
  if (col < 16 * mb && co < Cout && ci < Cin) v = w[(size_t)co * co_stride + (size_t)ci * ci_stride + tap] * scale;
  wp[e] = v;
}

static inline int wrow_of(int mb) { return 16 * mb + ((mb & 1) ? 0 : 16); }

extern "C" long long codd_conv2d_packed_size(int Cout, int Cin, int kh, int kw, int mb, int ck) {
  if (mb < 1 || ck < 4 || (ck & 3)) return -1;
  long long ncog = cdiv(Cout, 16 * mb), nchunks = cdiv(Cin, ck);
  return ncog * nchunks * (long long)(kh * kw) * ck * wrow_of(mb);
}

extern "C" int codd_conv2d_pack_weights_ex(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                           int ck, long long co_stride, long long ci_stride, float scale,
                                           void* stream) {
  long long total = codd_conv2d_packed_size(Cout, Cin, kh, kw, mb, ck);
  if (total <= 0 || !w || !wpacked) return CODD_EINVAL;
  int nchunks = cdiv(Cin, ck);
  conv_pack_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wpacked, Cout, Cin, kh * kw, mb, ck,
                                                                      wrow_of(mb), nchunks, total, co_stride,
                                                                      ci_stride, scale);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_conv2d_pack_weights(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                        int ck, void* stream) {
  return codd_conv2d_pack_weights_ex(w, wpacked, Cout, Cin, kh, kw, mb, ck, (long long)Cin * kh * kw,
                                     (long long)kh * kw, 1.f, stream);
}

template <int NPB, int MB>
static int launch_conv(const ConvK& k, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<NPB, MB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_mfma_kernel<NPB, MB><<<grid, 256, lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_conv2d(const codd_conv_params* pp, void* stream) {
  if (!pp) return CODD_EINVAL;
  ConvK k;
  k.p = *pp;
  const codd_conv_params& p = k.p;
  if (!p.in0.ptr || !p.out || !p.wpacked || p.C0 <= 0 || p.C1 < 0 || (p.C1 > 0 && !p.in1.ptr)) return CODD_EINVAL;
  if (p.ck < 4 || (p.ck & 3) || p.B < 1 || p.Cout < 1 || p.kh < 1 || p.kw < 1) return CODD_EINVAL;
  if (!(p.mb == 1 || p.mb == 2 || p.mb == 4) || !(p.npb == 1 || p.npb == 2 || p.npb == 4)) return CODD_EINVAL;
  if (p.store_mode && (p.kh != 1 || p.kw != 1 || p.res1.ptr || p.res2.ptr || p.post.ptr)) return CODD_EUNSUPPORTED;
  k.cin = p.C0 + p.C1;
  k.nchunks = cdiv(k.cin, p.ck);
  k.ntaps = p.kh * p.kw;
  const int xb = p.npb >= 2 ? 2 : 1, rpw = p.npb / xb;
  k.th = 4 * rpw;
  k.tw = 16 * xb;
  k.thi = (k.th - 1) * p.sy + (p.kh - 1) * p.dil_y + 1;
  k.twi = (k.tw - 1) * p.sx + (p.kw - 1) * p.dil_x + 1;
  int per = k.thi * k.twi;
  if (p.sx == 1) k.chs = ((per + 15) / 32) * 32 + 16;  // == 16 (mod 32), >= per
  else k.chs = per | 1;
  if (k.chs < per) k.chs += 32;
  k.wrow = wrow_of(p.mb);
  k.wchunk = k.ntaps * p.ck * k.wrow;
  k.tiles_x = cdiv(p.Wout, k.tw);
  k.tiles_y = cdiv(p.Hout, k.th);
  k.cout_eff = p.store_mode ? 4 * p.Cout : p.Cout;
  k.ncog = cdiv(k.cout_eff, 16 * p.mb);
  k.step_c = 256 / per;
  int rem = 256 - k.step_c * per;
  k.step_y = rem / k.twi;
  k.step_x = rem - k.step_y * k.twi;
  size_t lds = ((size_t)k.wchunk + (size_t)p.ck * k.chs) * sizeof(float);
  if (lds > 160 * 1024) return CODD_EUNSUPPORTED;
  long long grid = (long long)k.tiles_x * k.tiles_y * k.ncog * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
#define CASE(N, M) if (p.npb == N && p.mb == M) return launch_conv<N, M>(k, lds, (int)grid, s)
  CASE(1, 1); CASE(1, 2); CASE(1, 4);
  CASE(2, 1); CASE(2, 2); CASE(2, 4);
  CASE(4, 1); CASE(4, 2); CASE(4, 4);
#undef CASE
  return CODD_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// All-pairs correlation pyramid (reference blocks/corr.py:28-45,56-62) as four 1x1 "convolutions":
// output channel = source pixel n1 (weights = f1^T / 16, re-packed every frame), image = f2 pooled
// i times.  avg_pool2d commutes with the inner product, so level i never reads level i-1.
// ------------------------------------------------------------------------------------------------
extern "C" int codd_avgpool2(const float* in, int BC, int h, int w, float* out, void* stream);

#define CORR_MB 4
#define CORR_CK 32
extern "C" long long codd_allpairs_corr_scratch(int B, int D, int h, int w) {
  long long packed = codd_conv2d_packed_size(h * w, D, 1, 1, CORR_MB, CORR_CK);
  long long pooled = 0;
  int hh = h, ww = w;
  for (int i = 1; i < 4; ++i) { hh >>= 1; ww >>= 1; pooled += (long long)D * hh * ww; }
  return (long long)B * (packed + pooled);
}

extern "C" int codd_allpairs_corr(const float* f1, const float* f2, int B, int D, int h, int w, float* lvl0,
                                  float* lvl1, float* lvl2, float* lvl3, float* scratch, void* stream) {
  if (!f1 || !f2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || !scratch || (h >> 3) < 1 || (w >> 3) < 1) return CODD_EINVAL;
  const int N = h * w;
  const long long packed = codd_conv2d_packed_size(N, D, 1, 1, CORR_MB, CORR_CK);
  float* lv[4] = {lvl0, lvl1, lvl2, lvl3};
  for (int b = 0; b < B; ++b) {
    float* wp = scratch + (size_t)b * packed;
    // weights[co = n1][ci = d] = f1[b, d, n1] / 16
    int rc = codd_conv2d_pack_weights_ex(f1 + (size_t)b * D * N, wp, N, D, 1, 1, CORR_MB, CORR_CK, 1, N, 1.f / 16.f,
                                         stream);
    if (rc) return rc;
  }
  float* pool = scratch + (size_t)B * packed;
  const float* src = f2;
  int hh = h, ww = w;
  for (int i = 0; i < 4; ++i) {
    if (i > 0) {
      int rc = codd_avgpool2(src, B * D, hh, ww, pool, stream);
      if (rc) return rc;
      src = pool;
      hh >>= 1; ww >>= 1;
      pool += (size_t)B * D * hh * ww;
    }
    for (int b = 0; b < B; ++b) {
      codd_conv_params p;
      memset(&p, 0, sizeof(p));
      p.in0.ptr = src + (size_t)b * D * hh * ww; p.in0.ctot = D; p.in0.coff = 0;
      p.C0 = D; p.C1 = 0; p.B = 1; p.Hin = hh; p.Win = ww;
      p.wpacked = scratch + (size_t)b * packed;
      p.out = lv[i] + (size_t)b * N * hh * ww; p.out_ctot = N; p.out_coff = 0;
      p.Cout = N; p.Hout = hh; p.Wout = ww;
      p.kh = p.kw = 1; p.sy = p.sx = 1; p.dil_y = p.dil_x = 1;
      p.act = CODD_ACT_NONE; p.mb = CORR_MB; p.ck = CORR_CK;
      p.npb = (hh * ww >= 4096) ? 4 : 1;
      int rc = codd_conv2d(&p, stream);
      if (rc) return rc;
    }
  }
  return CODD_OK;
}
