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
#include "conv_quad_kernel.h"
#include <stdio.h>
#include <stdlib.h>

CONV_ALL_GROUPS(CONV_DECLARE)
CONVQ_ALL(CONVQ_DECLARE)
CONVQ_MULTI(CONVQM_DECLARE)


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

// quad layout: [cog][chunk][tap][cq = c/4][co (16*mb)][4]
__global__ void conv_pack_quad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int ntaps,
                                      int mb, int ck, int nchunks, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int q = (int)(e & 3);
  long long t = e >> 2;
  const int col = (int)(t % (16 * mb)); t /= 16 * mb;
  const int cq = (int)(t % (ck >> 2)); t /= ck >> 2;
  const int tap = (int)(t % ntaps); t /= ntaps;
  const int chunk = (int)(t % nchunks);
  const int cog = (int)(t / nchunks);
  const int co = cog * 16 * mb + col, ci = chunk * ck + 4 * cq + q;
  wp[e] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * ntaps + tap] : 0.f;
}

extern "C" long long codd_conv2d_packed_size_quad(int Cout, int Cin, int kh, int kw, int mb, int ck) {
  if (mb < 1 || !(ck == 16 || ck == 32)) return -1;
  return (long long)cdiv(Cout, 16 * mb) * cdiv(Cin, ck) * (kh * kw) * ck * 16 * mb;
}

extern "C" int codd_conv2d_pack_weights_quad(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                             int ck, void* stream) {
  const long long total = codd_conv2d_packed_size_quad(Cout, Cin, kh, kw, mb, ck);
  if (total <= 0 || !w || !wpacked) return CODD_EINVAL;
  conv_pack_quad_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wpacked, Cout, Cin, kh * kw, mb, ck,
                                                                           cdiv(Cin, ck), total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
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

template <int NW, int NPB, int MB, int WREG, int IREG>
static int launch_conv(const ConvK& k, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<NW, NPB, MB, WREG, IREG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_mfma_kernel<NW, NPB, MB, WREG, IREG><<<grid, NW * 64, lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

template <int NPB, int MB>
static int launch_conv_regs(const ConvK& k, size_t lds, int grid, hipStream_t s) {
  const int wr = cdiv(k.wchunk >> 2, 256), ir = cdiv(k.nunits, 256);
  if (wr <= 4 && ir <= 4) return launch_conv<4, NPB, MB, 4, 4>(k, lds, grid, s);
  if (wr <= 4 && ir <= 8) return launch_conv<4, NPB, MB, 4, 8>(k, lds, grid, s);
  if (wr <= 12 && ir <= 4) return launch_conv<4, NPB, MB, 12, 4>(k, lds, grid, s);
  if (wr <= 12 && ir <= 8) return launch_conv<4, NPB, MB, 12, 8>(k, lds, grid, s);
  if (wr <= 16 && ir <= 8) return launch_conv<4, NPB, MB, 16, 8>(k, lds, grid, s);
  return CODD_EUNSUPPORTED;
}

// Workgroups of NW != 4 waves (NW tile rows of 16 pixels).  9: for maps where a 4-row tiling leaves the last
// round of workgroups nearly empty (72 rows x 8 column tiles x 4 channel groups = 576 blocks = 2.25 per CU with 4
// waves, exactly 256 with 9: the weights of a chunk are then fetched once per 9 rows instead of per 4).  2 and
// 8: more / fewer, smaller / larger workgroups for the autotuner (tiny maps, 144- and 288-row maps).
template <int NW, int MB>
static int launch_conv_nwx(const ConvK& k, size_t lds, int grid, hipStream_t s) {
  const int wr = cdiv(k.wchunk >> 2, NW * 64), ir = cdiv(k.nunits, NW * 64);
  if (wr <= 8 && ir <= 4) return launch_conv<NW, 1, MB, 8, 4>(k, lds, grid, s);
  if (wr <= 16 && ir <= 4) return launch_conv<NW, 1, MB, 16, 4>(k, lds, grid, s);
  return CODD_EUNSUPPORTED;
}

template <int NW, int NPB, int MB, int WREG, int QREG>
static int launch_quad(const ConvK& k, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_quad_kernel<NW, NPB, MB, WREG, QREG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_quad_kernel<NW, NPB, MB, WREG, QREG><<<grid, NW * 64, lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// quad-layout dispatch: wr / qr = float4 of weights / quad units of input per thread and chunk
static int launch_quad_any(const ConvK& k, int nw, size_t lds, int grid, hipStream_t s) {
  const codd_conv_params& p = k.p;
  const int nt = nw * 64, wr = cdiv(k.wchunk >> 2, nt), qr = cdiv((p.ck >> 2) * k.upc, nt);
#define Q(W, N, M, R, QQ) if (nw == W && p.npb == N && p.mb == M && wr <= R && qr <= QQ) return launch_quad<W, N, M, R, QQ>(k, lds, grid, s);
  Q(4, 1, 1, 8, 2) Q(4, 1, 2, 8, 1) Q(4, 1, 2, 16, 2) Q(4, 1, 4, 12, 1) Q(4, 1, 4, 16, 2)
  Q(4, 2, 1, 8, 2) Q(4, 2, 2, 8, 2) Q(4, 2, 2, 16, 2) Q(4, 2, 4, 12, 2) Q(4, 2, 4, 16, 2)
  Q(4, 4, 1, 8, 4) Q(4, 4, 2, 8, 4)
  Q(9, 1, 1, 8, 1) Q(9, 1, 2, 8, 1) Q(9, 1, 4, 8, 1) Q(9, 1, 4, 16, 1)
#undef Q
  return CODD_EUNSUPPORTED;
}

/* staging limits the host heuristics must respect: <= 16 float4 of weights and <= 8 float4 of input
 * per thread and chunk (codd_conv2d returns CODD_EUNSUPPORTED otherwise) */
int codd_conv2d_bf16(const codd_conv_params* pp, void* stream, int dry_run);  // conv_bf16.hip

extern "C" int codd_conv2d_check(const codd_conv_params* pp) {
  if (!pp || pp->layout != 2) return CODD_EINVAL;
  return codd_conv2d_bf16(pp, nullptr, 1);
}

/* launch geometry of the exact-fp32 kernels (layouts 0 and 1) for ``pp`` */
static int conv_fill(const codd_conv_params* pp, ConvK& k, size_t& lds, long long& grid, int& nw) {
  k.p = *pp;
  const codd_conv_params& p = k.p;
  if (!p.in0.ptr || !p.out || !p.wpacked || p.C0 <= 0 || p.C1 < 0 || (p.C1 > 0 && !p.in1.ptr)) return CODD_EINVAL;
  if (p.ck < 4 || (p.ck & 3) || p.B < 1 || p.Cout < 1 || p.kh < 1 || p.kw < 1 || p.pad_l < 0 || p.pad_t < 0)
    return CODD_EINVAL;
  if (!(p.mb == 1 || p.mb == 2 || p.mb == 4) || !(p.npb == 1 || p.npb == 2 || p.npb == 4)) return CODD_EINVAL;
  if (p.store_mode && (p.kh != 1 || p.kw != 1 || p.res1.ptr || p.res2.ptr || p.post.ptr)) return CODD_EUNSUPPORTED;
  k.cin = p.C0 + p.C1;
  k.nchunks = cdiv(k.cin, p.ck);
  k.ntaps = p.kh * p.kw;
  const int xb = p.npb >= 2 ? 2 : 1, rpw = p.npb / xb;
  nw = p.nw ? p.nw : 4;
  if (nw != 4 && (!(nw == 9 || nw == 2 || nw == 8) || p.npb != 1)) return CODD_EUNSUPPORTED;
  k.th = nw * rpw;
  k.tw = 16 * xb;
  k.thi = (k.th - 1) * p.sy + (p.kh - 1) * p.dil_y + 1;
  k.twi = (k.tw - 1) * p.sx + (p.kw - 1) * p.dil_x + 1;
  k.xoff = (4 - (p.pad_l % 4)) % 4;  // gx0 = tile*16k*sx - pad_l  ->  gx0 - xoff is a multiple of 4
  k.twp = ((k.xoff + k.twi + 3) / 4) * 4;
  k.twp4 = k.twp / 4;
  k.upc = k.thi * k.twp4;
  k.nunits = p.ck * k.upc;
  const int per = k.thi * k.twp;
  if (p.sx == 1) k.chs = ((per + 15) / 32) * 32 + 16;  // == 16 (mod 32): the two ci-groups of a half-wave hit disjoint banks
  else k.chs = per + 4;                                // strided: 16-byte aligned, 2-way conflicts accepted
  k.wrow = wrow_of(p.mb);
  k.wchunk = k.ntaps * p.ck * k.wrow;
  k.tiles_x = cdiv(p.Wout, k.tw);
  k.tiles_y = cdiv(p.Hout, k.th);
  k.cout_eff = p.store_mode ? 4 * p.Cout : p.Cout;
  k.ncog = cdiv(k.cout_eff, 16 * p.mb);
  const size_t hwb = (size_t)p.Hin * p.Win * sizeof(float);
  k.vec_ok = (p.Win % 4 == 0) && ((uintptr_t)p.in0.ptr % 16 == 0) && (hwb % 16 == 0) &&
             (p.C1 == 0 || (uintptr_t)p.in1.ptr % 16 == 0);
  lds = ((size_t)k.wchunk + (size_t)p.ck * k.chs) * sizeof(float);
  if (p.layout == 1) {  // quad layout: unpadded weights, input tile [cq][y][x][4]
    if (!(p.ck == 16 || p.ck == 32) || p.sx > 2 || !k.vec_ok) return CODD_EUNSUPPORTED;
    k.wchunk = k.ntaps * p.ck * 16 * p.mb;
    lds = ((size_t)k.wchunk + (size_t)p.ck * k.thi * k.twp) * sizeof(float);
  } else if (p.layout != 0) {
    return CODD_EINVAL;
  }
  if (lds > 160 * 1024) return CODD_EUNSUPPORTED;
  grid = (long long)k.tiles_x * k.tiles_y * k.ncog * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return CODD_EINVAL;
  // XCD-contiguous tile walk (conv_kernel.h conv_xcd_item): neighbouring tiles share halo rows and all weights, and the
  // dispatcher places workgroup b on XCD b % 8 -- measured +0.9 % on the frame, 2 x 60 -> 35 MB fetched per full-resolution
  // 16-channel HITNet layer (DESIGN.md finding 31)
  k.xcd = 1;
  return CODD_OK;
}

extern "C" int codd_conv2d(const codd_conv_params* pp, void* stream) {
  if (!pp) return CODD_EINVAL;
  if (pp->layout != 2 && (pp->dil2 || pp->gate)) return CODD_EUNSUPPORTED;  // dual tap sets / gate epilogues: layout 2 only
  if (pp->layout == 2) {
    const codd_conv_params& q = *pp;
    if (!q.xs || (!q.out && !q.xso) || !q.wpacked || q.C0 <= 0 || q.C1 < 0 || q.B < 1 || q.Cout < 1 || q.kh < 1 || q.kw < 1 ||
        q.pad_l < 0 || q.pad_t < 0)
      return CODD_EINVAL;
    if (q.store_mode && (q.kh != 1 || q.kw != 1 || q.res1.ptr || q.res2.ptr || q.post.ptr)) return CODD_EUNSUPPORTED;
    return codd_conv2d_bf16(pp, stream, 0);
  }
  ConvK k;
  size_t lds;
  long long grid;
  int nw;
  const int rc = conv_fill(pp, k, lds, grid, nw);
  if (rc != CODD_OK) return rc;
  const codd_conv_params& p = k.p;
  hipStream_t s = (hipStream_t)stream;
  if (p.layout == 1) return launch_quad_any(k, nw, lds, (int)grid, s);
#define CASEW(W, M) if (nw == W && p.mb == M) return launch_conv_nwx<W, M>(k, lds, (int)grid, s)
  CASEW(9, 1); CASEW(9, 2); CASEW(9, 4); CASEW(2, 1); CASEW(2, 2); CASEW(2, 4); CASEW(8, 1); CASEW(8, 2); CASEW(8, 4);
#undef CASEW
#define CASE(N, M) if (p.npb == N && p.mb == M) return launch_conv_regs<N, M>(k, lds, (int)grid, s)
  CASE(1, 1); CASE(1, 2); CASE(1, 4);
  CASE(2, 1); CASE(2, 2); CASE(2, 4);
  CASE(4, 1); CASE(4, 2); CASE(4, 4);
#undef CASE
  return CODD_EUNSUPPORTED;
}

template <int NW, int NPB, int MB, int WREG, int QREG>
static int launch_quad_multi(const ConvKN& kn, size_t lds, int grid, hipStream_t s) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_quad_multi_kernel<NW, NPB, MB, WREG, QREG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_quad_multi_kernel<NW, NPB, MB, WREG, QREG><<<grid, NW * 64, lds, s>>>(kn);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

/* n <= 4 independent quad-layout convolutions (layout 1, 4-wave workgroups, 4 x 16 tiles: npb = 1; the same mb = 1 | 2
 * for all) as ONE launch; every job is validated as codd_conv2d would.  CODD_EUNSUPPORTED when a job does not fit the
 * instantiated register classes -- the caller then launches the jobs one by one. */
extern "C" int codd_conv2d_multi(const codd_conv_params* ps, int n, void* stream) {
  if (!ps || n < 1 || n > CONVQ_MULTI_MAX) return CODD_EINVAL;
  if (n == 1) return codd_conv2d(ps, stream);
  ConvKN kn;
  memset(&kn, 0, sizeof(kn));
  size_t lds = 0;
  long long total = 0;
  int wr = 0, qr = 0;
  for (int i = 0; i < n; ++i) {
    size_t l;
    long long g;
    int nw;
    if (ps[i].layout != 1) return CODD_EUNSUPPORTED;
    const int rc = conv_fill(&ps[i], kn.k[i], l, g, nw);
    if (rc != CODD_OK) return rc;
    const codd_conv_params& p = kn.k[i].p;
    if (nw != 4 || p.npb != 1 || p.mb != ps[0].mb || !(p.mb == 1 || p.mb == 2)) return CODD_EUNSUPPORTED;
    const int w_ = cdiv(kn.k[i].wchunk >> 2, 256), q_ = cdiv((p.ck >> 2) * kn.k[i].upc, 256);
    wr = w_ > wr ? w_ : wr;
    qr = q_ > qr ? q_ : qr;
    lds = l > lds ? l : lds;
    kn.start[i] = (int)total;
    total += g;
  }
  for (int i = n; i <= CONVQ_MULTI_MAX; ++i) kn.start[i] = (int)total;  // jobs past n own no workgroup
  if (total > 0x7fffffffLL) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (ps[0].mb == 1 && wr <= 8 && qr <= 2) return launch_quad_multi<4, 1, 1, 8, 2>(kn, lds, (int)total, s);
  if (ps[0].mb == 2 && wr <= 16 && qr <= 2) return launch_quad_multi<4, 1, 2, 16, 2>(kn, lds, (int)total, s);
  return CODD_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// All-pairs correlation pyramid (reference blocks/corr.py:28-45,56-62) as four 1x1 "convolutions":
// output channel = source pixel n1 (weights = f1^T / 16, re-packed every frame), image = f2 pooled
// i times.  avg_pool2d commutes with the inner product, so level i never reads level i-1.
// ------------------------------------------------------------------------------------------------
extern "C" int codd_avgpool2(const float* in, int BC, int h, int w, float* out, void* stream);

#define CORR_MB 4   // 16-channel blocks per workgroup of the all-pairs GEMM
#define CORR_CK 32  // its chunk depth
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
      p.npb = 1;  // measured: 4x16-pixel tiles 559 us vs 890 us with 8x32 tiles for the whole pyramid
      int rc = codd_conv2d(&p, stream);
      if (rc) return rc;
    }
  }
  return CODD_OK;
}
