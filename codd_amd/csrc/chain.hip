// LDS-resident convolution chains (include/codd_hip.h, codd_conv_chain): several small stride-1 convolutions in ONE
// launch, intermediates in LDS, halo recomputed per tile.  Exact fp32 on v_mfma_f32_16x16x4_f32:
//   A operand (16 x 4) = weights : lane (j = l & 15, g = l >> 4) holds W[co = 16 c + j][ci = 4 ks + g]  (pre-packed)
//   B operand (4 x 16) = pixels  : lane (j, g) holds X[ci = 4 ks + g][pixel j of the block]
//   D (16 x 16)                  : lane holds D[co = 16 c + 4 g + r][pixel j], r = 0..3
// A layer's output region is walked as a FLAT list of pixels, 16 per block (no row padding: a 22-pixel-wide halo
// region wastes nothing), two blocks per wave pass so that every weight fragment feeds two MFMAs.
// A wave's k-loop is a serial chain (operand read -> MFMA), so it is built to need few, wide LDS reads and to have the
// next operands in flight under the current MFMAs: both operands live in LDS as 8-byte PAIRS of consecutive k-steps --
//   activations [channel octet q][region pixel][8]: channel 8q + 4t + g at float 2g + t  (lane (j, g) reads k-steps
//                                                   2q, 2q + 1 of its pixel with one ds_read_b64)
//   weights     [tap][q][16-channel block][lane][2]
// -- and the (tap, q) loop is software-pipelined one step ahead.  The next layer's weight block is fetched into
// registers while the current layer computes (issue early, commit late), so its L2 latency hides behind the MFMAs.
#include "common.h"
#include <string.h>

constexpr int CHAIN_NW = 8;             // waves per workgroup (two per SIMD)
constexpr int CHAIN_NT = CHAIN_NW * 64;
constexpr int CHAIN_WR = 8;             // float4 of the next layer's weights held per thread (<= 64 KiB per layer)
constexpr int CHAIN_PB = 2;             // pixel blocks per wave pass

struct ChainK {
  codd_chain_params p;
  int halo, rh, rw;  // total halo; region = (th + 2 halo) x (tw + 2 halo)
  int cs, bufsz;     // region pixels / size of one LDS buffer (floats)
  int cb;            // channels per LDS buffer
  int margin[CODD_CHAIN_MAX_LAYERS];  // margin of layer l's output region inside the region frame
  int wsz[CODD_CHAIN_MAX_LAYERS];     // floats of layer l's packed block
  int tiles_x, tiles_y;
  int stage_in;      // first layer is a 3x3: the input tile is staged into buffer p.stage
};

static inline int chain_cob(int cout) { return (cout + 15) / 16; }
static inline int chain_nq(int cin) { return (cin + 7) / 8; }  // channel octets = pairs of k-steps

extern "C" long long codd_chain_layer_size(int cout, int cin, int k) {
  if (cout < 1 || cout > 48 || cin < 1 || cin > 64 || !(k == 1 || k == 3)) return -1;
  return (long long)k * k * chain_nq(cin) * chain_cob(cout) * 128 + 16 * chain_cob(cout);
}

__global__ void chain_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin, int taps,
                                  int nq, int cob, float* __restrict__ dst, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int nw = taps * nq * cob * 128;
  if (e >= nw) {  // bias, zero-padded to the channel blocks
    const int co = e - nw;
    dst[e] = (bias && co < cout) ? bias[co] : 0.f;
    return;
  }
  const int t2 = e & 1, lane = (e >> 1) & 63;
  int t = e >> 7;
  const int c = t % cob; t /= cob;
  const int q = t % nq;
  const int tap = t / nq;
  const int co = 16 * c + (lane & 15), ci = 8 * q + 4 * t2 + (lane >> 4);
  dst[e] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] : 0.f;
}

extern "C" int codd_chain_pack_layer(const float* w, const float* bias, int cout, int cin, int k, float* dst,
                                     void* stream) {
  const long long total = codd_chain_layer_size(cout, cin, k);
  if (total <= 0 || !w || !dst) return CODD_EINVAL;
  chain_pack_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, bias, cout, cin, k * k, chain_nq(cin),
                                                                       chain_cob(cout), dst, (int)total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

__device__ __forceinline__ const float* chain_view(const codd_view& v, int b, int c, int hw) {
  return v.ptr + ((size_t)b * v.ctot + v.coff + c) * (size_t)hw;
}

// float offset of channel ``co`` of region pixel ``pos`` in an LDS activation buffer of ``npos`` pixels
__device__ __forceinline__ int chain_at(int co, int pos, int npos) {
  return (((co >> 3) * npos + pos) << 3) + ((co & 3) << 1) + ((co >> 2) & 1);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One layer of the chain for the calling workgroup.  SRCG: the layer reads the chain input from global memory (1x1 first
// layer); DSTG: it writes the chain output to global memory (last layer).
template <int COB>
__device__ __forceinline__ void chain_layer(const ChainK& k, const int l, const float* __restrict__ src,
                                            float* __restrict__ dst, const float* __restrict__ resb,
                                            const float* __restrict__ wl, const int b, const int gy0, const int gx0,
                                            const bool SRCG, const bool DSTG) {
  const codd_chain_params& p = k.p;
  const codd_chain_layer& L = p.layer[l];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int m = k.margin[l];
  const int Hl = k.rh - 2 * m, Wl = k.rw - 2 * m, npx = Hl * Wl;
  const int nblk = (npx + 15) >> 4;
  const int nq = (L.cin + 7) >> 3, taps = L.k * L.k;
  const int cpad = ((L.cout + 7) >> 3) << 3;
  const float* biasl = wl + taps * nq * COB * 128;
  const int hw = p.H * p.W, npos = k.rh * k.rw;

  for (int blk0 = wave * CHAIN_PB; blk0 < nblk; blk0 += CHAIN_NW * CHAIN_PB) {
    int pos[CHAIN_PB], gpix[CHAIN_PB];
    bool ok[CHAIN_PB], inside[CHAIN_PB];
#pragma unroll
    for (int q = 0; q < CHAIN_PB; ++q) {
      int f = (blk0 + q) * 16 + j;
      ok[q] = f < npx;
      f = ok[q] ? f : npx - 1;
      const int yy = f / Wl, xx = f - yy * Wl;
      pos[q] = (m + yy) * k.rw + m + xx;
      const int gy = gy0 + m + yy, gx = gx0 + m + xx;
      inside[q] = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      gpix[q] = inside[q] ? gy * p.W + gx : 0;
    }
    f32x4 acc[CHAIN_PB][COB];
#pragma unroll
    for (int q = 0; q < CHAIN_PB; ++q)
#pragma unroll
      for (int c = 0; c < COB; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (SRCG) {  // 1x1 straight from global memory: X[ci][pixel], zero outside the image / past cin
      // the loads of two channel octets (four k-steps) are issued before their MFMAs: one L2 round trip per four
      // k-steps instead of one per k-step
      const f32x2* wp = (const f32x2*)wl + lane;
      constexpr int QU = 2;
      for (int q0 = 0; q0 < nq; q0 += QU) {
        float bv[QU][2][CHAIN_PB];
#pragma unroll
        for (int u = 0; u < QU; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ci = 8 * (q0 + u) + 4 * t + g;
            const bool cok = ci < L.cin;  // (also false for octets past nq)
            const int cc = cok ? ci : 0;
            const float* sp = cc < p.C0 ? chain_view(p.in0, b, cc, hw) : chain_view(p.in1, b, cc - p.C0, hw);
#pragma unroll
            for (int q = 0; q < CHAIN_PB; ++q) {
              const float v = sp[gpix[q]];  // in-bounds address even when masked
              bv[u][t][q] = (cok && inside[q]) ? v : 0.f;
            }
          }
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          if (q0 + u < nq) {
#pragma unroll
            for (int c = 0; c < COB; ++c) {
              const f32x2 av = wp[((q0 + u) * COB + c) * 64];
#pragma unroll
              for (int q = 0; q < CHAIN_PB; ++q) {
                acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[u][0][q], acc[q][c], 0, 0, 0);
                acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[u][1][q], acc[q][c], 0, 0, 0);
              }
            }
          }
        }
      }
    } else {
      // flattened (tap, octet) loop, operands of step it + 1 in flight while the MFMAs of step it issue
      const int hk = L.k >> 1, nit = taps * nq;
      const f32x2* wp = (const f32x2*)wl + lane;
      const f32x2* sb[CHAIN_PB];
#pragma unroll
      for (int q = 0; q < CHAIN_PB; ++q) sb[q] = (const f32x2*)src + pos[q] * 4 + g;  // + (octet * npos + tap offset) * 4
      f32x2 a0[COB], b0[CHAIN_PB], a1[COB], b1[CHAIN_PB];
      int tq = 0, ty = 0, tx = 0;
      int toff = ((0 - hk) * k.rw + (0 - hk)) * L.dil;
#define CHAIN_LOAD(A, B, IT)                                                                   \
  {                                                                                            \
    const int o_ = (tq * npos + toff) * 4;                                                     \
    _Pragma("unroll") for (int q = 0; q < CHAIN_PB; ++q) B[q] = sb[q][o_];                     \
    _Pragma("unroll") for (int c = 0; c < COB; ++c) A[c] = wp[((IT) * COB + c) * 64];          \
    if (++tq == nq) {                                                                          \
      tq = 0;                                                                                  \
      if (++tx == L.k) { tx = 0; ++ty; }                                                       \
      toff = ((ty - hk) * k.rw + (tx - hk)) * L.dil;                                           \
    }                                                                                          \
  }
  // (the two k-steps of a pair go to the same accumulator: all first halves, then all second halves, so that
  // dependent MFMAs are COB * PB instructions apart)
#define CHAIN_MFMA(A, B)                                                                       \
  _Pragma("unroll") for (int c = 0; c < COB; ++c) _Pragma("unroll") for (int q = 0; q < CHAIN_PB; ++q)      \
    acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].x, B[q].x, acc[q][c], 0, 0, 0);      \
  _Pragma("unroll") for (int c = 0; c < COB; ++c) _Pragma("unroll") for (int q = 0; q < CHAIN_PB; ++q)      \
    acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].y, B[q].y, acc[q][c], 0, 0, 0);
      CHAIN_LOAD(a0, b0, 0);
      int it = 0;
      for (; it + 2 <= nit - 1; it += 2) {
        CHAIN_LOAD(a1, b1, it + 1);
        CHAIN_MFMA(a0, b0);
        CHAIN_LOAD(a0, b0, it + 2);
        CHAIN_MFMA(a1, b1);
      }
      if (it + 1 < nit) {  // two steps left
        CHAIN_LOAD(a1, b1, it + 1);
        CHAIN_MFMA(a0, b0);
        CHAIN_MFMA(a1, b1);
      } else {
        CHAIN_MFMA(a0, b0);
      }
#undef CHAIN_LOAD
#undef CHAIN_MFMA
    }

    // epilogue: bias, residual, activation; intermediates are zero outside the image
    if (DSTG) {
      // the global residual operand of a pixel block is fetched BEFORE the block's first store (unconditional, clamped
      // addresses): interleaved with the stores every load would be a serialised round trip
#pragma unroll
      for (int q = 0; q < CHAIN_PB; ++q) {
        float rv[COB][4];
#pragma unroll
        for (int c = 0; c < COB; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * c + 4 * g + r;
            rv[c][r] = 0.f;
            if (p.res1.ptr) rv[c][r] = chain_view(p.res1, b, co < p.cout_store ? co : p.cout_store - 1, hw)[gpix[q]];
          }
        if (!ok[q] || !inside[q]) continue;
#pragma unroll
        for (int c = 0; c < COB; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * c + 4 * g + r;
            if (co >= p.cout_store) continue;
            float v = acc[q][c][r] + biasl[co] + rv[c][r];
            if (resb) v += resb[chain_at(co, pos[q], npos)];
            v = act_apply(v, L.act, co);
            p.out[((size_t)b * p.out_ctot + p.out_coff + co) * (size_t)hw + gpix[q]] = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < CHAIN_PB; ++q) {
        if (!ok[q]) continue;
#pragma unroll
        for (int c = 0; c < COB; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * c + 4 * g + r;
            if (co >= cpad) continue;
            float v = acc[q][c][r] + biasl[co];
            const int at = chain_at(co, pos[q], npos);
            if (resb) v += resb[at];
            v = act_apply(v, L.act, co);
            dst[at] = inside[q] ? v : 0.f;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(CHAIN_NT) void conv_chain_kernel(const ChainK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const codd_chain_params& p = k.p;
  float* const buf0 = smem;
  float* const buf1 = smem + k.bufsz;
  float* wl = smem + 2 * k.bufsz;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y;
  const int b = bid / k.tiles_y;
  const int gy0 = ty * p.th - k.halo, gx0 = tx * p.tw - k.halo;  // image coordinates of the region origin
  const int hw = p.H * p.W;

  // next layer's weight block: global -> registers (ext-vector loads: a HIP float4 struct array would live in scratch)
  f32x4 wreg[CHAIN_WR];
#define CHAIN_ISSUE(LL)                                                              \
  {                                                                                  \
    const f32x4* s_ = (const f32x4*)(p.wpacked + p.layer[LL].wofs);                  \
    const int n4_ = k.wsz[LL] >> 2;                                                  \
    _Pragma("unroll") for (int r = 0; r < CHAIN_WR; ++r) {                           \
      const int e = tid + r * CHAIN_NT;                                              \
      wreg[r] = s_[e < n4_ ? e : n4_ - 1]; /* unconditional load, masked at commit */ \
    }                                                                                \
  }
#define CHAIN_COMMIT(LL)                                                             \
  {                                                                                  \
    f32x4* d_ = (f32x4*)wl;                                                          \
    const int n4_ = k.wsz[LL] >> 2;                                                  \
    _Pragma("unroll") for (int r = 0; r < CHAIN_WR; ++r) {                           \
      const int e = tid + r * CHAIN_NT;                                              \
      if (e < n4_) d_[e] = wreg[r];                                                  \
    }                                                                                \
  }

  CHAIN_ISSUE(0);
  if (k.stage_in) {  // chain input tile -> LDS buffer (channels padded to a multiple of 8 with zeros)
    // four independent, UNCONDITIONAL loads per thread and trip (clamped addresses, masked values): a conditional
    // load makes hipcc wait for each element in turn -- 20 serialised L2 round trips per thread on a 32-channel tile
    float* d = p.stage ? buf1 : buf0;
    const int cin = p.C0 + p.C1, c8 = ((cin + 7) >> 3) << 3, n = k.rh * k.rw, total = c8 * n;
    constexpr int SU = 4;
    for (int e0 = tid; e0 < total; e0 += CHAIN_NT * SU) {
      float v[SU];
      int at[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        int e = e0 + u * CHAIN_NT;
        const bool live = e < total;
        e = live ? e : total - 1;
        const int c = e / n, pos = e - c * n;
        const int y = pos / k.rw, x = pos - y * k.rw;
        const int gy = gy0 + y, gx = gx0 + x;
        const bool ok = live && c < cin && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cc = c < cin ? c : cin - 1;
        const int gyc = min(max(gy, 0), p.H - 1), gxc = min(max(gx, 0), p.W - 1);
        const float* sp = (cc < p.C0 ? chain_view(p.in0, b, cc, hw) : chain_view(p.in1, b, cc - p.C0, hw)) + gyc * p.W + gxc;
        const float t = *sp;
        v[u] = ok ? t : 0.f;
        at[u] = live ? chain_at(c, pos, n) : -1;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if (at[u] >= 0) d[at[u]] = v[u];
    }
  }
  for (int l = 0; l < p.nlayers; ++l) {
    const codd_chain_layer& L = p.layer[l];
    CHAIN_COMMIT(l);
    __syncthreads();
    if (l + 1 < p.nlayers) CHAIN_ISSUE(l + 1);
    const int si = L.src >= 0 ? L.src : (k.stage_in ? p.stage : -1);
    const float* src = si < 0 ? nullptr : (si ? buf1 : buf0);
    float* dst = L.dst < 0 ? nullptr : (L.dst ? buf1 : buf0);
    const float* resb = L.res < 0 ? nullptr : (L.res ? buf1 : buf0);
    const bool srcg = (l == 0 && !k.stage_in), dstg = L.dst < 0;
    const int cob = (L.cout + 15) >> 4;
#define CHAIN_CASE(C) case C: chain_layer<C>(k, l, src, dst, resb, wl, b, gy0, gx0, srcg, dstg); break;
    switch (cob) { CHAIN_CASE(1) CHAIN_CASE(2) CHAIN_CASE(3) }  // cout <= 48
#undef CHAIN_CASE
    __syncthreads();
  }
}

static int chain_geometry(const codd_chain_params* pp, ChainK& k, size_t& lds, long long& grid) {
  if (!pp) return CODD_EINVAL;
  k.p = *pp;
  const codd_chain_params& p = k.p;
  if (p.nlayers < 1 || p.nlayers > CODD_CHAIN_MAX_LAYERS || p.B < 1 || p.H < 1 || p.W < 1 || p.th < 1 || p.tw < 1 ||
      p.C0 < 1 || p.C1 < 0)
    return CODD_EINVAL;
  int halo = 0, cb = 8, wmax = 0;
  for (int l = 0; l < p.nlayers; ++l) {
    const codd_chain_layer& L = p.layer[l];
    const long long sz = codd_chain_layer_size(L.cout, L.cin, L.k);
    if (sz <= 0 || L.dil < 1 || (L.wofs & 3) || L.wofs < 0) return CODD_EINVAL;
    const bool first = l == 0, last = l == p.nlayers - 1;
    if ((L.src < 0) != first || (L.dst < 0) != last || L.src > 1 || L.dst > 1 || L.res > 1) return CODD_EINVAL;
    if (!first && p.layer[l - 1].cout != L.cin) return CODD_EINVAL;
    if (L.src >= 0 && L.src == L.dst) return CODD_EUNSUPPORTED;  // a 3x3 cannot run in place
    k.wsz[l] = (int)sz;
    wmax = sz > wmax ? (int)sz : wmax;
    halo += L.dil * (L.k / 2);
    k.margin[l] = halo;
    if (!last) cb = ((L.cout + 7) / 8 * 8) > cb ? (L.cout + 7) / 8 * 8 : cb;
  }
  if (p.layer[0].cin != p.C0 + p.C1 || (p.C1 > 0 && !p.in1.ptr)) return CODD_EINVAL;
  k.stage_in = p.layer[0].k != 1;
  if (k.stage_in) {
    if (p.stage < 0 || p.stage > 1 || p.stage == p.layer[0].dst) return CODD_EINVAL;
    const int c8 = (p.C0 + p.C1 + 7) / 8 * 8;
    cb = c8 > cb ? c8 : cb;
  }
  if (p.cout_store < 1 || p.cout_store > p.layer[p.nlayers - 1].cout) return CODD_EINVAL;
  if (wmax > CHAIN_WR * CHAIN_NT * 4) return CODD_EUNSUPPORTED;
  k.halo = halo;
  k.rh = p.th + 2 * halo;
  k.rw = p.tw + 2 * halo;
  const int n = k.rh * k.rw;
  k.cs = n;  // (a channel OCTET occupies 8 * n floats)
  k.cb = cb;
  k.bufsz = cb * k.cs;
  lds = ((size_t)2 * k.bufsz + wmax) * sizeof(float);
  if (lds > 160 * 1024) return CODD_EUNSUPPORTED;
  k.tiles_x = cdiv(p.W, p.tw);
  k.tiles_y = cdiv(p.H, p.th);
  grid = (long long)k.tiles_x * k.tiles_y * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return CODD_EINVAL;
  return CODD_OK;
}

extern "C" int codd_conv_chain_check(const codd_chain_params* pp) {
  ChainK k;
  size_t lds;
  long long grid;
  return chain_geometry(pp, k, lds, grid);
}

extern "C" int codd_conv_chain(const codd_chain_params* pp, void* stream) {
  ChainK k;
  size_t lds;
  long long grid;
  const int rc = chain_geometry(pp, k, lds, grid);
  if (rc != CODD_OK) return rc;
  if (!k.p.in0.ptr || !k.p.wpacked || !k.p.out) return CODD_EINVAL;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_chain_kernel<<<(int)grid, CHAIN_NT, lds, (hipStream_t)stream>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
