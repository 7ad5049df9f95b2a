// HITNetMF non-convolution kernels: tile cost volume + arg-min, slanted-plane local correlation,
// hypothesis plane up-sampling and selection.  All HBM/LDS-bound fp32 work (no MFMA).
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// Tile cost volume + first arg-min (reference initialization.py:18-45,167-183), never materialised.
// Workgroup = 256 threads = 4 waves, handles TPB = 16 consecutive tiles of one tile row.  The right
// feature window [4*x0 - (D-1), 4*(x0+15)] x C is staged in LDS once (coalesced), each wave then
// scans 4 tiles with lane = disparity (d = lane, lane+64, ...): LDS reads are stride-1 across
// lanes, the channel sum is sequential (c = 0..C-1), and the arg-min is a wave shuffle reduction
// that keeps the LOWEST index among equal costs (torch.min returns the first minimum).
// ------------------------------------------------------------------------------------------------
#define CV_TPB 16
__global__ __launch_bounds__(256) void costvol_argmin_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                             int C, int Ht, int Wt, int Wr, int D, float* cost,
                                                             int cost_ctot, int cost_coff, float* disp, int disp_ctot,
                                                             int disp_coff, int zero_dxdy) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xt0 = blockIdx.x * CV_TPB, y = blockIdx.y, b = blockIdx.z;
  const int win = D + 4 * (CV_TPB - 1);  // window length per channel
  const int base = 4 * xt0 - (D - 1);    // image x of window element 0
  float* rl = sm;                        // [C][win]
  float* ll = sm + C * win;              // [C][CV_TPB]
  const float* Rb = R + ((size_t)b * C * Ht + y) * Wr;
  for (int e = tid; e < C * win; e += 256) {
    const int c = e / win, i = e - c * win, gx = base + i;
    rl[e] = (gx >= 0 && gx < Wr) ? Rb[(size_t)c * Ht * Wr + gx] : 0.f;
  }
  const float* Lb = L + ((size_t)b * C * Ht + y) * Wt;
  for (int e = tid; e < C * CV_TPB; e += 256) {
    const int c = e / CV_TPB, i = e - c * CV_TPB, gx = xt0 + i;
    ll[e] = gx < Wt ? Lb[(size_t)c * Ht * Wt + gx] : 0.f;
  }
  __syncthreads();
  for (int t = wave; t < CV_TPB; t += 4) {
    const int xt = xt0 + t;
    if (xt >= Wt) break;
    float best = INFINITY;
    int bi = 0x7fffffff;
    // window index of R~[4*xt - d] = 4*t + (D-1) - d
    const int o = 4 * t + (D - 1);
    for (int d = lane; d < D; d += 64) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += fabsf(ll[c * CV_TPB + t] - rl[c * win + o - d]);
      if (s < best) { best = s; bi = d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      const size_t hw = (size_t)Ht * Wt, pix = (size_t)y * Wt + xt;
      cost[((size_t)b * cost_ctot + cost_coff) * hw + pix] = best;
      float* dp = disp + ((size_t)b * disp_ctot + disp_coff) * hw + pix;
      dp[0] = (float)bi;
      if (zero_dxdy) { dp[hw] = 0.f; dp[2 * hw] = 0.f; }
    }
  }
}

// D % 4 == 0 (every level of the benchmarked configurations): a lane owns FOUR consecutive disparities, so the right
// window is read as aligned 16-byte LDS vectors (4 values per ds_read_b128 instead of 1 per ds_read_b32) and a wave
// keeps the left features of its 4 consecutive tiles in one vector -- the scalar form issues 2 LDS reads per
// |l - r| term and is LDS-issue bound (68 us at the finest level).  Same arithmetic per (tile, d): the channel sum is
// sequential, the first minimum wins.
template <int TPB>
__global__ __launch_bounds__(256) void costvol_argmin4_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                              int C, int Ht, int Wt, int Wr, int D, float* cost,
                                                              int cost_ctot, int cost_coff, float* disp, int disp_ctot,
                                                              int disp_coff, int zero_dxdy) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx_, y, b;  // x-neighbours share all but four columns of their right-feature windows: keep them on one XCD's L2
  codd_xcd_block(bx_, y, b);
  const int xt0 = bx_ * TPB;
  const int win = D + 4 * (TPB - 1);  // window length per channel (multiple of 4)
  const int base = 4 * xt0 - (D - 1); // image x of window element 0
  float* rl = sm;                     // [C][win]
  float* ll = sm + C * win;           // [C][TPB]
  // TPB tiles share one staged window: D + 4 (TPB - 1) columns for 4 TPB new ones -- 16 tiles re-read every right
  // feature 6x at the finest level (D = 320), 64 tiles 2.2x
  const float* Rb = R + ((size_t)b * C * Ht + y) * Wr;
  for (int c = wave; c < C; c += 4) {
    const float* rc = Rb + (size_t)c * Ht * Wr;
    for (int i = lane; i < win; i += 64) {
      const int gx = base + i;
      rl[c * win + i] = (gx >= 0 && gx < Wr) ? rc[gx] : 0.f;
    }
  }
  const float* Lb = L + ((size_t)b * C * Ht + y) * Wt;
  for (int e = tid; e < C * TPB; e += 256) {
    const int c = e / TPB, i = e - c * TPB, gx = xt0 + i;
    ll[e] = gx < Wt ? Lb[(size_t)c * Ht * Wt + gx] : 0.f;
  }
  __syncthreads();
  for (int t0 = 4 * wave; t0 < TPB && xt0 + t0 < Wt; t0 += 16) {  // this wave's tiles: t0 .. t0 + 3
    float best[4];
    int bi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { best[q] = INFINITY; bi[q] = 0x7fffffff; }
    for (int d0 = 4 * lane; d0 < D; d0 += 256) {
      float s[4][4];  // [tile][k]: disparity d0 + k
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[q][k] = 0.f;
      // window index of R~[4*xt - d] = 4*t + (D-1) - d; d = d0 + 3 .. d0 sit at 4*t + D - 4 - d0 .. + 3 (aligned)
      const int o = 4 * t0 + D - 4 - d0;
      for (int c = 0; c < C; ++c) {
        const float4 l4 = *(const float4*)(ll + c * TPB + t0);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 r4 = *(const float4*)(rl + c * win + o + 4 * q);
          s[q][3] += fabsf(lv[q] - r4.x);
          s[q][2] += fabsf(lv[q] - r4.y);
          s[q][1] += fabsf(lv[q] - r4.z);
          s[q][0] += fabsf(lv[q] - r4.w);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (s[q][k] < best[q]) { best[q] = s[q][k]; bi[q] = d0 + k; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float bb = best[q];
      int ii = bi[q];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(bb, off, 64);
        const int oi = __shfl_xor(ii, off, 64);
        if (ob < bb || (ob == bb && oi < ii)) { bb = ob; ii = oi; }
      }
      const int xt = xt0 + t0 + q;
      if (lane == 0 && xt < Wt) {
        const size_t hw = (size_t)Ht * Wt, pix = (size_t)y * Wt + xt;
        cost[((size_t)b * cost_ctot + cost_coff) * hw + pix] = bb;
        float* dp = disp + ((size_t)b * disp_ctot + disp_coff) * hw + pix;
        dp[0] = (float)ii;
        if (zero_dxdy) { dp[hw] = 0.f; dp[2 * hw] = 0.f; }
      }
    }
  }
}

extern "C" int codd_tile_costvol_argmin(const float* L, const float* R, int B, int C, int Ht, int Wt, int Wr, int D,
                                        float* cost, int cost_ctot, int cost_coff, float* disp, int disp_ctot,
                                        int disp_coff, int zero_dxdy, void* stream) {
  if (!L || !R || !cost || !disp || B < 1 || C < 1 || D < 1 || Ht < 1 || Wt < 1) return CODD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if ((D & 3) == 0) {
    // tiles per workgroup: 16.  More tiles share more of the staged window (64 tiles re-read a right feature 2.2x
    // instead of 6x at D = 320) but measured SLOWER at the finest level -- 40.5 (16) / 43.8 (32) / 55.2 us (64): the
    // kernel is a stage-then-scan latency chain per workgroup, and more, smaller workgroups overlap it better
    int tpb = 16;
    for (;; tpb >>= 1) {
      const int win = D + 4 * (tpb - 1);
      const size_t lds = (size_t)(C * win + C * tpb) * sizeof(float);
      if (lds > 64 * 1024) { if (tpb == 16) return CODD_EUNSUPPORTED; continue; }
      dim3 grid(cdiv(Wt, tpb), Ht, B);
#define CV_LAUNCH(T) costvol_argmin4_kernel<T><<<grid, 256, lds, st>>>(L, R, C, Ht, Wt, Wr, D, cost, cost_ctot, cost_coff, disp, disp_ctot, disp_coff, zero_dxdy)
      if (tpb == 64) CV_LAUNCH(64); else if (tpb == 32) CV_LAUNCH(32); else CV_LAUNCH(16);
#undef CV_LAUNCH
      break;
    }
    CODD_LAUNCH_CHECK();
    return CODD_OK;
  }
  const int win = D + 4 * (CV_TPB - 1);
  size_t lds = (size_t)(C * win + C * CV_TPB) * sizeof(float);
  if (lds > 64 * 1024) return CODD_EUNSUPPORTED;
  dim3 grid(cdiv(Wt, CV_TPB), Ht, B);
  costvol_argmin_kernel<<<grid, 256, lds, st>>>(L, R, C, Ht, Wt, Wr, D, cost, cost_ctot, cost_coff,
                                                disp, disp_ctot, disp_coff, zero_dxdy);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// TileWarping (reference propagation.py:61-86) + ||fea_l||_1 unshuffle (:157,207).
// Thread = (tile tx, feature row y = 4*ty + iy); it walks the 4 in-tile columns.  The three
// disparity offsets k = -1,0,1 share one fractional weight (xs_k = xs_0 - k), so 4 taps of the
// right row serve all three.  Lanes are consecutive tiles -> the 64 output channels are written
// fully coalesced; the left row is read as float4.
// ------------------------------------------------------------------------------------------------
// STAGE: the workgroup first copies the right feature row (C x W floats, coalesced 16-byte loads) into LDS and gathers
// from there: the 16 taps per channel and hypothesis set are 4-byte gathers 16 bytes apart between neighbouring lanes,
// i.e. a quarter of every fetched line is used and the texture addresser sees 32 scattered loads per channel; from
// LDS they cost a ds_read each.  A workgroup covers (up to) a whole tile row, so the row is read once.
template <bool STAGE>
__global__ __launch_bounds__(256) void tile_warp_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                        int C, int Ht, int Wt, codd_view h0, codd_view h1, int nhyp,
                                                        float* out0, float* out1) {
  extern __shared__ __attribute__((aligned(16))) float srow[];  // STAGE: [C][W]
  const int txr = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y;  // feature row
  const int b = blockIdx.z;
  if (!STAGE && txr >= Wt) return;
  const bool live = txr < Wt;
  const int tx = live ? txr : Wt - 1;
  const int ty = yy >> 2, iy = yy & 3;
  const int H = 4 * Ht, W = 4 * Wt;
  if (STAGE) {
    const float4* src = (const float4*)(fr + (size_t)b * C * H * W + (size_t)yy * W);
    const int w4 = W >> 2, chw4 = (H * W) >> 2;
    for (int c = 0; c < C; ++c)
      for (int x4 = threadIdx.x; x4 < w4; x4 += blockDim.x) ((float4*)srow)[c * w4 + x4] = src[(size_t)c * chw4 + x4];
    __syncthreads();
  }
  const size_t thw = (size_t)Ht * Wt, tpix = (size_t)ty * Wt + tx;
  const float* flb = fl + (size_t)b * C * H * W + (size_t)yy * W + 4 * tx;
  const float* frb = fr + (size_t)b * C * H * W + (size_t)yy * W;
  const size_t chw = (size_t)H * W;

  // Both hypothesis sets in ONE pass over the channels (they share the left-feature load), every gather of a channel
  // issued unconditionally (clamped address, masked value) and two channels per trip: the loop used to be a chain of
  // C dependent L2 round trips per hypothesis set (~27 us at every coarse level whatever its size).
  int f0[2][4];
  float a[2][4];
  unsigned okm[2] = {0u, 0u};  // bit ix * 4 + q: tap q of sub-pixel ix lies inside the row
#pragma unroll
  for (int hsel = 0; hsel < 2; ++hsel) {
    const codd_view hv = (hsel && nhyp == 2) ? h1 : h0;
    const float* hp = hv.ptr + ((size_t)b * hv.ctot + hv.coff) * thw + tpix;
    const float d = hp[0], dx = hp[thw], dy = hp[2 * thw];
#pragma unroll
    for (int ix = 0; ix < 4; ++ix) {
      const float delta = d + (ix - 1.5f) * dx + (iy - 1.5f) * dy;
      const float xs = (float)(4 * tx + ix) - delta;  // k = 0
      const float fl0 = floorf(xs);
      a[hsel][ix] = xs - fl0;
      // clamp far-away samples so that the int conversion is defined; all 4 taps are then OOB
      f0[hsel][ix] = (int)fminf(fmaxf(fl0, -4.f), (float)W + 4.f);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((unsigned)(f0[hsel][ix] - 1 + q) < (unsigned)W) okm[hsel] |= 1u << (ix * 4 + q);
    }
  }
  float fea[4] = {0.f, 0.f, 0.f, 0.f};
  float cv[2][3][4] = {{{0.f}}};
#pragma unroll 2
  for (int c = 0; c < C; ++c) {
    const float4 l4 = *(const float4*)(flb + c * chw);
    const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
    const float* rr = frb + c * chw;
    float t[2][4][4];
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      if (hsel < nhyp) {
#pragma unroll
        for (int ix = 0; ix < 4; ++ix)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int xi = min(max(f0[hsel][ix] - 1 + q, 0), W - 1);
            t[hsel][ix][q] = STAGE ? srow[c * W + xi] : rr[xi];
          }
      }
    }
#pragma unroll
    for (int ix = 0; ix < 4; ++ix) fea[ix] += fabsf(lv[ix]);
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      if (hsel < nhyp) {
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          float tq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) tq[q] = (okm[hsel] >> (ix * 4 + q)) & 1u ? t[hsel][ix][q] : 0.f;
          const float w1 = a[hsel][ix], w0 = 1.f - a[hsel][ix];
          // k = -1: xs+1 -> taps (f+1, f+2); k = 0: (f, f+1); k = +1: (f-1, f)
          cv[hsel][0][ix] += fabsf(lv[ix] - (w0 * tq[2] + w1 * tq[3]));
          cv[hsel][1][ix] += fabsf(lv[ix] - (w0 * tq[1] + w1 * tq[2]));
          cv[hsel][2][ix] += fabsf(lv[ix] - (w0 * tq[0] + w1 * tq[1]));
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int ix = 0; ix < 4; ++ix) {
    const int ch = iy * 4 + ix;
    const size_t o = (size_t)b * 64 * thw + tpix;
    out0[o + (size_t)ch * thw] = fea[ix];
    if (nhyp == 2) out1[o + (size_t)ch * thw] = fea[ix];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      out0[o + (size_t)(16 + k * 16 + ch) * thw] = cv[0][k][ix];
      if (nhyp == 2) out1[o + (size_t)(16 + k * 16 + ch) * thw] = cv[1][k][ix];
    }
  }
}

extern "C" int codd_tile_warp_cost(const float* fl, const float* fr, int B, int C, int Ht, int Wt, codd_view hyp0,
                                   codd_view hyp1, int nhyp, float* out0, float* out1, void* stream) {
  if (!fl || !fr || !hyp0.ptr || !out0 || nhyp < 1 || nhyp > 2 || (nhyp == 2 && (!hyp1.ptr || !out1)))
    return CODD_EINVAL;
  const size_t lds = (size_t)C * 4 * Wt * sizeof(float);
  if (lds <= 96 * 1024 && ((uintptr_t)fr & 15) == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)tile_warp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return CODD_EUNSUPPORTED;
    const int bx = Wt > 128 ? 256 : (Wt > 64 ? 128 : 64);
    dim3 grid(cdiv(Wt, bx), 4 * Ht, B);
    tile_warp_kernel<true><<<grid, bx, lds, (hipStream_t)stream>>>(fl, fr, C, Ht, Wt, hyp0, hyp1, nhyp, out0, out1);
    CODD_LAUNCH_CHECK();
    return CODD_OK;
  }
  const int bx = 64;
  dim3 grid(cdiv(Wt, bx), 4 * Ht, B);
  tile_warp_kernel<false><<<grid, bx, 0, (hipStream_t)stream>>>(fl, fr, C, Ht, Wt, hyp0, hyp1, nhyp, out0, out1);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Plane up-sampling x2 (reference propagation.py:10-32) and hypothesis selection (:225-240).
// ------------------------------------------------------------------------------------------------
__global__ void hyp_upsample_kernel(codd_view in, int h, int w, float scale, float* out, int out_ctot, int out_coff,
                                    long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int W2 = 2 * w, H2 = 2 * h;
  const int x = (int)(e % W2);
  long long t = e / W2;
  const int y = (int)(t % H2); t /= H2;
  const int c = (int)(t % 16);
  const int b = (int)(t / 16);
  const size_t hw = (size_t)h * w;
  const float* ip = in.ptr + ((size_t)b * in.ctot + in.coff) * hw + (size_t)(y >> 1) * w + (x >> 1);
  float v;
  if (c == 0) {
    const float cx = (x & 1) ? 0.5f : -0.5f, cy = (y & 1) ? 0.5f : -0.5f;
    v = (ip[0] + cx * ip[hw] + cy * ip[2 * hw]) * scale;
  } else {
    v = ip[(size_t)c * hw];
  }
  out[((size_t)b * out_ctot + out_coff + c) * (size_t)(4 * hw) + (size_t)y * W2 + x] = v;
}

extern "C" int codd_hyp_upsample(codd_view in, int B, int h, int w, float scale, float* out, int out_ctot,
                                 int out_coff, void* stream) {
  if (!in.ptr || !out) return CODD_EINVAL;
  long long total = (long long)B * 16 * 4 * h * w;
  hyp_upsample_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, h, w, scale, out, out_ctot, out_coff,
                                                                         total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

__global__ void hyp_select_kernel(const float* __restrict__ upd, codd_view cur, codd_view prev, int hw, float* out,
                                  int out_ctot, int out_coff, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int pix = (int)(e % hw);
  long long t = e / hw;
  const int c = (int)(t % 16);
  const int b = (int)(t / 16);
  const float* u = upd + (size_t)b * 34 * hw + pix;
  // torch.max(conf, dim=1) returns index 0 (previous) on ties / first maximum
  const bool sel_cur = u[hw] > u[0];
  float v;
  if (sel_cur) v = cur.ptr[((size_t)b * cur.ctot + cur.coff + c) * hw + pix] + u[(size_t)(18 + c) * hw];
  else v = prev.ptr[((size_t)b * prev.ctot + prev.coff + c) * hw + pix] + u[(size_t)(2 + c) * hw];
  if (c == 0) v = fmaxf(v, 0.f);
  out[((size_t)b * out_ctot + out_coff + c) * hw + pix] = v;
}

extern "C" int codd_hyp_select(const float* upd, codd_view cur, codd_view prev, int B, int h, int w, float* out,
                               int out_ctot, int out_coff, void* stream) {
  if (!upd || !cur.ptr || !prev.ptr || !out) return CODD_EINVAL;
  long long total = (long long)B * 16 * h * w;
  hyp_select_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(upd, cur, prev, h * w, out, out_ctot, out_coff,
                                                                       total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_abi_version(void) { return CODD_ABI_VERSION; }

// ---- option table (include/codd_hip.h: the library's only process-wide state; nothing is read from the environment) ----
static int g_opt[CODD_OPT_COUNT] = {192, 5};
int codd_opt(int key) { return g_opt[key]; }
extern "C" int codd_get_option(int key) { return key >= 0 && key < CODD_OPT_COUNT ? g_opt[key] : CODD_EINVAL; }
extern "C" int codd_set_option(int key, int value) {
  if (key < 0 || key >= CODD_OPT_COUNT) return CODD_EINVAL;
  if (key == CODD_OPT_GN_Q4 && (value < 16 || value > 4096)) return CODD_EINVAL;
  if (key == CODD_OPT_GN_BUILDER && value != 3 && value != 5) return CODD_EINVAL;
  const int prev = g_opt[key];
  g_opt[key] = value;
  return prev;
}
