// Rolling-window convolutions (exact fp32, v_mfma_f32_16x16x4_f32) for the large-map, few-channel layers of HITNet:
// one 3x3 convolution, a PAIR of 3x3 convolutions (BasicBlock: reference model/stereo/hitnet/propagation.py:103-121;
// the two trailing 3x3 layers of a U-Net merge, backbone.py:24-39) or a 1x1 -> 3x3 pair (the head of a merge /
// PostTileUpdate, backbone.py:24-30, propagation.py:255-258) per launch, C = 16 or 32 channels, stride 1, "same".
//
// Why: at 576x960 / 288x480 these layers have 36 FLOP per byte, so they sit between the HBM and the fp32-matrix
// roofline, but the tile-per-workgroup kernels (conv_quad_kernel.h) spend most of a tile's life in its prologue
// (stage the halo tile + the weights, barrier, ~1 us of MFMAs, store) and re-read every halo row.  Here
//   * a workgroup owns a 64-column STRIP and walks down RH rows, one row per step: every input row is fetched once
//     per strip (registers -> LDS ring, loads issued one step ahead), no halo rows are re-read or re-computed inside
//     a strip;
//   * a wave owns one 16-pixel segment of the strip and 16 output channels for ALL layers, and keeps its share of
//     the WEIGHTS IN REGISTERS for the whole launch (36 VGPRs per 16->16 3x3 layer, 72 per 32->32): the only LDS
//     operand traffic is one ds_read_b128 per four MFMAs;
//   * the intermediate activation of a pair never leaves the chip: stage A's row t is written (channel-quad layout
//     = the accumulator layout, one ds_write_b128 per lane) to a 4-row LDS ring, stage B computes row t - 3 from it
//     in the same step -- ONE barrier per step, both stages' MFMA chains independent (two accumulators in flight);
//   * the BasicBlock residual is read back from the input ring (it is the chain input).
// Layout of a ring row: [channel quad][pixel -1 .. 64][4 floats]; lane (g, j) of a wave reads the 16 bytes
// "channels 4g .. 4g+3 of pixel j + dx" -- 16 consecutive lanes read 256 consecutive bytes (conflict-free).
//
// Columns: a 3x3 layer consumes one column on each side of the strip, so a strip of 64 input columns yields 62
// (one layer) or 60 (pair of 3x3) output columns; the strips overlap by that much (3-6 % extra work instead of the
// 30-55 % of a 2-D halo tile).  Everything outside the image is zero for every layer (each layer sees the zero
// padding of a stand-alone "same" convolution of its input map).
#include "common.h"
#include "conv_kernel.h"  // view_ptr, conv_xcd_item

#ifdef ROLL_ABL_CLK  // dev (tools/ubench/roll_ablate.hip): shader-clock / wall-clock ticks of workgroup 0's main loop
__device__ long long roll_clk[16];
#endif
namespace {

// the activations HITNet uses (the generic act_apply carries exp / log / tanh code for every epilogue value)
__device__ __forceinline__ float roll_act(float v, int act, int co) {
  if (act == CODD_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
  if (act == CODD_ACT_RELU || (act == CODD_ACT_RELU_CH0 && co == 0)) return fmaxf(v, 0.f);
  return v;
}

constexpr int RPITCH = 66;  // float4 slots per (ring row, channel quad): pixels -1 .. 64

struct RollK {
  codd_roll_params p;
  int nstrips, nrb;  // strips per row, row blocks per image
  int stride;        // output columns per strip (62 | 60)
  int halo;          // columns a strip's input starts left of its first output column (1 | 2)
};

// MODE 0: 3x3            input ring -> out
// MODE 1: 3x3 -> 3x3     input ring -> ring A -> out  (+ chain input as residual)
// MODE 2: 1x1 -> 3x3     global (in0 | in1, NG groups of 16 channels) -> ring A -> out
template <int C, int MODE, int NG>
__global__ __launch_bounds__(64 * 4 * (C / 16)) void conv_roll_kernel(const RollK k) {
  constexpr int NH = C / 16;        // 16-channel halves of the output
  constexpr int NW = 4 * NH;        // waves: (segment 0..3, half)
  constexpr int NT = 64 * NW;
  constexpr int NCQ = C / 4;        // channel quads per ring row
  constexpr int GR = C / 16;        // 16-channel groups of a C-channel input
  constexpr int NIN = MODE == 2 ? 0 : (MODE == 0 ? 5 : 6);  // input ring rows
  constexpr int NA = MODE == 0 ? 0 : 4;                      // stage-A ring rows
  constexpr int LAG = MODE == 0 ? 0 : 3;                     // output row of step t = t - LAG
  extern __shared__ __attribute__((aligned(16))) float4 smem[];
  float4* rin = smem;                           // [NIN][NCQ][RPITCH]
  float4* ra = smem + NIN * NCQ * RPITCH;       // [NA][NCQ][RPITCH]
  const codd_roll_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int seg = wave & 3, half = wave >> 2;

#ifdef ROLL_ABL_CLK
  const long long wal_in_ = wall_clock64();
#endif
  int bid = conv_xcd_item(blockIdx.x, gridDim.x);
  const int strip = bid % k.nstrips; bid /= k.nstrips;
  const int rb = bid % k.nrb;
  const int b = bid / k.nrb;
  const int y0 = rb * p.rh;
  const int RH = min(p.rh, p.H - y0);
  const int xs = strip * k.stride - k.halo;  // image column of the strip's local column 0
  const int HW = p.H * p.W;

  // ---- weights + biases -> registers (once per workgroup) --------------------------------------------------
  // packed 3x3 stage: [half][tap][gr][lane][4]; packed 1x1 stage: [half][gr][lane][4]
  float4 w3a[MODE == 2 ? 1 : 9][GR], w3b[MODE == 0 ? 1 : 9][GR], w1[MODE == 2 ? NG : 1];
  {
    const float4* wa = (const float4*)p.wA;
    const float4* wb = (const float4*)p.wB;
    if constexpr (MODE != 2) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int gr = 0; gr < GR; ++gr) w3a[t][gr] = wa[((half * 9 + t) * GR + gr) * 64 + lane];
    } else {
#pragma unroll
      for (int gr = 0; gr < NG; ++gr) w1[gr] = wa[(half * NG + gr) * 64 + lane];
    }
    if constexpr (MODE != 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int gr = 0; gr < GR; ++gr) w3b[t][gr] = wb[((half * 9 + t) * GR + gr) * 64 + lane];
    }
  }
  f32x4 biasA = {0.f, 0.f, 0.f, 0.f}, biasB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (p.bA) biasA[r] = p.bA[16 * half + 4 * g + r];
    if (MODE != 0 && p.bB) biasB[r] = p.bB[16 * half + 4 * g + r];
  }

  // ---- input ring staging (MODE 0 / 1): thread -> (channel, four pixels) of a row --------------------------
  const int s_ch = tid >> 4, s_x4 = tid & 15;      // NT / 16 = C channels, 16 float4 columns
  const int s_gx = xs + 4 * s_x4;
  const bool s_vec = s_gx >= 0 && s_gx + 3 < p.W;  // the four pixels are inside the image (else per-pixel)
  const float* s_src = nullptr;
  if constexpr (MODE != 2)
    s_src = (s_ch < p.C0 ? view_ptr(p.in0, b, s_ch, HW) : view_ptr(p.in1, b, s_ch - p.C0, HW));
  float* s_dst = (float*)rin + ((s_ch >> 2) * RPITCH + 4 * s_x4 + 1) * 4 + (s_ch & 3);
  float sreg[4];
  // local input row li <-> image row y0 - HIN + li
  constexpr int HIN = MODE == 1 ? 2 : 1;
  auto issue = [&](int li) {
    const int y = y0 - HIN + li;
    sreg[0] = sreg[1] = sreg[2] = sreg[3] = 0.f;
    if ((unsigned)y < (unsigned)p.H) {
      const float* s_ = s_src + (size_t)y * p.W + s_gx;
      if (s_vec) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u v = *(const f4u*)s_;
        sreg[0] = v[0]; sreg[1] = v[1]; sreg[2] = v[2]; sreg[3] = v[3];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((unsigned)(s_gx + q) < (unsigned)p.W) sreg[q] = s_[q];
      }
    }
  };
  auto commit = [&](int slot) {
    float* d_ = s_dst + slot * (NCQ * RPITCH * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) d_[4 * q] = sreg[q];
  };

  // ---- per-lane constants of the compute phase ---------------------------------------------------------------
  const int px = 16 * seg + j;             // local column of this lane's pixel
  const int gx = xs + px;                  // image column
  const bool col_in = (unsigned)gx < (unsigned)p.W;
  // MODE 2: stage A reads the chain input straight from global memory: channel 16 gr + 4 g + s of pixel (y, gx)
  float a_in[MODE == 2 ? NG : 1][4];
  auto issue_a = [&](int y) {
    if constexpr (MODE == 2) {
    const bool ok = col_in && (unsigned)y < (unsigned)p.H;
#pragma unroll
    for (int gr = 0; gr < (MODE == 2 ? NG : 1); ++gr)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = 16 * gr + 4 * g + s;
        float v = 0.f;
        if (ok && c < p.C0 + p.C1)
          v = (c < p.C0 ? view_ptr(p.in0, b, c, HW) : view_ptr(p.in1, b, c - p.C0, HW))[(size_t)y * p.W + gx];
        a_in[gr][s] = v;
      }
    }
  };

  // one 3x3 stage row: ring rows (slots s0, s1, s2 = the rows above / at / below), this lane's 16 x 16 tile
#define ROLL_3X3(ACC, RING, W3, S0, S1, S2)                                                                  \
  {                                                                                                          \
    const float4* r0_ = (RING) + ((S0) * NCQ + g) * RPITCH + px;                                              \
    const float4* r1_ = (RING) + ((S1) * NCQ + g) * RPITCH + px;                                              \
    const float4* r2_ = (RING) + ((S2) * NCQ + g) * RPITCH + px;                                              \
    _Pragma("unroll") for (int gr = 0; gr < GR; ++gr) {                                                       \
      float4 bv_[9];                                                                                          \
      _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                      \
        bv_[kx] = r0_[gr * 4 * RPITCH + kx];                                                                  \
        bv_[3 + kx] = r1_[gr * 4 * RPITCH + kx];                                                              \
        bv_[6 + kx] = r2_[gr * 4 * RPITCH + kx];                                                              \
      }                                                                                                       \
      __builtin_amdgcn_sched_barrier(0); /* all nine reads in flight before the first MFMA waits */           \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                         \
        ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(W3[t][gr].x, bv_[t].x, ACC, 0, 0, 0);                      \
        ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(W3[t][gr].y, bv_[t].y, ACC, 0, 0, 0);                      \
        ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(W3[t][gr].z, bv_[t].z, ACC, 0, 0, 0);                      \
        ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(W3[t][gr].w, bv_[t].w, ACC, 0, 0, 0);                      \
      }                                                                                                       \
    }                                                                                                         \
  }

  // two independent 3x3 stage rows with their MFMA chains interleaved (dependent distance 2 issue slots = 64 cycles >
  // the 40-cycle latency of v_mfma_f32_16x16x4_f32): stage A's row t and stage B's row t - 3 of a pair
#ifdef ROLL_ABL_NOMFMA  // (dev ablation: tools/ubench/roll_ablate.hip) operands consumed by one VALU op instead of MFMAs
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) ((c) + (a) * (b))
#endif
#define ROLL_MM8(ACCA, WA_, VA_, ACCB, WB_, VB_)                                                             \
  ACCA = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.x, VA_.x, ACCA, 0, 0, 0);                                   \
  ACCB = __builtin_amdgcn_mfma_f32_16x16x4f32(WB_.x, VB_.x, ACCB, 0, 0, 0);                                   \
  ACCA = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.y, VA_.y, ACCA, 0, 0, 0);                                   \
  ACCB = __builtin_amdgcn_mfma_f32_16x16x4f32(WB_.y, VB_.y, ACCB, 0, 0, 0);                                   \
  ACCA = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.z, VA_.z, ACCA, 0, 0, 0);                                   \
  ACCB = __builtin_amdgcn_mfma_f32_16x16x4f32(WB_.z, VB_.z, ACCB, 0, 0, 0);                                   \
  ACCA = __builtin_amdgcn_mfma_f32_16x16x4f32(WA_.w, VA_.w, ACCA, 0, 0, 0);                                   \
  ACCB = __builtin_amdgcn_mfma_f32_16x16x4f32(WB_.w, VB_.w, ACCB, 0, 0, 0);
  // (operand reads of kernel row r + 1 are issued before the MFMAs of row r: at most two rows of operands live)
#define ROLL_3X3_X2(ACCA, RINGA, WA, A0, A1, A2, ACCB, RINGB, WB, B0, B1, B2)                                \
  {                                                                                                          \
    const float4* a_[3] = {(RINGA) + ((A0) * NCQ + g) * RPITCH + px, (RINGA) + ((A1) * NCQ + g) * RPITCH + px, \
                           (RINGA) + ((A2) * NCQ + g) * RPITCH + px};                                         \
    const float4* b_[3] = {(RINGB) + ((B0) * NCQ + g) * RPITCH + px, (RINGB) + ((B1) * NCQ + g) * RPITCH + px, \
                           (RINGB) + ((B2) * NCQ + g) * RPITCH + px};                                         \
    _Pragma("unroll") for (int gr = 0; gr < GR; ++gr) {                                                       \
      float4 va_[3][3], vb_[3][3];                                                                            \
      _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                      \
        va_[0][kx] = a_[0][gr * 4 * RPITCH + kx];                                                             \
        vb_[0][kx] = b_[0][gr * 4 * RPITCH + kx];                                                             \
      }                                                                                                       \
      _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                      \
        if (ky < 2) {                                                                                         \
          _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                  \
            va_[ky + 1][kx] = a_[ky + 1][gr * 4 * RPITCH + kx];                                               \
            vb_[ky + 1][kx] = b_[ky + 1][gr * 4 * RPITCH + kx];                                               \
          }                                                                                                   \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                    \
          ROLL_MM8(ACCA, WA[3 * ky + kx][gr], va_[ky][kx], ACCB, WB[3 * ky + kx][gr], vb_[ky][kx])            \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }                                                                                                       \
    }                                                                                                         \
  }

  // ---- prologue: the rows step 0 needs ---------------------------------------------------------------------------
  // MODE 0: output row t needs input li = t, t+1, t+2;  MODE 1: stage-A row la = t (image row y0 - 1 + t) needs
  // input li = t, t+1, t+2.  Step t loads li = t + 4 at its start and commits it at its END, in front of the step's
  // global stores: the wait for the load then never waits for a store younger than the previous step's (vmcnt counts
  // loads and stores together), and the stores' latency runs under the next step's MFMAs.
  if constexpr (MODE != 2) {
    float keep[4][4];
#pragma unroll
    for (int li = 0; li < 4; ++li) {
      issue(li);
#pragma unroll
      for (int q = 0; q < 4; ++q) keep[li][q] = sreg[q];
    }
#pragma unroll
    for (int li = 0; li < 4; ++li) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sreg[q] = keep[li][q];
      commit(li);
    }
  }
  // LDS-only barrier: __syncthreads() also waits for the step's GLOBAL stores (release fence), ~1-2 k cycles per step
#ifdef ROLL_ABL_NOBARRIER
#define ROLL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define ROLL_BARRIER()                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
  __builtin_amdgcn_s_barrier();
#endif
  ROLL_BARRIER();

  const int T = RH + LAG;  // steps: outputs lo = t - LAG, lo = 0 .. RH - 1
  int sl_in = 0;                                  // t % NIN
  float* outp = p.out + ((size_t)b * p.out_ctot + p.out_coff + 16 * half + 4 * g) * (size_t)HW;
  const bool st_col = col_in && px >= (MODE == 1 ? 2 : 1) && px <= (MODE == 1 ? 61 : 62);

#ifdef ROLL_ABL_CLK
  const long long clk0_ = clock64(), wal0_ = wall_clock64();
#endif
  for (int t = 0; t < T; ++t) {
    // (1) start the load of input row li = t + 4 (MODE 2: of this step's stage-A operands)
#ifndef ROLL_ABL_NOLOAD
    if constexpr (MODE != 2) issue(t + 4);
#endif
    // (2) + (3) MFMA phase.  stage A (pairs): row la = t, image row ya = y0 - 1 + t, for la <= RH + 1;
    // last stage: output row lo = t - LAG (lo < RH by the loop bound)
    const int ya = y0 - 1 + t;
    const bool rowA = MODE != 0 && (unsigned)ya < (unsigned)p.H && t <= RH + 1;
    const int lo = t - LAG, yo = y0 + lo;
    const bool rowB = lo >= 0;
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    int s1 = sl_in + 1, s2 = sl_in + 2;
    if (NIN) {
      if (s1 >= NIN) s1 -= NIN;
      if (s2 >= NIN) s2 -= NIN;
    }
    if constexpr (MODE == 0) {
      ROLL_3X3(accB, rin, w3a, sl_in, s1, s2);
    } else if constexpr (MODE == 1) {
      if (rowA && rowB) {
        ROLL_3X3_X2(accA, rin, w3a, sl_in, s1, s2, accB, ra, w3b, (lo & 3), ((lo + 1) & 3), ((lo + 2) & 3));
      } else if (rowA) {
        ROLL_3X3(accA, rin, w3a, sl_in, s1, s2);
      } else if (rowB) {
        ROLL_3X3(accB, ra, w3b, (lo & 3), ((lo + 1) & 3), ((lo + 2) & 3));
      }
    } else {
      issue_a(ya);  // this step's stage-A operands: in flight under stage B's MFMAs
      if (rowB) ROLL_3X3(accB, ra, w3b, (lo & 3), ((lo + 1) & 3), ((lo + 2) & 3));
      if (rowA) {
#pragma unroll
        for (int gr = 0; gr < NG; ++gr) {
          accA = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[gr].x, a_in[gr][0], accA, 0, 0, 0);
          accA = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[gr].y, a_in[gr][1], accA, 0, 0, 0);
          accA = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[gr].z, a_in[gr][2], accA, 0, 0, 0);
          accA = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[gr].w, a_in[gr][3], accA, 0, 0, 0);
        }
      }
    }
    // commit input row li = t + 4 (ring slot (t + 4) % NIN: not read by this step)
    if constexpr (MODE != 2) {
      int slc = sl_in + 4; if (slc >= NIN) slc -= NIN;
      commit(slc);
    }
    // stage A epilogue: the row goes to ring A (zero outside the image)
    if constexpr (MODE != 0) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rowA && col_in) {
        v.x = roll_act(accA[0] + biasA[0], p.actA, 16 * half + 4 * g + 0);
        v.y = roll_act(accA[1] + biasA[1], p.actA, 16 * half + 4 * g + 1);
        v.z = roll_act(accA[2] + biasA[2], p.actA, 16 * half + 4 * g + 2);
        v.w = roll_act(accA[3] + biasA[3], p.actA, 16 * half + 4 * g + 3);
      }
      ra[((t & 3) * NCQ + 4 * half + g) * RPITCH + px + 1] = v;
    }
    // last stage epilogue
    if (rowB) {
      const f32x4& bias = MODE == 0 ? biasA : biasB;
      const int act = MODE == 0 ? p.actA : p.actB;
      f32x4 res = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == 1) if (p.residual) {  // the chain input at (yo, gx): input row li = lo + 2 = t - 1
        int sr = sl_in - 1; if (sr < 0) sr += NIN;
        const float4 rv = rin[(sr * NCQ + 4 * half + g) * RPITCH + px + 1];
        res = f32x4{rv.x, rv.y, rv.z, rv.w};
      }
      if (st_col) {
        float* o_ = outp + (size_t)yo * p.W + gx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * half + 4 * g + r;
#ifdef ROLL_ABL_NOSTORE
          if (co < p.cout_store && accB[r] == 1.2345e30f) o_[(size_t)r * HW] = roll_act(accB[r] + bias[r] + res[r], act, co);
#else
          if (co < p.cout_store) o_[(size_t)r * HW] = roll_act(accB[r] + bias[r] + res[r], act, co);
#endif
        }
      }
    }
    ROLL_BARRIER();
    if (++sl_in >= (NIN ? NIN : 1)) sl_in = 0;
  }
#ifdef ROLL_ABL_CLK
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == gridDim.x / 2)) {
    const int o_ = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 4 : 8);
    roll_clk[o_] = clock64() - clk0_; roll_clk[o_ + 1] = wal_in_; roll_clk[o_ + 2] = wal0_; roll_clk[o_ + 3] = wall_clock64();
  }
#endif
#undef ROLL_3X3
#undef ROLL_3X3_X2
#undef ROLL_MM8
#undef ROLL_BARRIER
}

template <int C, int MODE, int NG>
int roll_launch(const RollK& k, int grid, hipStream_t s) {
  constexpr int NCQ = C / 4;
  constexpr int NIN = MODE == 2 ? 0 : (MODE == 0 ? 5 : 6), NA = MODE == 0 ? 0 : 4;
  const size_t lds = (size_t)(NIN + NA) * NCQ * RPITCH * sizeof(float4);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_roll_kernel<C, MODE, NG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  conv_roll_kernel<C, MODE, NG><<<grid, 64 * 4 * (C / 16), lds, s>>>(k);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

__global__ void roll_pack3_kernel(const float* __restrict__ w, float4* __restrict__ out, int C, int Cout, int Cin) {
  // out[((half * 9 + tap) * GR + gr) * 64 + lane] = W[co = 16 half + (lane & 15)][ci = 16 gr + 4 (lane >> 4) + s][tap]
  const int GR = C / 16, n = (C / 16) * 9 * GR * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int lane = i & 63, gr = (i >> 6) % GR, tap = ((i >> 6) / GR) % 9, half = (i >> 6) / (GR * 9);
  const int co = 16 * half + (lane & 15);
  float v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ci = 16 * gr + 4 * (lane >> 4) + s;
    v[s] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
  out[i] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void roll_pack1_kernel(const float* __restrict__ w, float4* __restrict__ out, int C, int Cout, int Cin, int NG) {
  // out[(half * NG + gr) * 64 + lane] = W[co = 16 half + (lane & 15)][ci = 16 gr + 4 (lane >> 4) + s]
  const int n = (C / 16) * NG * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int lane = i & 63, gr = (i >> 6) % NG, half = (i >> 6) / NG;
  const int co = 16 * half + (lane & 15);
  float v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ci = 16 * gr + 4 * (lane >> 4) + s;
    v[s] = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.f;
  }
  out[i] = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

extern "C" long long codd_roll_packed_size(int C, int k, int Cin) {
  if (!(C == 16 || C == 32) || !(k == 1 || k == 3) || Cin < 1) return CODD_EINVAL;
  if (k == 3) return (long long)(C / 16) * 9 * (C / 16) * 64 * 4;
  return (long long)(C / 16) * ((Cin + 15) / 16) * 64 * 4;
}

extern "C" int codd_roll_pack_weights(const float* w, float* packed, int C, int Cout, int Cin, int k, void* stream) {
  if (!w || !packed || !(C == 16 || C == 32) || Cout < 1 || Cout > C || Cin < 1) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k == 3) {
    if (Cin > C) return CODD_EINVAL;
    const int n = (C / 16) * 9 * (C / 16) * 64;
    roll_pack3_kernel<<<cdiv(n, 256), 256, 0, s>>>(w, (float4*)packed, C, Cout, Cin);
  } else if (k == 1) {
    if (Cin > 64) return CODD_EINVAL;
    const int NG = (Cin + 15) / 16, n = (C / 16) * NG * 64;
    roll_pack1_kernel<<<cdiv(n, 256), 256, 0, s>>>(w, (float4*)packed, C, Cout, Cin, NG);
  } else {
    return CODD_EINVAL;
  }
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_conv_roll(const codd_roll_params* pp, void* stream) {
  if (!pp) return CODD_EINVAL;
  const codd_roll_params& p = *pp;
  if (!(p.C == 16 || p.C == 32) || p.mode < 0 || p.mode > 2) return CODD_EUNSUPPORTED;
  if (!p.in0.ptr || !p.out || !p.wA || p.B < 1 || p.H < 1 || p.W < 1 || p.rh < 1 || p.C0 < 1 || p.C1 < 0) return CODD_EINVAL;
  if (p.C1 > 0 && !p.in1.ptr) return CODD_EINVAL;
  if (p.mode != 0 && !p.wB) return CODD_EINVAL;
  if (p.cout_store < 1 || p.cout_store > p.C) return CODD_EINVAL;
  if (((uintptr_t)p.wA & 15) || ((uintptr_t)p.wB & 15)) return CODD_EINVAL;
  const int cin = p.C0 + p.C1;
  if (p.mode != 2 && cin != p.C) return CODD_EUNSUPPORTED;  // 3x3 first stage: C -> C
  if (p.mode == 2 && cin > 64) return CODD_EUNSUPPORTED;
  if (p.residual && p.mode != 1) return CODD_EINVAL;
  for (int a : {p.actA, p.mode ? p.actB : CODD_ACT_NONE})
    if (!(a == CODD_ACT_NONE || a == CODD_ACT_LRELU02 || a == CODD_ACT_RELU || a == CODD_ACT_RELU_CH0)) return CODD_EUNSUPPORTED;
  RollK k;
  k.p = p;
  k.halo = p.mode == 1 ? 2 : 1;
  k.stride = 64 - 2 * k.halo;
  k.nstrips = cdiv(p.W, k.stride);
  k.nrb = cdiv(p.H, p.rh);
  const long long grid = (long long)k.nstrips * k.nrb * p.B;
  if (grid > 0x7fffffffLL) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ng = (cin + 15) / 16;
#define ROLL_CASE(C_, M_, N_) return roll_launch<C_, M_, N_>(k, (int)grid, s)
  if (p.C == 16) {
    if (p.mode == 0) ROLL_CASE(16, 0, 1);
    if (p.mode == 1) ROLL_CASE(16, 1, 1);
    if (ng == 1) ROLL_CASE(16, 2, 1);
    if (ng == 2) ROLL_CASE(16, 2, 2);
    if (ng == 3) ROLL_CASE(16, 2, 3);
    ROLL_CASE(16, 2, 4);
  }
  if (p.mode == 0) ROLL_CASE(32, 0, 1);
  if (p.mode == 1) ROLL_CASE(32, 1, 1);
  if (ng == 1) ROLL_CASE(32, 2, 1);
  if (ng == 2) ROLL_CASE(32, 2, 2);
  if (ng == 3) ROLL_CASE(32, 2, 3);
  ROLL_CASE(32, 2, 4);
#undef ROLL_CASE
}
