#pragma once
#include <stdlib.h>
// Split-bf16 convolution kernel: implicit GEMM on v_mfma_f32_16x16x32_bf16 (16x the rate of the f32-input MFMA).
//
// Numerics.  TERMS = 3 ("split-bf16", the fp32-grade path): every fp32 operand x is split into two bf16 numbers
//   hi = bf16_rne(x), lo = bf16_rne(x - hi)          (|x - hi - lo| <= 2^-18 |x|),
// and a product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi -- three bf16 MFMAs whose products are exact
// in the fp32 accumulator; the dropped terms (a_lo*b_lo and the two split residuals) are <= 3 * 2^-18 |a b|
// (measured on MI355X: max error 1e-6 of sum |w||x| against 3e-7 for the exact-fp32 kernels; TF32, the reference's
// own CUDA conv arithmetic under torch >= 1.12 defaults, keeps 10 mantissa bits).  TERMS = 1 is the plain
// bf16-operand / fp32-accumulate path of BASELINE.json configs[4]; TERMS = 16 (CODD_TERMS_F16, round 5) the same
// kernel on ONE plane of IEEE-fp16 records and v_mfma_f32_16x16x32_f16 -- the reference's own reduced precision
// (auto_fp16, model/codd.py:37,128): 11 mantissa bits at the bf16 rate.
//
// GEMM view (same roles as conv_kernel.h):  D[co, pixel] = sum_k W[co, k] X[k, pixel],  k = (tap, ci).
//   A operand (16 x 32) = weights : lane (j = l&15, g = l>>4) holds W[co = j][entry g][8 channels]
//   B operand (32 x 16) = im2col  : lane (j, g) holds X[entry g][8 channels][pixel j]
//   D (16 x 16)                   : lane holds D[co = 4g + r][pixel = j], r = 0..3
// A k-step consumes four ENTRIES; an entry is (tap, channel octet) in the order e = tap * noct + oct.  The LDS slot
// offset of every entry is kept in a small table (built once per workgroup), so any (kh, kw, dilation, stride, ck)
// runs through the same loop; entries beyond ntaps * noct point at entry 0 and carry zero weights.
//
// Workgroup = PGW x CGW CONSUMER waves + 4 PRODUCER waves (wave specialisation).  With the matrix pipe 5x faster
// than on the fp32 path the kernel lives or dies by its staging, so staging gets its own waves and is pure LDS-DMA:
//   producers  global_load_lds_dwordx4 (1 KiB per wave-instruction, no registers, no VALU) of BOTH operands, issued
//              two chunks ahead into a ring of three LDS buffers; counted s_waitcnt vmcnt + s_barrier per chunk
//   consumers  LDS buffer c % 3 -> operand fragments (read one k-step ahead, interleaved 1 read : 2 MFMAs) -> MFMA
// Both operands therefore exist in global memory in exactly their LDS form:
//   weights [plane][k-step][g][co (16 * mb)][8]  packed once per layer (codd_conv2d_pack_weights_bf16)
//   input   [plane][octet][y][x][8]              the activation tensor re-laid-out by codd_split_bf16 (conv_bf16.hip):
//                                                 8 channels innermost, (hi, lo) bf16 planes, ZERO BORDER of the
//                                                 conv's padding plus the tile overhang, so a halo tile is plain
//                                                 row segments of 16-byte records and needs no bounds logic.
// LDS images are in 16-byte slots (= 8 bf16, one ds_read_b128 per operand fragment); the octet stride of the input
// image is a multiple of 256 B so the 16-lane groups of a ds_read_b128 never meet on a bank.
// Consumer wave (pg, cg) owns pixel units [pg*A, pg*A + A) (a unit = 16 consecutive pixels of one tile row; the tile
// has th rows x xb units) and the B 16-channel blocks [cg*B, cg*B + B) of the workgroup's 16*mb output channels
// (mb = B * CGW): A*B accumulator tiles per wave, (A + B) fragment reads per plane and k-step.
#include "conv_kernel.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvB {
  codd_conv_params p;
  int cin, nchunks, ntaps, noct;  // noct = ck / 8 channel octets per chunk
  int nk;                         // k-steps per chunk = ceil(ntaps * noct / 4)
  int th, tw, thi, twi;           // tile: th x tw output pixels, thi x twi input pixels (halo)
  int pu, xb;                     // pixel units per tile (th * xb), units per tile row
  int npix;                       // thi * twi
  int xplane;                     // 16-byte records per precision plane of the split input = xs_c8 * xs_hp * xs_wp
  int nring;                      // LDS ring depth: 3 (DMA two chunks ahead) when it fits, else 2
  int os16;                       // 16-byte slots per octet plane of the input image (multiple of 16)
  int iplane16;                   // slots per precision plane of the input image = noct * os16
  int ibuf16;                     // slots of one input buffer = planes * iplane16
  int wplane16;                   // slots per precision plane of the weight image = nk * 4 * nco
  int nco;                        // output channels per workgroup = 16 * mb
  int wslots;                     // slots of one (channel group, chunk) weight image = planes * wplane16
  int tiles_x, tiles_y, ncog, cout_eff;
  int khe;                        // tap rows of one tap set (kh, or kh / 2 with a dual tap set)
  // magic multipliers ceil(2^32 / d) of the producers' prologue divisions (convb_div; exact for the < 2^16 operands of
  // the slot / entry arithmetic): a 32-bit udiv is ~35 VALU instructions, and 20 of them stood between the start of a
  // producer wave and its first LDS-DMA -- ~1.2 us of every launch's exposed ring fill (round 5)
  unsigned m_iplane16, m_os16, m_twi, m_noct, m_kw;
};
static inline unsigned convb_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }
// n / d for 0 <= n < 2^16, 1 <= d < 2^16 with M = convb_magic(d): floor(n M / 2^32) = floor(n / d) (the error term
// n (M d - 2^32) / (d 2^32) < 2^-16 never reaches the next integer: frac(n / d) <= 1 - 1/d)
__device__ __forceinline__ int convb_div(int n, unsigned M, int d) { return d == 1 ? n : (int)__umulhi((unsigned)n, M); }

constexpr int CONVB_NWP = 4;  // producer waves per workgroup
constexpr int CONVB_MAXP = 6; // input DMA pieces per producer wave whose source offsets are kept in registers
// dev ablations (tools/ubench/convb_ablate.hip): drop the weight DMA / the input loads; CONVB_NO_BARRIER (with
// CONVB_NO_PRODUCER) / CONVB_NO_LDSREAD time the consumer loop without its chunk barriers / its fragment reads
#ifdef CONVB_NO_DMA
#define CONVB_DMA_N(n) 0
#else
#define CONVB_DMA_N(n) (n)
#endif
#ifdef CONVB_NO_INPUT
#define CONVB_IN_N(n) 0
#else
#define CONVB_IN_N(n) (n)
#endif

/* launch geometry of the split-bf16 kernel for layout-2 parameters: fills k, the dynamic LDS size and the grid.
 * Field use: nw = tile rows, npb = 16-pixel units per tile row (1 or 2), mb = 16-channel blocks per workgroup,
 * ck = channels per chunk (multiple of 8), pgw x cgw = consumer wave grid, terms = 1 | 3. */
static inline int convb_geometry(const codd_conv_params* pp, ConvB& k, size_t& lds, long long& grid, bool need_xs) {
  k.p = *pp;
  const codd_conv_params& p = k.p;
  if (p.ck < 8 || (p.ck & 7) || !CODD_TERMS_OK(p.terms) || p.nw < 1 || p.npb < 1 || p.npb > 2 ||
      p.pgw < 1 || p.cgw < 1 || p.mb < 1 || p.mb % p.cgw)
    return CODD_EINVAL;
  const int planes = CODD_TERMS_PLANES(p.terms);
  k.cin = p.C0 + p.C1;
  k.ntaps = p.kh * p.kw;
  k.noct = p.ck >> 3;
  k.nchunks = cdiv(k.cin, p.ck);
  k.nk = cdiv((long long)k.ntaps * k.noct, 4);
  k.th = p.nw; k.xb = p.npb; k.pu = k.th * k.xb; k.tw = 16 * k.xb;
  // dual tap set (dil2 > 0): kh counts the rows of both sets; the halo is that of the larger dilation (dil_y, dil_x)
  if (p.dil2 < 0 || (p.dil2 > 0 && ((p.kh & 1) || !(p.kh / 2 & 1) || !(p.kw & 1) || p.dil2 > p.dil_y || p.dil2 > p.dil_x)))
    return CODD_EINVAL;
  k.khe = p.dil2 > 0 ? p.kh / 2 : p.kh;
  k.thi = (k.th - 1) * p.sy + (k.khe - 1) * p.dil_y + 1;
  k.twi = (k.tw - 1) * p.sx + (p.kw - 1) * p.dil_x + 1;
  k.npix = k.thi * k.twi;
  k.os16 = ((k.npix + 15) / 16) * 16;
  while ((planes * k.noct * k.os16) & 63) k.os16 += 16;  // an input buffer is a whole number of 1 KiB DMA pieces
  k.iplane16 = k.noct * k.os16;
  k.ibuf16 = planes * k.iplane16;
  k.nco = 16 * p.mb;
  k.wplane16 = k.nk * 4 * k.nco;
  k.wslots = planes * k.wplane16;
  k.tiles_x = cdiv(p.Wout, k.tw);
  k.tiles_y = cdiv(p.Hout, k.th);
  k.cout_eff = p.store_mode ? 4 * p.Cout : p.Cout;
  k.ncog = cdiv(k.cout_eff, k.nco);
  // ring of (weights + input) buffers + entry table: three deep when that fits the 160 KiB of a CU, else two
  const size_t per = ((size_t)k.wslots + (size_t)k.ibuf16) * 16, tab = ((size_t)k.nk + 2) * 16;
  k.nring = (k.nchunks >= 3 && 3 * per + tab <= 160 * 1024) ? 3 : 2;
  // (a two-deep ring where that lets TWO 8-wave workgroups share a CU: +3 % per layer stand-alone in round 3, 101.8 / 101.6
  // against 101.6 / 101.7 frames/s in round 6's same-lease A/B, profiles/r06_ab_hr_fuse_ring2.log -- not kept)
  lds = k.nring * per + tab;
  if (lds > 160 * 1024) return CODD_EUNSUPPORTED;
  if (p.ksplit < 0 || p.ksplit > 2) return CODD_EINVAL;
  // k-split pairs exchange half of their accumulator tiles through the (then idle) ring: 1 KiB per tile and wave
  if (p.ksplit == 2 && (size_t)cdiv(k.pu, p.pgw) * p.mb * p.pgw * 1024 > k.nring * per) return CODD_EUNSUPPORTED;
  // DMA pieces a producer wave issues per chunk; nring - 1 chunks are in flight and vmcnt counts to 63
  if ((k.nring - 1) * (cdiv(k.wslots >> 6, CONVB_NWP) + cdiv(k.ibuf16 >> 6, CONVB_NWP)) > 56) return CODD_EUNSUPPORTED;
  k.xplane = p.xs_c8 * p.xs_hp * p.xs_wp;
  if (k.nring * (k.wslots + k.ibuf16) + 64 >= 65536 || (k.nk + 2) * 4 >= 65536) return CODD_EUNSUPPORTED;  // (convb_div's operand range; LDS bounds it far below)
  k.m_iplane16 = convb_magic(k.iplane16); k.m_os16 = convb_magic(k.os16); k.m_twi = convb_magic(k.twi);
  k.m_noct = convb_magic(k.noct); k.m_kw = convb_magic(p.kw);
  // the split input must hold every halo tile (codd_split_bf16_dims gives a sufficient size)
  if (need_xs && (!p.xs || p.xs_o8 < 0 || p.xs_c8 < p.xs_o8 + k.nchunks * k.noct || p.xs_bt < p.pad_t ||
                  p.xs_bl < p.pad_l || p.xs_hp < p.xs_bt + p.Hin || p.xs_wp < p.xs_bl + p.Win ||
                  p.xs_hp < p.xs_bt - p.pad_t + (k.tiles_y * k.th - 1) * p.sy + (k.khe - 1) * p.dil_y + 1 ||
                  p.xs_wp < p.xs_bl - p.pad_l + (k.tiles_x * k.tw - 1) * p.sx + (p.kw - 1) * p.dil_x + 1))
    return CODD_EINVAL;
  if (p.gate < 0 || p.gate > 3) return CODD_EINVAL;
  if (need_xs && p.gate) {  // ConvGRU gate epilogues (include/codd_hip.h): operands are channel-quad fp32 tensors
    const int G = p.gate == 2 ? k.cout_eff / 3 : k.cout_eff;
    auto c4ok = [](const codd_view& v, int need) {
      return v.ptr && !((uintptr_t)v.ptr & 15) && !(v.ctot & 3) && !(v.coff & 3) && v.coff >= 0 && v.coff + need <= v.ctot;
    };
    if (p.store_mode || (G & 15) || (p.gate == 2 && 3 * G != k.cout_eff)) return CODD_EUNSUPPORTED;
    if (!p.out || ((uintptr_t)p.out & 15) || (p.out_ctot & 3) || (p.out_coff & 3) ||
        p.out_coff + (p.gate == 2 ? 2 * G : G) > p.out_ctot || (p.bias && ((uintptr_t)p.bias & 15)))
      return CODD_EINVAL;
    if (p.gate == 1 && (p.xso || p.res1.ptr || p.res2.ptr || p.post.ptr)) return CODD_EUNSUPPORTED;
    if (p.gate == 2 && (!p.xso || !c4ok(p.res1, 3 * G) || !c4ok(p.res2, 2 * G) || !c4ok(p.post, G))) return CODD_EINVAL;
    if (p.gate == 3 && (!p.xso || !c4ok(p.res1, 2 * G) || !c4ok(p.post, G) || p.res2.ptr)) return CODD_EINVAL;
    if (p.xso && (p.xso_terms != p.terms || p.xso_o8 < 0 || p.xso_bt < 0 || p.xso_bl < 0 ||
                  p.xso_c8 < p.xso_o8 + cdiv(G, 8) || p.xso_hp < p.xso_bt + p.Hout || p.xso_wp < p.xso_bl + p.Wout))
      return CODD_EINVAL;
  } else if (need_xs && p.xso) {  // split-record output: plain conv only; the tensor must hold the image inside its borders
    if (p.store_mode || p.res1.ptr || p.res2.ptr || p.post.ptr || p.act == CODD_ACT_RELU_CH0) return CODD_EUNSUPPORTED;
    if (p.bias && ((uintptr_t)p.bias & 15)) return CODD_EINVAL;
    // (a kernel writes records in its OWN operand format: the epilogue's conversion is chosen at compile time)
    if (p.xso_terms != p.terms || p.xso_o8 < 0 || p.xso_bt < 0 || p.xso_bl < 0 ||
        p.xso_c8 < p.xso_o8 + cdiv(k.cout_eff, 8) || p.xso_hp < p.xso_bt + p.Hout || p.xso_wp < p.xso_bl + p.Wout)
      return CODD_EINVAL;
  }
  grid = (long long)k.tiles_x * k.tiles_y * k.ncog * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return CODD_EINVAL;
  // (an XCD-contiguous work-item walk LOSES 3 % on this family: its 240-workgroup update-block launches are one dispatch
  // round whose workgroups of one channel group then start in lock-step on one XCD -- DESIGN.md finding 31)
  return CODD_OK;
}

// Activation of one accumulator tile.  The cheap ones are inlined; the transcendental ones (a handful of small head
// layers use them) go through ONE out-of-line copy -- inlined into all A*B unrolled tile epilogues they made the
// kernel 170-220 KB of code.
__device__ __attribute__((noinline)) f32x4 convb_act_slow(f32x4 v, int act) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], act, 1);
  return v;
}
__device__ __forceinline__ f32x4 convb_act(f32x4 v, int act, int co) {
  if (act == CODD_ACT_NONE) return v;
  if (act == CODD_ACT_RELU || (act == CODD_ACT_RELU_CH0 && co == 0)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    return v;
  }
  if (act == CODD_ACT_RELU_CH0) return v;
  if (act == CODD_ACT_LRELU02) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
    return v;
  }
  return convb_act_slow(v, act);
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void convb_wait_vmcnt(int n) {
#define CONVB_W1(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
#define CONVB_W8(N) CONVB_W1(N) CONVB_W1(N + 1) CONVB_W1(N + 2) CONVB_W1(N + 3) CONVB_W1(N + 4) CONVB_W1(N + 5) CONVB_W1(N + 6) CONVB_W1(N + 7)
  switch (n) {
    CONVB_W8(0) CONVB_W8(8) CONVB_W8(16) CONVB_W8(24) CONVB_W8(32) CONVB_W8(40) CONVB_W8(48)
    default: break;  // >= 56 outstanding allowed: nothing to wait for (callers keep their queues shorter)
  }
#undef CONVB_W8
#undef CONVB_W1
}

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1024).
// Inline asm on purpose: hipcc drains vmcnt(0) in front of every LDS access that follows a __builtin LDS-DMA, which
// would serialise the weight stream with the producers' own ds_writes; issued from asm the DMA is invisible to its
// bookkeeping and the producers count the queue themselves (convb_wait_vmcnt).  M0 is written and restored inside the
// statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void convb_dma16(const uint4* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Compile-time interleave of one k-step: the NM MFMAs of the current fragment set with the NR LDS reads of the next
// one (q_i MFMAs, then one read), so that the reads are in flight under the matrix pipe instead of in front of it.
template <int I, int NR, int NM>
struct ConvbSched {
  static __device__ __forceinline__ void run() {
    constexpr int q = ((I + 1) * NM) / NR - (I * NM) / NR;
    if constexpr (q > 0) __builtin_amdgcn_sched_group_barrier(0x008, q, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // DS read
    if constexpr (I + 1 < NR) ConvbSched<I + 1, NR, NM>::run();
  }
};

template <int PGW, int CGW, int A, int B, int TERMS, int OUTF, int KS>
__global__ __launch_bounds__((PGW * CGW * KS + CONVB_NWP) * 64) void conv_bf16_kernel(const ConvB k) {
  constexpr int NWT = PGW * CGW;          // accumulator-tile sets (one per consumer wave, or per k-split pair)
  constexpr int NWC = NWT * KS;           // consumer waves
  constexpr int NTP = CONVB_NWP * 64;     // producer threads
  constexpr int NPL = CODD_TERMS_PLANES(TERMS); // precision planes (3: hi | lo bf16; 1: bf16; 16: fp16; 48: hi | lo fp16)
  constexpr bool F16 = CODD_TERMS_IS_F16(TERMS);
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* wl = smem4;                             // ring of nring weight buffers (the DMA runs nring - 1 chunks ahead)
  uint4* il = smem4 + k.nring * k.wslots;        // ring of nring input buffers
  int* etab = (int*)(il + k.nring * k.ibuf16);   // entry table: slot offset of (k-step, g) inside an input plane
  const codd_conv_params& p = k.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  const int tx = bid % k.tiles_x; bid /= k.tiles_x;
  const int ty = bid % k.tiles_y; bid /= k.tiles_y;
  const int cog = bid % k.ncog;
  const int b = bid / k.ncog;

  if (wave >= NWC) {
    // =============================== producers ===============================
    const int pt = tid - NWC * 64;
    const int pw = wave - NWC;                      // producer wave index
    const int nwv = k.wslots >> 6, niv = k.ibuf16 >> 6;  // 1 KiB pieces of a weight / an input image
    const int nd = (nwv - pw + CONVB_NWP - 1) / CONVB_NWP + (niv - pw + CONVB_NWP - 1) / CONVB_NWP;  // pieces per chunk, this wave
    const unsigned wl_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint4*)wl;
    const unsigned il_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint4*)il;
    // Chunk CH -> ring slot BUF: weights are a linear copy; an input piece is 64 consecutive LDS slots
    // s = (plane, octet, pixel) gathered from per-lane record addresses
#define BF_DMA_W(CH, BUF)                                                                                 \
  {                                                                                                       \
    const uint4* wsrc_ = (const uint4*)p.wpacked + ((size_t)(cog * k.nchunks + (CH))) * k.wslots + lane;  \
    const unsigned wdst_ = wl_lds + (unsigned)(BUF) * (unsigned)k.wslots * 16u;                           \
    for (int i_ = pw; i_ < nwv; i_ += CONVB_NWP)                                                          \
      convb_dma16(wsrc_ + i_ * 64, __builtin_amdgcn_readfirstlane(wdst_ + (unsigned)i_ * 1024u));         \
  }
#ifndef CONVB_NO_PRODUCER
    // the weights of chunk 0 need no per-lane address arithmetic: their DMA is in flight before anything else happens
    BF_DMA_W(0, 0);
#endif
    // first record of this tile in an octet plane of the split input (its border makes every halo tile in-bounds)
    const int oct_rec = p.xs_hp * p.xs_wp;  // records per octet plane
    const uint4* xs0 = (const uint4*)p.xs + (size_t)b * NPL * k.xplane + (size_t)p.xs_o8 * oct_rec +
                       (size_t)(ty * k.th * p.sy + p.xs_bt - p.pad_t) * p.xs_wp + tx * k.tw * p.sx + p.xs_bl - p.pad_l;
    // per-lane record offset of this wave's input pieces (chunk-independent; the first CONVB_MAXP pieces live in
    // registers, configurations with more fall back to recomputing the divisions per piece)
    auto piece_off = [&](int i_) -> int {
      const int s_ = i_ * 64 + lane;
      const int pl_ = convb_div(s_, k.m_iplane16, k.iplane16), r_ = s_ - pl_ * k.iplane16;
      const int oc_ = convb_div(r_, k.m_os16, k.os16);
      int px_ = r_ - oc_ * k.os16;
      px_ = px_ < k.npix ? px_ : 0;  // slots in the octet padding re-read pixel 0 (never consumed)
      const int y_ = convb_div(px_, k.m_twi, k.twi), x_ = px_ - y_ * k.twi;
      return pl_ * k.xplane + oc_ * oct_rec + y_ * p.xs_wp + x_;
    };
    int ioff[CONVB_MAXP];
#pragma unroll
    for (int q_ = 0; q_ < CONVB_MAXP; ++q_) ioff[q_] = piece_off(pw + q_ * CONVB_NWP);
#define BF_DMA_I(CH, BUF)                                                                                 \
  {                                                                                                       \
    const uint4* xsrc_ = xs0 + (size_t)(CH) * k.noct * oct_rec;                                           \
    const unsigned idst_ = il_lds + (unsigned)(BUF) * (unsigned)k.ibuf16 * 16u;                           \
    _Pragma("unroll") for (int q_ = 0; q_ < CONVB_MAXP; ++q_) {                                           \
      const int i_ = pw + q_ * CONVB_NWP;                                                                 \
      if (i_ < niv) convb_dma16(xsrc_ + ioff[q_], __builtin_amdgcn_readfirstlane(idst_ + (unsigned)i_ * 1024u)); \
    }                                                                                                     \
    for (int i_ = pw + CONVB_MAXP * CONVB_NWP; i_ < niv; i_ += CONVB_NWP)                                 \
      convb_dma16(xsrc_ + piece_off(i_), __builtin_amdgcn_readfirstlane(idst_ + (unsigned)i_ * 1024u));   \
  }
#define BF_DMA(CH, BUF) { BF_DMA_W(CH, BUF) BF_DMA_I(CH, BUF) }
    // Phase c (the consumers compute chunk c from ring slot c % nring): issue the DMA of chunk c + nring - 1 into the
    // slot last read in phase c-1, then wait until the DMA of chunk c+1 has landed -- vmcnt counts in order, so "at
    // most the younger chunks' pieces outstanding" -- and meet the consumers at the barrier.  With a ring of three
    // every global access has two full phases to complete (round trip measured here ~1.7 us, a phase ~1.4 us).
#ifndef CONVB_NO_PRODUCER
    const int ahead = k.nring - 1;
    BF_DMA_I(0, 0);
    if (ahead > 1 && k.nchunks > 1) BF_DMA(1, 1);
#endif
    // entry table (slot offset of every (tap, octet) entry inside an input plane), built while the ring fills
    for (int e = pt; e < (k.nk + 2) * 4; e += NTP) {  // spare rows: the consumers fetch one (k-split: two) k-steps ahead
      const int tap = convb_div(e, k.m_noct, k.noct), oct = e - tap * k.noct;
      int off = 0;
      if (e < k.nk * 4 && tap < k.ntaps) {
        const int ky = convb_div(tap, k.m_kw, p.kw), kx = tap - ky * p.kw;
        if (p.dil2 > 0 && ky < k.khe)  // the small-dilation tap set, centred inside the halo of the large one
          off = oct * k.os16 + ((k.khe >> 1) * (p.dil_y - p.dil2) + ky * p.dil2) * k.twi +
                (p.kw >> 1) * (p.dil_x - p.dil2) + kx * p.dil2;
        else
          off = oct * k.os16 + (ky - (p.dil2 > 0 ? k.khe : 0)) * p.dil_y * k.twi + kx * p.dil_x;
      }
      etab[e] = off;
    }
#ifndef CONVB_NO_PRODUCER
    convb_wait_vmcnt(ahead > 1 && k.nchunks > 1 ? nd : 0);  // chunk 0 landed (chunk 1 may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // entry table
    __builtin_amdgcn_s_barrier();
    int slot = ahead == k.nring ? 0 : ahead;  // ring slot of chunk c + ahead
    for (int c = 0; c < k.nchunks; ++c) {
      if (c + ahead < k.nchunks) BF_DMA(c + ahead, slot);
      slot = slot + 1 == k.nring ? 0 : slot + 1;
      // chunk c + 1 must have landed; the chunks issued after it (c + 2 .. c + ahead, where they exist) may fly on
      int fly_ = k.nchunks - (c + 2);
      fly_ = fly_ < 0 ? 0 : (fly_ > ahead - 1 ? ahead - 1 : fly_);
      convb_wait_vmcnt(fly_ * nd);
      __builtin_amdgcn_s_barrier();
    }
#endif
#undef BF_DMA
#undef BF_DMA_W
#undef BF_DMA_I
    return;
  }

  // ================================= consumers =================================
  const int g = lane >> 4, j = lane & 15;
  // KS = 2: waves w and w + NWT (same SIMD: a workgroup's waves cycle over the four SIMDs) share the tile set of
  // w and take alternate k-steps of every chunk
  const int kpart = KS == 1 ? 0 : wave / NWT, wset = KS == 1 ? wave : wave - kpart * NWT;
  const int cgi = wset % CGW, pgi = wset / CGW;
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
    pbase[a] = prow * p.sy * k.twi + pcol * p.sx;
  }
  const int woff = g * k.nco + cgi * B * 16 + j;  // + kstep * 4 * nco + m * 16
  const int wstep = 4 * k.nco;

  struct Frag { bf16x8 ah[B], al[B], bh[A], bl[A]; };
  // reads the fragments of k-step KS with the entry offset fetched one load earlier, and fetches the next offset
  // (the table has one spare row), so that no ds_read_b128 waits for a dependent table read
#define BF_LOAD(F, KSTP)                                                                                   \
  {                                                                                                       \
    const int eo_ = eo;                                                                                   \
    eo = etab[((KSTP) + KS) * 4 + g];                                                                   \
    const uint4* wp_ = wb + woff + (KSTP) * wstep;                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < B; ++m_) {                                                    \
      F.ah[m_] = __builtin_bit_cast(bf16x8, wp_[m_ * 16]);                                                \
      if (NPL == 2) F.al[m_] = __builtin_bit_cast(bf16x8, wp_[k.wplane16 + m_ * 16]);                   \
    }                                                                                                     \
    _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) {                                                    \
      F.bh[a_] = __builtin_bit_cast(bf16x8, ib[eo_ + pbase[a_]]);                                         \
      if (NPL == 2) F.bl[a_] = __builtin_bit_cast(bf16x8, ib[k.iplane16 + eo_ + pbase[a_]]);            \
    }                                                                                                     \
  }
  // one MFMA on two 16-byte fragments: bf16 or (TERMS = 16) fp16 operands, fp32 accumulate
  auto mma = [](bf16x8 x, bf16x8 y, f32x4 c) -> f32x4 {
    if constexpr (F16)
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
  };
  // term-major order: consecutive MFMAs go to different accumulators (small terms first)
#define BF_MFMA(F)                                                                                        \
  {                                                                                                       \
    if (NPL == 2) {                                                                                       \
      _Pragma("unroll") for (int a = 0; a < A; ++a) _Pragma("unroll") for (int m = 0; m < B; ++m)         \
        acc[a][m] = OUTF ? mma(F.al[m], F.bh[a], acc[a][m]) : mma(F.bh[a], F.al[m], acc[a][m]);           \
      _Pragma("unroll") for (int a = 0; a < A; ++a) _Pragma("unroll") for (int m = 0; m < B; ++m)         \
        acc[a][m] = OUTF ? mma(F.ah[m], F.bl[a], acc[a][m]) : mma(F.bl[a], F.ah[m], acc[a][m]);           \
    }                                                                                                     \
    _Pragma("unroll") for (int a = 0; a < A; ++a) _Pragma("unroll") for (int m = 0; m < B; ++m)           \
      acc[a][m] = OUTF ? mma(F.ah[m], F.bh[a], acc[a][m]) : mma(F.bh[a], F.ah[m], acc[a][m]);             \
  }

  __syncthreads();  // chunk 0 and the entry table are in LDS
#ifndef CONVB_NO_CONSUMER
  const int klast = k.nk - 1;
  constexpr int NRD = 1 + NPL * (A + B), NMF = (NPL == 2 ? 3 : 1) * A * B;  // LDS reads / MFMAs per k-step
  int wsel = 0;
  for (int ch = 0; ch < k.nchunks; ++ch) {
    const uint4* wb = wl + wsel * k.wslots;  // ring slot ch % nring
    const uint4* ib = il + wsel * k.ibuf16;
    wsel = wsel + 1 == k.nring ? 0 : wsel + 1;
    // software pipeline over the k-steps, two register sets, no branch inside the loop body (hipcc can then count
    // the outstanding LDS reads instead of draining them): the fragments of step s+1 are in flight while the MFMAs
    // of step s issue.  A clamped (redundant) load replaces the conditional one at the end of the chunk.
    // k-split: this wave's k-steps are kfirst, kfirst + 2, ... (the odd one of an odd count alternates between the
    // partners chunk by chunk)
    const int kfirst = KS == 1 ? 0 : ((kpart + ch) & 1);
    if (KS == 1 || kfirst < k.nk) {
      const int mylast = KS == 1 ? klast : kfirst + ((klast - kfirst) / KS) * KS;
      Frag f0, f1;
      int eo = etab[kfirst * 4 + g];
      BF_LOAD(f0, kfirst);
      int ks = kfirst;
      for (; ks + KS < k.nk; ks += 2 * KS) {
#ifdef CONVB_NO_LDSREAD
        if (ch == 0 && ks == kfirst)
#endif
        BF_LOAD(f1, ks + KS);
        BF_MFMA(f0);
#ifndef CONVB_NO_SCHED
        ConvbSched<0, NRD, NMF>::run();
#endif
#ifdef CONVB_NO_LDSREAD
        if (ch == 0 && ks == kfirst)
#endif
        BF_LOAD(f0, ks + 2 * KS < mylast ? ks + 2 * KS : mylast);
        BF_MFMA(f1);
#ifndef CONVB_NO_SCHED
        ConvbSched<0, NRD, NMF>::run();
#endif
      }
      if (ks < k.nk) BF_MFMA(f0);
    }
#ifndef CONVB_NO_BARRIER
    __syncthreads();  // every consumer is done with buffer ch & 1; the producers have filled the other one
#endif
  }
#endif
#undef BF_LOAD
#undef BF_MFMA
  if constexpr (KS == 2) {
    // partial sums of the pair: tile t = a * B + m is finished (and stored) by the partner with kpart == (t & 1), so
    // each wave ships half of its tiles through the ring (idle now: the last chunk's barrier is behind us, the
    // producers have retired) and runs half of the epilogue
    f32x4* red = (f32x4*)smem4 + (size_t)wset * (A * B) * 64 + lane;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int m = 0; m < B; ++m)
        if (((a * B + m) & 1) != kpart) red[(a * B + m) * 64] = acc[a][m];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int m = 0; m < B; ++m)
        if (((a * B + m) & 1) == kpart) acc[a][m] += red[(a * B + m) * 64];
  }

#ifdef CONVB_NO_EPILOGUE
  {  // dev ablation: one store per lane keeps the accumulators alive
    float s_ = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int m = 0; m < B; ++m) s_ += acc[a][m][0] + acc[a][m][1] + acc[a][m][2] + acc[a][m][3];
    p.out[(size_t)blockIdx.x * 256 + tid] = s_;
    return;
  }
#endif
  if constexpr (OUTF == 1) {
    // Split-record epilogue (the output IS the next convolution's input, codd_split_bf16 layout): the MFMA was issued
    // weights-first, D[m = co 4g + r][n = pixel j], so a lane holds 4 consecutive CHANNELS of one pixel; lanes g and
    // g ^ 1 (16 apart) own the two halves of a channel octet.  After bias + activation each lane splits its 4 values
    // into (hi, lo) bf16, the pair swaps one half through ds_bpermute, the even lane stores the 16-byte HI record and
    // the odd lane the LO record: 16 pixels j -> 256 contiguous bytes per (octet, plane).  No fp32 tensor is written.
    const int orec = p.xso_hp * p.xso_wp;
    uint4* xo = (uint4*)p.xso + (size_t)b * NPL * p.xso_c8 * orec;
    const bool odd = g & 1;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const int u = pgi * A + a;
      const int prow = u / k.xb;
      const int oy = ty * k.th + prow;
      const int ox = tx * k.tw + (u - prow * k.xb) * 16 + j;
      const bool inb = u < k.pu && oy < p.Hout && ox < p.Wout;
#pragma unroll
      for (int m = 0; m < B; ++m) {
        if (KS == 2 && ((a * B + m) & 1) != kpart) continue;  // the k-split partner's tile
        const int co0 = (cog * CGW * B + cgi * B + m) * 16 + 4 * g;  // this lane's first channel
        f32x4 v = acc[a][m];
        if (p.bias) {
          if (co0 + 3 < k.cout_eff) v += *(const f32x4*)(p.bias + co0);  // co0 is a multiple of 4: 16-byte aligned
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += co0 + r < k.cout_eff ? p.bias[co0 + r] : 0.f;
        }
        int rco = co0;  // first channel of the lane's record half
        if (p.gate) {
          // ConvGRU gate epilogues on channel-quad fp32 tensors: a lane's 4 channels of one pixel are ONE 16-byte
          // access, 16 lanes j = 256 contiguous bytes.  The 16-channel tile (and with it every branch) is wave-uniform.
          const int hw_ = p.Hout * p.Wout, pix_ = oy * p.Wout + ox;
          auto c4 = [&](const float* base, int ctot, int c) {
            return (f32x4*)(base + (((size_t)b * (ctot >> 2) + (c >> 2)) * hw_ + pix_) * 4);
          };
          const bool live = inb && co0 < k.cout_eff;
          const int ctile = co0 & ~15;
          if (p.gate == 1) {
            if (live) *c4(p.out, p.out_ctot, p.out_coff + co0) = v;
            continue;
          }
          const int G = p.gate == 2 ? k.cout_eff / 3 : k.cout_eff;
          if (p.gate == 2) {
            if (live) v += *c4(p.res1.ptr, p.res1.ctot, p.res1.coff + co0);
            if (ctile >= 2 * G) {  // q's input stream
              if (live) *c4(p.out, p.out_ctot, p.out_coff + co0 - G) = v;
              continue;
            }
            if (live) v += *c4(p.res2.ptr, p.res2.ctot, p.res2.coff + co0);
            v = convb_act_slow(v, CODD_ACT_SIGMOID);
            if (ctile < G) {  // z
              if (live) *c4(p.out, p.out_ctot, p.out_coff + co0) = v;
              continue;
            }
            rco = co0 - G;  // r * h -> records
            if (live) v *= *c4(p.post.ptr, p.post.ctot, p.post.coff + rco);
          } else {
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f}, h = z;
            if (live) {
              v += *c4(p.res1.ptr, p.res1.ctot, p.res1.coff + G + co0);
              z = *c4(p.res1.ptr, p.res1.ctot, p.res1.coff + co0);
              h = *c4(p.post.ptr, p.post.ctot, p.post.coff + co0);
            }
            v = convb_act_slow(v, CODD_ACT_TANH);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (1.f - z[r]) * h[r] + z[r] * v[r];
            if (live) *c4(p.out, p.out_ctot, p.out_coff + co0) = v;
          }
        } else {
          v = convb_act(v, p.act, 1);  // (CODD_ACT_RELU_CH0 is a per-channel activation of fp32 outputs only)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = co0 + r < k.cout_eff ? v[r] : 0.f;  // channel padding of the octet stays 0
        unsigned hi2[2], lo2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if constexpr (F16) {  // fp16 records: one plane, or hi | lo (xso_terms == terms: convb_geometry)
            const _Float16 h0 = (_Float16)v[2 * q], h1 = (_Float16)v[2 * q + 1];
            const _Float16 l0 = (_Float16)(v[2 * q] - (float)h0), l1 = (_Float16)(v[2 * q + 1] - (float)h1);
            hi2[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lo2[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
          } else {
            const __bf16 h0 = (__bf16)v[2 * q], h1 = (__bf16)v[2 * q + 1];
            const __bf16 l0 = (__bf16)(v[2 * q] - (float)h0), l1 = (__bf16)(v[2 * q + 1] - (float)h1);
            hi2[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lo2[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
          }
        }
        // even lane needs the partner's hi, odd lane the partner's lo
        const unsigned s0 = odd ? hi2[0] : lo2[0], s1 = odd ? hi2[1] : lo2[1];
        const unsigned r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
        const uint4 rec = odd ? make_uint4(r0, r1, lo2[0], lo2[1]) : make_uint4(hi2[0], hi2[1], r0, r1);
        const int oct = (rco >> 3) + p.xso_o8;  // both lanes of the pair: same octet (co0 differs by 4)
        if (inb && (co0 & ~7) < k.cout_eff && (!odd || NPL == 2))
          xo[(size_t)(odd ? p.xso_c8 * orec : 0) + ((size_t)oct * p.xso_hp + oy + p.xso_bt) * p.xso_wp + ox + p.xso_bl] = rec;
      }
    }
    return;
  }
  // epilogue.  The MFMA is issued with the im2col fragment as its first operand, so D[m = pixel 4g + r][n = co j]:
  // a lane holds FOUR CONSECUTIVE PIXELS of one output channel -- bias is one value per tile, residual / output
  // accesses are 16-byte vectors (4x fewer memory instructions than a channel-major accumulator layout)
  const int hwout = p.Hout * p.Wout;
  const bool vec = (p.Wout & 3) == 0 && p.store_mode == 0;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int u = pgi * A + a;
    if (u >= k.pu) continue;
    const int prow = u / k.xb;
    const int oy = ty * k.th + prow;
    const int ox = tx * k.tw + (u - prow * k.xb) * 16 + 4 * g;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    const int pix = oy * p.Wout + ox;
#pragma unroll
    for (int m = 0; m < B; ++m) {
      if (KS == 2 && ((a * B + m) & 1) != kpart) continue;  // the k-split partner's tile
      const int co = (cog * CGW * B + cgi * B + m) * 16 + j;
      if (co >= k.cout_eff) continue;
      f32x4 v = acc[a][m];
      if (p.store_mode == 0) {
        if (p.bias) v += p.bias[co];
        float* op = p.out + ((size_t)b * p.out_ctot + p.out_coff + co) * (size_t)hwout + pix;
        if (vec) {  // ox + 3 < Wout: ox and Wout are multiples of 4
          if (p.res1.ptr) v += *(const f32x4*)(view_ptr(p.res1, b, co, hwout) + pix);
          if (p.res2.ptr) v += *(const f32x4*)(view_ptr(p.res2, b, co, hwout) + pix);
          v = convb_act(v, p.act, co);
          if (p.post.ptr) v += *(const f32x4*)(view_ptr(p.post, b, co, hwout) + pix);
#if defined(CONVB_NT_STORE)
          __builtin_nontemporal_store(v, (f32x4*)op);
#elif defined(CONVB_SC1_STORE)
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(op), "v"(v) : "memory");
#else
          *(f32x4*)op = v;
#endif
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ox + r < p.Wout) {
              if (p.res1.ptr) v[r] += view_ptr(p.res1, b, co, hwout)[pix + r];
              if (p.res2.ptr) v[r] += view_ptr(p.res2, b, co, hwout)[pix + r];
            }
          v = convb_act(v, p.act, co);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ox + r < p.Wout) op[r] = v[r] + (p.post.ptr ? view_ptr(p.post, b, co, hwout)[pix + r] : 0.f);
        }
      } else {  // ConvTranspose2d k=2 s=2: co = (a2*2+b2)*Cout + c
        const int q = co / p.Cout, c = co - q * p.Cout;
        if (p.bias) v += p.bias[c];
        v = convb_act(v, p.act, c);
        const int W2 = 2 * p.Wout;
        float* op = p.out + ((size_t)b * p.out_ctot + p.out_coff + c) * (size_t)(4 * hwout) +
                    (size_t)(2 * oy + (q >> 1)) * W2 + (q & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ox + r < p.Wout) op[2 * (ox + r)] = v[r];
      }
    }
  }
}

// ---- instantiation list: X(PGW, CGW, A, B, KS), each for TERMS = 1 and 3 and both output forms ----------------
//   (2,2,5,2): 9/10-unit tiles x 64 channels (the 72x120 GRU maps: 256 workgroups of 144 px x 64 co)
//   (4,1,4,4) / (4,1,4,2) / (4,1,4,1): 16-unit tiles (8 x 32 px) x 64 / 32 / 16 channels
//   (4,1,2,2) / (4,1,2,1): 8-unit tiles (4 x 32 or 8 x 16 px) x 32 / 16 channels (small maps)
//   (4,1,8,1): 32-unit tiles (16 x 32 px) x 16 channels (full-resolution 16-channel layers)
//   (4,1,3,4) / (4,1,3,2): 12-unit tiles (12 x 16 or 6 x 32 px) x 64 / 32 channels: 192 workgroups = ONE dispatch
//   round on a 72x120 map with 256 / 128 output channels (the 16-unit tiles give 136-160, the 10-unit ones 216-256
//   with a worse LDS-read : MFMA ratio)
//   KS = 2 (k-split pairs, 8 consumer waves = 2 per SIMD; needs <= 168 registers): the tiles of the update block
//   (4,2,4,2) / (4,2,3,2): 16- / 12-unit tiles x 64 channels on 8 consumer waves that split the TILE instead
#define CONVB_GROUP_A(X) X(2, 2, 5, 2, 1)
#define CONVB_GROUP_B(X) X(4, 1, 4, 4, 1)
#define CONVB_GROUP_C(X) X(4, 1, 4, 2, 1)
#define CONVB_GROUP_D(X) X(4, 1, 4, 1, 1) X(4, 1, 8, 1, 1)
#define CONVB_GROUP_E(X) X(4, 1, 2, 2, 1) X(4, 1, 2, 1, 1)
#define CONVB_GROUP_F(X) X(4, 1, 3, 4, 1) X(4, 1, 3, 2, 1)
#define CONVB_GROUP_G(X) X(2, 2, 5, 2, 2)
#define CONVB_GROUP_H(X) X(4, 1, 4, 2, 2)
#define CONVB_GROUP_I(X) X(4, 1, 3, 2, 2) X(4, 1, 3, 4, 2)
#define CONVB_GROUP_J(X) X(4, 2, 4, 2, 1)
#define CONVB_GROUP_K(X) X(4, 2, 3, 2, 1)
//   (8,1,4,4): 32-unit tiles (16 x 32 px) x 64 channels on 8 consumer waves of 64 px x 64 co each (the largest
//   per-wave tile: 8 fragment reads per 16 MFMAs against 6 per 8 for (4,2,4,2); two-deep ring)
#define CONVB_GROUP_L(X) X(8, 1, 4, 4, 1)
#define CONVB_ALL(X)                                                                                   \
  CONVB_GROUP_A(X) CONVB_GROUP_B(X) CONVB_GROUP_C(X) CONVB_GROUP_D(X) CONVB_GROUP_E(X) CONVB_GROUP_F(X) \
  CONVB_GROUP_G(X) CONVB_GROUP_H(X) CONVB_GROUP_I(X) CONVB_GROUP_J(X) CONVB_GROUP_K(X) CONVB_GROUP_L(X)
#define CONVB_DECLARE(PGW, CGW, A, B, KS)                                                        \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, 0, KS>(const ConvB);      \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, 0, KS>(const ConvB);      \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, 1, KS>(const ConvB);      \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, 1, KS>(const ConvB);      \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 16, 0, KS>(const ConvB);     \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 16, 1, KS>(const ConvB);     \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 48, 0, KS>(const ConvB);     \
  extern template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 48, 1, KS>(const ConvB);
#define CONVB_DEFINE(PGW, CGW, A, B, KS)                                                         \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, 0, KS>(const ConvB);             \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, 0, KS>(const ConvB);             \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 1, 1, KS>(const ConvB);             \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 3, 1, KS>(const ConvB);             \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 16, 0, KS>(const ConvB);            \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 16, 1, KS>(const ConvB);            \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 48, 0, KS>(const ConvB);            \
  template __global__ void conv_bf16_kernel<PGW, CGW, A, B, 48, 1, KS>(const ConvB);
