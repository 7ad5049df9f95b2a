// Shared helpers for the gfx950 kernels of libcodd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codd_hip.h"

#define CODD_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e_ = hipGetLastError();           \
    if (e_ != hipSuccess) return (int)e_;        \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

int codd_opt(int key);  // current value of a CODD_OPT_* option (stereo.hip)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float act_apply(float v, int act, int co) {
  switch (act) {
    case CODD_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case CODD_ACT_RELU: return fmaxf(v, 0.f);
    case CODD_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case CODD_ACT_TANH: return tanhf(v);
    case CODD_ACT_MISH: {
      float sp = v > 20.f ? v : log1pf(expf(v));
      return v * tanhf(sp);
    }
    case CODD_ACT_RELU_CH0: return co == 0 ? fmaxf(v, 0.f) : v;
    default: return v;
  }
}

// 8 consecutive channels of one pixel -> the split-bf16 record(s) of a codd_xs_view (same rounding as
// split_bf16_kernel: hi = RNE(v), lo = RNE(v - hi))
typedef __bf16 codd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 codd_f16x8 __attribute__((ext_vector_type(8)));
// one fp32 value -> the 16-bit element of a record plane: bf16 (terms 1 | 3) or IEEE fp16 (terms 16), both RNE
__device__ __forceinline__ unsigned short xs_elem16(float v, bool f16) {
  return f16 ? __builtin_bit_cast(unsigned short, (_Float16)v) : __builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ void xs_store8(const codd_xs_view& d, int b, int oct, int y, int x, const float* v) {
  const size_t per = (size_t)d.c8 * d.hp * d.wp;
  const int planes = CODD_TERMS_PLANES(d.terms);
  codd_bf16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)v[i];
    h[i] = hh;
    l[i] = (__bf16)(v[i] - (float)hh);
  }
  uint4* dst = (uint4*)d.ptr + (size_t)b * planes * per + ((size_t)(d.o8 + oct) * d.hp + (y + d.bt)) * d.wp + (x + d.bl);
  if (CODD_TERMS_IS_F16(d.terms)) {
    codd_f16x8 q, ql;
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
static inline bool xs_view_ok(const codd_xs_view& d, int C, int H, int W) {
  return d.ptr && !((uintptr_t)d.ptr & 15) && CODD_TERMS_OK(d.terms) && d.o8 >= 0 && 8 * (d.c8 - d.o8) >= C &&
         d.bt >= 0 && d.bl >= 0 && d.hp >= d.bt + H && d.wp >= d.bl + W;
}

// XCD-contiguous work-item walk for kernels whose NEIGHBOURING workgroups share input (halo rows / columns): the
// dispatcher places workgroup L (linear id, x fastest) on XCD L % 8, so neighbours land on eight different L2s and
// every shared byte is fetched up to 8 times from HBM.  Workgroup L takes item start[L % 8] + L / 8 instead: every
// XCD walks ONE contiguous range of the x-fastest item order.  (Same mapping as conv_kernel.h conv_xcd_item.)
__device__ __forceinline__ int codd_xcd_item(int L, int n) {
  const int q = n >> 3, r = n & 7, x = L & 7, idx = L >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
}
// (bx, by, bz) of this workgroup's item under the walk above, for a 3-D grid
__device__ __forceinline__ void codd_xcd_block(int& bx, int& by, int& bz) {
  const int nx = gridDim.x, ny = gridDim.y;
  int it = codd_xcd_item(blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), nx * ny * gridDim.z);
  bx = it % nx; it /= nx;
  by = it % ny;
  bz = it / ny;
}

// wave64 butterfly helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
