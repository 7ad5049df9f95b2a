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

// wave64 butterfly helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
