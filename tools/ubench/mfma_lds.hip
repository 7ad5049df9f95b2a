// micro-benchmark: what bounds the conv kernel's MFMA phase?  V0: MFMAs only; V1: ds_read_b32 operands
// (current layout, software-pipelined groups of 4 k-steps); V2: ds_read_b128 operands (4 k-steps per read)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MB 4
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int ngroups, int wrow, int chs) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
  for (int i = tid; i < 16384; i += 256) sm[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x4 acc[MB];
  for (int m = 0; m < MB; ++m) acc[m] = (f32x4){0, 0, 0, 0};
  float a0[4][MB], b0[4], a1[4][MB], b1[4];
  for (int s = 0; s < 4; ++s) { for (int m = 0; m < MB; ++m) a0[s][m] = a1[s][m] = 1.f + s; b0[s] = b1[s] = 0.5f; }
  int off = 0;
#define LOAD(A, B)                                                                                         \
  if (V == 1) {                                                                                            \
    const float* wp = sm + j + g * wrow + off;                                                             \
    const float* ip = sm + 12288 + g * chs + j + (tid >> 6) * 20 + (off >> 3);                             \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                        \
      _Pragma("unroll") for (int m = 0; m < MB; ++m) A[s][m] = wp[s * 4 * wrow + m * 16];                  \
      B[s] = ip[s * 4 * chs];                                                                              \
    }                                                                                                      \
  } else if (V == 2) {                                                                                     \
    const float4* wp = (const float4*)(sm + off) + lane;                                                   \
    const float4* ip = (const float4*)(sm + 12288 + (off >> 3)) + lane + (tid >> 6) * 80;                  \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                       \
      const float4 t = wp[m * 64];                                                                         \
      A[0][m] = t.x; A[1][m] = t.y; A[2][m] = t.z; A[3][m] = t.w;                                          \
    }                                                                                                      \
    const float4 u = ip[0];                                                                                \
    B[0] = u.x; B[1] = u.y; B[2] = u.z; B[3] = u.w;                                                        \
  }                                                                                                        \
  off = (off + 1280) & 8191;
#define MMA_FIRST(A, B) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][0], B[0], acc[0], 0, 0, 0);
#define MMA_REST(A, B)                                                                                     \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int m = 0; m < MB; ++m)             \
    if (s || m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][m], B[s], acc[m], 0, 0, 0);
  LOAD(a0, b0);
  for (int n = 0; n < ngroups; n += 2) {
    MMA_FIRST(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    LOAD(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    MMA_REST(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    MMA_FIRST(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    LOAD(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    MMA_REST(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = 0;
  for (int m = 0; m < MB; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 256 + tid] = r;
}
template <int V>
void run(int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int ngroups = 1200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<V><<<blocks, 256, 65536>>>(out, ngroups, 80, 144);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<V><<<blocks, 256, 65536>>>(out, ngroups, 80, 144);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double fl = (double)blocks * 4 * ngroups * 16 * 2048.0;
  printf("V%d blocks %4d: %.1f us  %.1f TFLOP/s\n", V, blocks, ms * 1e3, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int blocks : {256, 512}) { run<0>(blocks); run<1>(blocks); run<2>(blocks); }
}
