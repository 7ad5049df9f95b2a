// micro-benchmark: LDS-fed v_mfma_f32_32x32x2_f32 loop (64 couts x 32 pixels per wave, 128-bit operand reads)
// vs the 16x16x4 quad loop of mfma_lds.hip V2.  Per 16 channels: 4 A reads + 2 B reads (b128) for 16 MFMAs of
// 64 cycles (16x16x4 quad: 4 + 1 reads for 16 MFMAs of 32 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int ngroups) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256) sm[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[2];
  for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  int off = 0;
  for (int n = 0; n < ngroups; ++n) {  // one group = 16 channels: two 8-channel b128 reads per operand block
    const float4* wp = (const float4*)(sm + off) + lane;
    const float4* ip = (const float4*)(sm + 12288 + (off >> 3)) + lane + (tid >> 6) * 70;
    float4 a[2][2], b[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { a[0][h] = wp[h * 128]; a[1][h] = wp[h * 128 + 64]; b[h] = ip[h * 64]; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][h].x, b[h].x, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][h].y, b[h].y, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][h].z, b[h].z, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][h].w, b[h].w, acc[m], 0, 0, 0);
      }
    off = (off + 1280) & 8191;
  }
  float r = 0;
  for (int m = 0; m < 2; ++m) for (int q = 0; q < 16; ++q) r += acc[m][q];
  out[blockIdx.x * 256 + tid] = r;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int blocks : {256, 512}) {
    int ngroups = 1200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, 256, 65536>>>(out, ngroups); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<<<blocks, 256, 65536>>>(out, ngroups);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double fl = (double)blocks * 4 * ngroups * 16 * 4096.0;
    printf("32x32x2 quad loop, blocks %4d: %.1f us  %.1f TFLOP/s\n", blocks, ms * 1e3, fl / ms / 1e9);
  }
}
