// calibrates clock64() (s_memtime) against hipEvent time under a pure-MFMA load
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int n) {
  long long t0 = clock64(), w0 = wall_clock64();
  f32x4 acc[4] = {};
  float a = threadIdx.x, b = 0.5f;
  for (int i = 0; i < n; ++i)
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
  long long t1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
int main() {
  float* out; long long* clk; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 16);
  for (int blocks : {256, 768}) for (int n : {20000, 200000}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, 256>>>(out, clk, n); hipDeviceSynchronize();
    hipEventRecord(a); k<<<blocks, 256>>>(out, clk, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("blocks %d n %d: %.1f us, clock64 %lld ticks (%.1f MHz), wall %lld ticks (%.1f MHz), %.1f TF, mfma cycles/inst @clock64 %.2f\n", blocks, n,
           ms * 1e3, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), blocks * 4.0 * n * 4 * 2048 / ms / 1e9, (double)h[0] / (4.0 * n) / (blocks / 256));
  }
}
