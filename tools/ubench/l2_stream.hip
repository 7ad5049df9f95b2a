// micro-benchmark: how fast can every CU pull the SAME small buffer (weights chunk) from L2?
// V0: global_load_dwordx4 into registers; V1: global_load_lds_dwordx4 (LDS-DMA)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k(const f32x4* src, float* out, int iters, int chunk4, int nchunks, int shared) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, wave = tid >> 6;
  f32x4 acc = {0, 0, 0, 0};
  const f32x4 __attribute__((address_space(1)))* g =
      (const f32x4 __attribute__((address_space(1)))*)(uintptr_t)(src + (shared ? 0 : (size_t)(blockIdx.x % 64) * chunk4 * nchunks));
  for (int it = 0; it < iters; ++it) {
    const f32x4 __attribute__((address_space(1)))* s = g + (size_t)(it % nchunks) * chunk4;
    for (int e0 = 0; e0 < chunk4; e0 += 256) {
      const int e = e0 + tid;
      if (V == 0) { if (e < chunk4) { f32x4 v = s[e]; acc += v; } }
      else { if (e < chunk4) __builtin_amdgcn_global_load_lds(s + e, (__attribute__((address_space(3))) void*)(sm + 4 * (e0 + wave * 64)), 16, 0, 0); }
    }
    if (V == 1) __syncthreads();
  }
  if (V == 1) acc[0] = sm[tid];
  out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int V>
void run(int blocks, int chunk4, int shared, const f32x4* src, float* out) {
  int iters = 400, nchunks = 32;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<V><<<blocks, 256, 49152>>>(src, out, iters, chunk4, nchunks, shared); hipDeviceSynchronize();
  hipEventRecord(a); k<V><<<blocks, 256, 49152>>>(src, out, iters, chunk4, nchunks, shared); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * iters * chunk4 * 16.0;
  printf("V%d blocks %4d chunk %5d B %s: %.1f us  %.2f TB/s total  %.1f B/clk/CU (2.4 GHz)\n", V, blocks, chunk4 * 16, shared ? "shared " : "per-blk", ms * 1e3,
         bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9));
}
int main() {
  f32x4* src; float* out;
  hipMalloc(&src, (size_t)64 * 32 * 2048 * 16); hipMemset(src, 0, (size_t)64 * 32 * 2048 * 16); hipMalloc(&out, 2048 * 256 * 4);
  for (int blocks : {256, 768}) for (int shared : {1, 0}) { run<0>(blocks, 1440, shared, src, out); run<1>(blocks, 1440, shared, src, out); }
}
