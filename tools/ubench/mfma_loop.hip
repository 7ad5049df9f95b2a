// micro-benchmark: LDS-fed fp32 MFMA inner loop of the conv kernel without any staging
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MB, int NPB, int U>
__global__ __launch_bounds__(256) void k(float* out, int nsteps, int wrow, int chs) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
  for (int i = tid; i < 12288; i += 256) sm[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x4 acc[NPB][MB];
  for (int a = 0; a < NPB; ++a) for (int m = 0; m < MB; ++m) acc[a][m] = (f32x4){0, 0, 0, 0};
  const float* wp = sm + j + g * wrow;
  const float* ip = sm + 8192 + g * chs + j + (tid >> 6) * 20;
  for (int s = 0; s < nsteps; s += U) {
    float av[U][MB], bv[U][NPB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < MB; ++m) av[u][m] = wp[((s + u) % 24) * 4 * wrow % 8000 + m * 16];
#pragma unroll
      for (int a = 0; a < NPB; ++a) bv[u][a] = ip[((s + u) % 16) * 4 * chs % 3000 + a * 16];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < NPB; ++a)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bv[u][a], acc[a][m], 0, 0, 0);
  }
  float r = 0;
  for (int a = 0; a < NPB; ++a) for (int m = 0; m < MB; ++m) r += acc[a][m][0] + acc[a][m][1] + acc[a][m][2] + acc[a][m][3];
  out[blockIdx.x * 256 + tid] = r;
}
template <int MB, int NPB, int U>
void run(int blocks, const char* name) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int nsteps = 4800;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MB, NPB, U><<<blocks, 256, 49152>>>(out, nsteps, 80, 144);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<MB, NPB, U><<<blocks, 256, 49152>>>(out, nsteps, 80, 144);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double fl = (double)blocks * 4 * nsteps * MB * NPB * 2048.0;
  printf("%s blocks %4d: %.1f us  %.1f TFLOP/s\n", name, blocks, ms * 1e3, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 768}) {
    run<4, 1, 1>(blocks, "MB4 NPB1 U1");
    run<4, 1, 2>(blocks, "MB4 NPB1 U2");
    run<4, 1, 4>(blocks, "MB4 NPB1 U4");
    run<2, 1, 1>(blocks, "MB2 NPB1 U1");
    run<2, 2, 1>(blocks, "MB2 NPB2 U1");
    run<4, 2, 1>(blocks, "MB4 NPB2 U1");
    run<4, 2, 2>(blocks, "MB4 NPB2 U2");
  }
}
