// VALU issue cost on gfx950: v_fma_f32 vs v_pk_fma_f32, VGPR vs SGPR operands, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ sc, int n) {
  float a[16]; v2f p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
  const float x = threadIdx.x * 1e-3f;
  const v2f x2 = {x, x + 1.f};
  const float s0 = sc[blockIdx.x & 1], s1 = sc[(blockIdx.x & 1) + 1];  // wave-uniform -> SGPRs
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x2.y));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(x2));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s0), "v"(x));
    } else if (MODE == 3) {
      const v2f s2 = {s0, s1};
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(s2), "v"(x2));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(x2), "v"(x2));
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(p[i]) : "v"(x2), "v"(x2));
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, float* out, float* sc, int per) {
  for (int wgs : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD
    const int n = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<wgs, 256>>>(out, sc, n); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<wgs, 256>>>(out, sc, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst_per_simd = (double)n * per * (wgs / 256);
    printf("%-28s %d waves/SIMD: %.1f us  -> %.2f cycles/instr @2.4GHz per SIMD\n", name, wgs / 256, ms * 1e3, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
}
int main() {
  float *out, *sc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&sc, 64); hipMemset(sc, 0, 64);
  run<0>("v_fma_f32 vgpr", out, sc, 16);
  run<2>("v_fma_f32 sgpr operand", out, sc, 16);
  run<1>("v_pk_fma_f32 vgpr", out, sc, 8);
  run<3>("v_pk_fma_f32 sgpr pair", out, sc, 8);
  run<4>("v_pk_fma_f32 op_sel bcast", out, sc, 8);
  run<5>("v_pk_mul_f32", out, sc, 8);
}
