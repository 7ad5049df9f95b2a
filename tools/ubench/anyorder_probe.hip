// dev probe: does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (plain stream and captured graph)?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder tools/ubench/anyorder_probe.hip && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(float* out, long long cycles) {
  const long long t0 = wall_clock64();
  float v = threadIdx.x;
  while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[0] = v;
}
static float run(hipStream_t s, int flagB, float* buf, long long cyc) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int r = 0; r < 10; ++r) {
    hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, 0, buf, cyc);
    hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, flagB, buf, cyc);
  }
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 100.f;  // us per pair
}
int main() {
  float* buf; hipMalloc(&buf, 1024);
  hipStream_t s; hipStreamCreate(&s);
  const long long cyc = 5000;  // wall_clock64 ticks at 100 MHz -> 50 us
  run(s, 0, buf, cyc);
  printf("stream  A;B           %.1f us per pair\n", run(s, 0, buf, cyc));
  printf("stream  A;B(anyorder) %.1f us per pair\n", run(s, hipExtAnyOrderLaunch, buf, cyc));
  for (int flag = 0; flag < 2; ++flag) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < 10; ++r) {
      hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, 0, buf, cyc);
      hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, buf, cyc);
    }
    if (hipStreamEndCapture(s, &g) != hipSuccess) { printf("capture failed\n"); return 1; }
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("graph   A;B%s %.1f us per pair\n", flag ? "(anyorder)" : "          ", ms * 100.f);
  }
  return 0;
}
