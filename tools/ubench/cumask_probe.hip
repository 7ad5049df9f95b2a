// Dev probe: does hipExtStreamCreateWithCUMask confine kernels (and hipGraph launches) to a subset of the XCDs on
// MI355X / ROCm 7.2, and how do mask bits map to XCDs?   hipcc --offload-arch=gfx950 cumask_probe.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>

__global__ void where_kernel(unsigned* xcc_hist, unsigned* cu_seen, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    atomicAdd(&xcc_hist[xcc & 15], 1u);
    const unsigned cu = (hwid >> 8) & 15, sh = (hwid >> 12) & 1, se = (hwid >> 13) & 7;  // CU_ID, SH_ID, SE_ID
    atomicOr(&cu_seen[(xcc & 15) * 8 + se], 1u << (sh * 16 + cu));
  }
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) xcc_hist[15] = 1;
}

static void report(const char* tag, hipStream_t s, bool graph) {
  unsigned *hist, *seen;
  hipMalloc(&hist, 64); hipMalloc(&seen, 16 * 8 * 4);
  hipMemset(hist, 0, 64); hipMemset(seen, 0, 16 * 8 * 4);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (graph) {
    hipStream_t cap; hipStreamCreate(&cap);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal);
    where_kernel<<<4096, 256, 0, cap>>>(hist, seen, 20000);
    hipStreamEndCapture(cap, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
  } else {
    hipEventRecord(e0, s);
    where_kernel<<<4096, 256, 0, s>>>(hist, seen, 20000);
    hipEventRecord(e1, s);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned h[16], c[128];
  hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost); hipMemcpy(c, seen, 512, hipMemcpyDeviceToHost);
  int cus = 0;
  for (int i = 0; i < 128; ++i) cus += __builtin_popcount(c[i]);
  printf("%-34s %s: %.3f ms, distinct CUs %3d, blocks per XCC:", tag, graph ? "graph " : "kernel", ms, cus);
  for (int i = 0; i < 8; ++i) printf(" %4u", h[i]);
  printf("\n");
}

int main() {
  hipStream_t s0; hipStreamCreate(&s0);
  report("no mask", s0, false);
  report("no mask", s0, true);
  struct { const char* tag; int kind; } tests[] = {{"bits 0..127", 0}, {"bits i%8<4", 1}, {"bits 128..255", 2}, {"bits i%8>=5", 3}, {"bits 0..31", 4}};
  for (auto& t : tests) {
    uint32_t mask[8];
    memset(mask, 0, sizeof(mask));
    for (int i = 0; i < 256; ++i) {
      bool on = t.kind == 0 ? i < 128 : t.kind == 1 ? (i % 8) < 4 : t.kind == 2 ? i >= 128 : t.kind == 3 ? (i % 8) >= 5 : i < 32;
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", t.tag, hipGetErrorString(e)); continue; }
    report(t.tag, s, false);
    report(t.tag, s, true);
  }
  // two masked streams side by side: do they overlap in time?  (bit i <-> XCC i % 8, CU i / 8 of that XCC: a mask must
  // leave every XCC at least one CU, else it is ignored)
  for (int split = 8; split <= 24; split += 8) {
    uint32_t ma[8], mb[8];
    memset(ma, 0, 32); memset(mb, 0, 32);
    for (int i = 0; i < 256; ++i) { if ((i / 8) < split) ma[i / 32] |= 1u << (i % 32); else mb[i / 32] |= 1u << (i % 32); }
    hipStream_t a, b; hipExtStreamCreateWithCUMask(&a, 8, ma); hipExtStreamCreateWithCUMask(&b, 8, mb);
    unsigned *hist, *seen; hipMalloc(&hist, 64); hipMalloc(&seen, 512);
    hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      if (rep != 1) { hipEventRecord(e0, a); where_kernel<<<4096, 256, 0, a>>>(hist, seen, 20000); hipEventRecord(e1, a); }
      if (rep != 0) { hipEventRecord(f0, b); where_kernel<<<4096, 256, 0, b>>>(hist, seen, 20000); hipEventRecord(f1, b); }
      hipDeviceSynchronize();
      float ma_ = 0, mb_ = 0;
      if (rep != 1) hipEventElapsedTime(&ma_, e0, e1);
      if (rep != 0) hipEventElapsedTime(&mb_, f0, f1);
      printf("A = CUs [0,%d) of every XCC, B = the rest; %s: A %.3f ms, B %.3f ms\n", split, rep == 0 ? "A alone" : rep == 1 ? "B alone" : "A and B together", ma_, mb_);
    }
  }
  return 0;
}
