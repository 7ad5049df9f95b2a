// dev: ablation builds of conv_roll_kernel (csrc/conv_roll.hip compiled with -DROLL_ABL_* switches), timed stand-alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I codd_amd/csrc -mllvm -amdgpu-mfma-vgpr-form [-DROLL_ABL_x] \
//         tools/ubench/roll_ablate.hip -o tools/ubench/roll_ablate_x.bin ;  ./roll_ablate_x.bin C mode B H W rh
#include "../../codd_amd/csrc/conv_roll.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 16, mode = argc > 2 ? atoi(argv[2]) : 1, B = argc > 3 ? atoi(argv[3]) : 1;
  const int H = argc > 4 ? atoi(argv[4]) : 576, W = argc > 5 ? atoi(argv[5]) : 960, rh = argc > 6 ? atoi(argv[6]) : 12;
  const int cin = mode == 2 ? 2 * C : C;
  size_t nin = (size_t)B * cin * H * W, nout = (size_t)B * C * H * W;
  std::vector<float> h(nin);
  for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  float *x, *y, *wa, *wb, *ba;
  hipMalloc(&x, nin * 4); hipMalloc(&y, nout * 4);
  hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice);
  const long long na = codd_roll_packed_size(C, mode == 2 ? 1 : 3, cin), nb = codd_roll_packed_size(C, 3, C);
  hipMalloc(&wa, na * 4); hipMalloc(&wb, nb * 4); hipMalloc(&ba, 256);
  hipMemcpy(wa, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(wb, h.data() + 1000, nb * 4, hipMemcpyHostToDevice);
  hipMemset(ba, 0, 256);
  codd_roll_params p = {};
  p.in0.ptr = x; p.in0.ctot = cin; p.C0 = cin; p.B = B; p.H = H; p.W = W; p.C = C; p.mode = mode;
  p.wA = wa; p.wB = wb; p.bA = ba; p.bB = ba; p.actA = p.actB = CODD_ACT_LRELU02; p.residual = mode == 1;
  p.out = y; p.out_ctot = C; p.cout_store = C; p.rh = rh;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) if (int rc = codd_conv_roll(&p, 0)) { printf("rc %d\n", rc); return 1; }
  hipEventRecord(e0, 0);
  const int N = 50;
  for (int i = 0; i < N; ++i) codd_conv_roll(&p, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("C %d mode %d B %d %dx%d rh %d: %.1f us\n", C, mode, B, H, W, rh, ms / N * 1e3);
#ifdef ROLL_ABL_CLK
  long long hc[16];
  hipMemcpyFromSymbol(hc, HIP_SYMBOL(roll_clk), 128);
  const long long base = hc[1] < hc[9] ? (hc[1] < hc[5] ? hc[1] : hc[5]) : (hc[9] < hc[5] ? hc[9] : hc[5]);
  for (int o = 0; o < 12; o += 4)
    printf("  workgroup %s: entry +%.2f us, prologue %.2f us, main loop %.2f us (%lld shader ticks), ends +%.2f us\n",
           o == 0 ? "first" : (o == 4 ? "last " : "mid  "), (hc[o + 1] - base) / 100.0, (hc[o + 2] - hc[o + 1]) / 100.0,
           (hc[o + 3] - hc[o + 2]) / 100.0, hc[o], (hc[o + 3] - base) / 100.0);
#endif
  return 0;
}
