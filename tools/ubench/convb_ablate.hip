// Ablation harness for the split-bf16 conv kernel: one instantiation, timed with hipEvents.  Built three times
// (full / -DCONVB_NO_PRODUCER / -DCONVB_NO_CONSUMER) by tools/ubench/run_convb_ablate.sh.
#include "conv_bf16_kernel.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#ifndef KSPLIT
#define KSPLIT 1
#endif
#ifndef CFG
#define CFG 2, 2, 5, 2, 3, 0, 1
#endif
template __global__ void conv_bf16_kernel<CFG>(const ConvB);

int main(int argc, char** argv) {
  int cin = 256, cout = 256, H = 72, W = 120, ks = 3, th = 9, xb = 1, ck = 16, mb = 4, pgw = 2, cgw = 2;
  if (argc > 1) cin = atoi(argv[1]);
  if (argc > 2) cout = atoi(argv[2]);
  if (argc > 3) th = atoi(argv[3]);
  if (argc > 4) xb = atoi(argv[4]);
  if (argc > 5) ck = atoi(argv[5]);
  if (argc > 6) mb = atoi(argv[6]);
  if (argc > 7) pgw = atoi(argv[7]);
  if (argc > 8) cgw = atoi(argv[8]);
  if (argc > 9) H = atoi(argv[9]);
  if (argc > 10) W = atoi(argv[10]);
  codd_conv_params p;
  memset(&p, 0, sizeof(p));
  float *x, *out;
  void* wp;
  hipMalloc(&x, (size_t)cin * H * W * 4);
  hipMalloc(&out, (size_t)cout * H * W * 4);
  hipMemset(x, 0, (size_t)cin * H * W * 4);
  const size_t wbytes = (size_t)((cout + 16 * mb - 1) / (16 * mb)) * ((cin + ck - 1) / ck) * 2 * ((ks * ks * (ck / 8) + 3) / 4) * 4 * 16 * mb * 16;
  hipMalloc(&wp, wbytes);
  hipMemset(wp, 0, wbytes);
  p.in0.ptr = x; p.in0.ctot = cin; p.C0 = cin; p.B = 1; p.Hin = H; p.Win = W;
  p.wpacked = (const float*)wp; p.out = out; p.out_ctot = cout; p.Cout = cout; p.Hout = H; p.Wout = W;
  p.kh = p.kw = ks; p.sy = p.sx = 1; p.pad_t = p.pad_l = ks / 2; p.dil_y = p.dil_x = 1;
  p.mb = mb; p.npb = xb; p.nw = th; p.ck = ck; p.layout = 2; p.terms = 3; p.pgw = pgw; p.cgw = cgw;
  // split input with borders (pad, pad) and room for the tile overhang
  const int c8 = ((cin + ck - 1) / ck) * (ck / 8), hp = H + 2 * (ks / 2) + 16, wpx = W + 2 * (ks / 2) + 32;
  void* xs;
  hipMalloc(&xs, (size_t)2 * c8 * hp * wpx * 16);
  hipMemset(xs, 0, (size_t)2 * c8 * hp * wpx * 16);
  p.xs = xs; p.xs_c8 = c8; p.xs_hp = hp; p.xs_wp = wpx; p.xs_bt = ks / 2; p.xs_bl = ks / 2;
  ConvB k;
  size_t lds;
  long long grid;
  int rc = convb_geometry(&p, k, lds, grid, true);
  if (rc) { printf("geometry rc %d\n", rc); return 1; }
  auto kern = conv_bf16_kernel<CFG>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nt = (pgw * cgw * KSPLIT + CONVB_NWP) * 64; p.ksplit = KSPLIT;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<<<(int)grid, nt, lds>>>(k);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) kern<<<(int)grid, nt, lds>>>(k);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 < best) best = ms / 10;
  }
  const double gf = 2.0 * cin * cout * ks * ks * H * W / 1e9;
  printf("%-22s cin %d cout %d tile %dx%d ck %d mb %d grid %lld lds %zu nk %d chunks %d qr %d: %.1f us  %.1f TF  err=%s\n",
#if defined(CONVB_NO_PRODUCER)
         "no-producer",
#elif defined(CONVB_NO_CONSUMER)
         "no-consumer",
#else
         "full",
#endif
         cin, cout, th, 16 * xb, ck, mb, grid, lds, k.nk, k.nchunks, k.ibuf16 >> 6, best * 1e3, gf / best,
         hipGetErrorString(hipGetLastError()));
  return 0;
}
