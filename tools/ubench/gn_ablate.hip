// dev: ablation builds of se3_gn_build_kernel (csrc/motion.hip compiled with -DGN_ABL_* switches), timed stand-alone at
// the update loop's shape (72x120, radius 32) on records made by codd_se3_gn_step from random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I codd_amd/csrc -fno-slp-vectorize [-DGN_ABL_x] \
//         tools/ubench/gn_ablate.hip -o tools/ubench/gn_ablate_x.bin ; ./gn_ablate_x.bin
#include "../../codd_amd/csrc/motion.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; }
int main() {
  const int B = 1, h = 72, w = 120, N = h * w, radius = 32;
  const float fx = 131.25f, fy = 131.25f, cx = 60.f, cy = 33.75f;
  unsigned s = 1;
  std::vector<float> T(N * 7), ae(32 * N), xyz(N * 3), delta(3 * N), weight(3 * N), d1(N);
  for (int i = 0; i < N; ++i) {
    for (int c = 0; c < 3; ++c) T[i * 7 + c] = (rnd(s) - 0.5f) * 0.02f;
    T[i * 7 + 3] = T[i * 7 + 4] = T[i * 7 + 5] = 0.f; T[i * 7 + 6] = 1.f;
    d1[i] = rnd(s) * 30.f + 3.f;
    for (int c = 0; c < 3; ++c) xyz[i * 3 + c] = rnd(s) * 50.f;
  }
  for (auto& v : ae) v = (rnd(s) - 0.5f) * 0.7f;  // small distances: every neighbour counts (as with the bench weights)
  for (auto& v : delta) v = (rnd(s) - 0.5f) * 0.2f;
  for (auto& v : weight) v = rnd(s);
  float *dT, *dae, *dxyz, *ddl, *dw, *dd1, *Hb;
  auto up = [](float** d, const std::vector<float>& v) { hipMalloc(d, v.size() * 4); hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice); };
  up(&dT, T); up(&dae, ae); up(&dxyz, xyz); up(&ddl, delta); up(&dw, weight); up(&dd1, d1);
  hipMalloc(&Hb, codd_se3_gn_scratch(B, h, w, radius) * 4);
  if (int rc = codd_se3_gn_step(dT, dae, 32, dxyz, ddl, dw, dd1, B, h, w, fx, fy, cx, cy, radius, 1e-4f, 0.1f, Hb, 0)) { printf("rc %d\n", rc); return 1; }
  hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice);
  const int tiles_x = cdiv(w, 8), ntiles = tiles_x * cdiv(h, 8);
  const int q4 = gn_q4(), gmax = gn_gmax(radius);
  const float* jd = Hb + (size_t)B * ntiles * gmax * 27 * 64;
  const bool pair = getenv("GN3") && atoi(getenv("GN3")) == 1;  // the pair builder (se3_gn_build3_kernel)
  auto launch = [&]() {
    if (pair && getenv("GN4") && atoi(getenv("GN4")) == 1) {  // the two-pass pair builder
      se3_gn_build4_kernel<<<dim3(ntiles, gmax, B), 64 * GN_WAVES, 0, 0>>>(dT, jd, Hb + gn_geo2_offset(B, h, w, radius), h, w, fx, fy, cx,
                                                                          cy, radius, tiles_x, ntiles, q4, gmax, Hb);
      return;
    }
    if (pair) {
      se3_gn_build3_kernel<<<dim3(ntiles, gmax, B), 64 * GN_WAVES, 0, 0>>>(dT, jd, Hb + gn_geo2_offset(B, h, w, radius), h, w, fx, fy, cx,
                                                                          cy, radius, tiles_x, ntiles, q4, gmax, Hb);
      return;
    }
    se3_gn_build_kernel<false><<<dim3(ntiles, gmax, B), 64 * GN_WAVES, 0, 0>>>(dT, jd, h, w, fx, fy, cx, cy, radius, tiles_x, ntiles, q4,
                                                                              gmax, Hb, nullptr, 1e-4f, 0.1f);
  };
  for (int i = 0; i < 5; ++i) launch();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  const int NI = 50;
  for (int i = 0; i < NI; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s builder (q4 %d, %d x %d workgroups): %.1f us%s\n", pair ? "pair" : "J-entry", q4, ntiles, gmax, ms / NI * 1e3, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
  return 0;
}
