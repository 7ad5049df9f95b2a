// dev microbenchmark: what does a barrier between the workgroups of ONE XCD cost on MI355X, against a
// device-wide one?  That number decides whether an "XCD-local persistent update step" (one launch for several dependent
// convolutions, workgroups of a 32-CU slab synchronising through an L2-resident counter) can beat the ~5-9 us of a kernel
// boundary (DESIGN.md finding 18).
//
// 256 workgroups x 256 threads, one per CU (100 KB of dynamic LDS keeps a second one off the CU).  Every workgroup reads its
// XCC id (s_getreg HW_REG_XCC_ID) and joins the group of that XCC.  Per round: every workgroup writes a 1 KB "halo" record,
// arrives at the group's counter, spins until the whole group has arrived and reads the record of its neighbour in the group
// (checked: the neighbour's value of THIS round must be visible).  Variants:
//   xcd/agent    per-XCC counter, agent-scope atomics + agent-scope loads of the halo (L2 of the XCC is the coherence point)
//   xcd/system   per-XCC counter, but __threadfence_system-style release / acquire (what a cross-XCD exchange needs)
//   device       ONE counter for all 256 workgroups, agent scope (crosses XCDs: sc1 write-through + invalidate)
//   xcd/light    per-XCC counter, relaxed agent-scope atomics, NO cache-maintenance fence: the halo stores are only waited for
//                (workgroup-scope release = s_waitcnt) and read back with L1-bypassing loads -- correct only if the XCC's L2 is
//                the coherence point for its own CUs (the stale-read count says whether it is)
// Every spin loop gives up after ~2 s (prints "TIMEOUT"): a workgroup that never became resident must not hang the box.
// Prints ns per round (kernel time / rounds) and the number of stale halo reads (must be 0).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/ubench/xcd_barrier.hip && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Ctl {
  unsigned arrive[8 * 32];   // per-XCC counters, one per 128 bytes
  unsigned all[32];          // device-wide counter
  unsigned members[8];       // workgroups that registered per XCC (set-up round)
  unsigned stale;
};

template <int MODE>  // 0: xcd/agent, 1: xcd/system fences, 2: device-wide counter, 3: xcd/light
__global__ __launch_bounds__(256) void barrier_kernel(Ctl* c, unsigned* halo, int rounds, unsigned long long* ticks) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 0.f;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  __shared__ unsigned slot, nmemb;
  if (threadIdx.x == 0) {
    slot = atomicAdd(&c->members[xcc], 1u);  // my index inside the XCC's group
    // device-wide set-up barrier so that every group knows its size
    __hip_atomic_fetch_add(&c->all[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    for (long long spin = 0; __hip_atomic_load(&c->all[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spin) {
      __builtin_amdgcn_s_sleep(1);
      if (spin > 20000000LL) { atomicOr(&c->stale, 0x80000000u); break; }
    }
    nmemb = __hip_atomic_load(&c->members[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned me = slot, n = MODE == 2 ? gridDim.x : nmemb;
  const unsigned gid = MODE == 2 ? blockIdx.x : xcc * 64 + me;              // record index of this workgroup
  const unsigned nb = MODE == 2 ? (blockIdx.x + 1) % gridDim.x : xcc * 64 + (me + 1) % nmemb;  // its neighbour's
  unsigned* cnt = MODE == 2 ? &c->all[16] : &c->arrive[xcc * 32];
  unsigned stale = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    // 1 KB halo record of this round
    // (double-buffered by round parity: a workgroup that is already in round r + 1 writes the OTHER buffer, so a mismatch
    // below can only mean that the neighbour's round-r record was not visible yet)
    unsigned* hb = halo + (size_t)(r & 1) * 8 * 64 * 256;
    hb[gid * 256 + threadIdx.x] = (unsigned)r * 1000u + gid;
    if (MODE == 1) __threadfence_system();
    else if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // wait for the stores, no cache maintenance
    else __threadfence();  // release: the record before the arrival
    __syncthreads();
    if (threadIdx.x == 0) {
      bool dead = false;
      if (MODE == 3) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (long long spin = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n * (unsigned)r; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if (spin > 20000000LL) { dead = true; break; }
        }
      } else {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, MODE == 1 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
        for (long long spin = 0; __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, MODE == 1 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT) < n * (unsigned)r; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if (spin > 20000000LL) { dead = true; break; }
        }
      }
      if (dead) { atomicOr(&c->stale, 0x80000000u); rounds = 0; }
    }
    __syncthreads();
    if (__hip_atomic_load(&c->stale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) break;
    const unsigned v = __hip_atomic_load(&hb[nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stale += v != (unsigned)r * 1000u + nb;
  }
  const unsigned long long t1 = wall_clock64();
  if (stale) atomicAdd(&c->stale, stale);
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* tag, int rounds) {
  Ctl* c; unsigned* halo; unsigned long long* ticks;
  CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&halo, 2 * 8 * 64 * 256 * 4)); CK(hipMalloc(&ticks, 256 * 8));
  CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(halo, 0, 2 * 8 * 64 * 256 * 4));
  CK(hipFuncSetAttribute((const void*)barrier_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  barrier_kernel<MODE><<<256, 256, 100 * 1024>>>(c, halo, rounds, ticks);  // MI355X: 256 CUs, one workgroup each
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> tk(256); CK(hipMemcpy(tk.data(), ticks, 256 * 8, hipMemcpyDeviceToHost));
  unsigned long long mx = 0; for (auto t : tk) mx = t > mx ? t : mx;
  printf("%-12s %d rounds: %.1f ns per round by HIP events, %.1f ns by the device's 100 MHz wall clock (slowest workgroup); workgroups per XCC:", tag,
         rounds, ms * 1e6 / rounds, (double)mx * 10.0 / rounds);
  for (int i = 0; i < 8; ++i) printf(" %u", h.members[i]);
  printf("; stale halo reads %u%s\n", h.stale & 0x7fffffffu, (h.stale & 0x80000000u) ? "  TIMEOUT (a spin loop gave up)" : "");
  CK(hipFree(c)); CK(hipFree(halo)); CK(hipFree(ticks));
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("xcd/agent", rounds);
    run<1>("xcd/system", rounds);
    run<2>("device", rounds);
    run<3>("xcd/light", rounds);
  }
  return 0;
}
