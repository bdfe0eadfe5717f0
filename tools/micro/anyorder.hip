// GPU microbenchmark: can two dependent kernels of ONE stream overlap on gfx950 (hipExtLaunchKernel with hipExtAnyOrderLaunch = the AQL
// barrier bit cleared), so that workgroups of launch k + 1 start on CUs that launch k has left and wait for their inputs on device-side
// ready flags instead of on the kernel boundary?  (VERDICT r5 "Next round" item 2: remove the chip-wide lock-step without one mega-kernel.)
//   (1) producer P: 256 workgroups spin until a word `go` is set - or a time budget runs out - and stamp start / end (100 MHz wall clock);
//       consumer C (launched behind P on the same stream) stamps its start and sets `go`.  In order: P runs its whole budget and C starts
//       after it.  Any order: C starts while P spins, P ends early.  The same with C on a second stream for reference.
//   (2) placement: XCC_ID of every workgroup of two back-to-back 256-workgroup launches (is b % 8 -> XCD stable from launch to launch?
//       32 workgroups per XCD exactly?), also for the any-order consumer.
//   (3) chained tiles: K launches of 256 workgroups, workgroup b of launch k waits for flag[k - 1][b] (same b = same XCD under round-robin
//       placement), works for T us, sets flag[k][b]; T jittered per workgroup.  In-order launches against any-order launches: the
//       any-order chain should take  K x mean(T)  instead of  K x (max(T) + boundary).
// build: hipcc -O3 --offload-arch=gfx950 -o anyorder tools/micro/anyorder.hip ; run: ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Stamp { unsigned long long t0, t1; unsigned xcc, pad; };

__global__ void k_producer(unsigned* go, Stamp* st, long budget_ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    while (ld_sc1(go) == 0 && (long)(wall_clock64() - t0) < budget_ticks) __builtin_amdgcn_s_sleep(8);
    st[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), xcc_id(), 0};
  }
}
__global__ void k_consumer(unsigned* go, Stamp* st) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) st_sc1(go, 1u);
    st[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), xcc_id(), 0};
  }
}
__global__ void k_reset(unsigned* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0; }

// (3) chained tiles.  flags[k][b]; work = spin for ticks[b ^ k-dependent]; LDS use keeps one workgroup per CU like the clip GEMMs
__global__ void k_chain(const unsigned* prev, unsigned* mine, const int* ticks, int k, unsigned* timeouts, Stamp* st) {
  extern __shared__ char lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    if (prev) {
      while (ld_sc1(prev + blockIdx.x) == 0) {
        if ((long)(wall_clock64() - t0) > 2000000) { atomicAdd(timeouts, 1u); break; }  // 20 ms: never hang the box
        __builtin_amdgcn_s_sleep(4);
      }
    }
    const unsigned long long tw = wall_clock64();
    const long T = ticks[(blockIdx.x * 7 + k * 13) & 255];
    while ((long)(wall_clock64() - tw) < T) __builtin_amdgcn_s_sleep(2);
    lds[0] = (char)k;
    st_sc1(mine + blockIdx.x, 1u);
    if (st) st[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), xcc_id(), 0};
  }
}

static void launch(const void* f, dim3 g, dim3 b, void** args, size_t lds, hipStream_t s, bool anyorder) {
  if (anyorder) CK(hipExtLaunchKernel(f, g, b, args, lds, s, nullptr, nullptr, hipExtAnyOrderLaunch));
  else CK(hipLaunchKernel(f, g, b, args, lds, s));
}

int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned* go;
  Stamp *sp, *sc;
  CK(hipMalloc(&go, 4096));
  CK(hipMalloc(&sp, 256 * sizeof(Stamp)));
  CK(hipMalloc(&sc, 256 * sizeof(Stamp)));
  std::vector<Stamp> hp(256), hc(256);
  const long budget = 20000;  // 200 us
  for (int mode = 0; mode < 3; ++mode) {  // 0 in order, 1 any order (same stream), 2 second stream
    for (int rep = 0; rep < 3; ++rep) {
      int n = 1;
      void* a0[] = {&go, &n};
      launch((const void*)k_reset, dim3(1), dim3(64), a0, 0, s0, false);
      CK(hipStreamSynchronize(s0));
      long b = budget;
      void* ap[] = {&go, &sp, &b};
      void* ac[] = {&go, &sc};
      launch((const void*)k_producer, dim3(256), dim3(256), ap, 0, s0, false);
      launch((const void*)k_consumer, dim3(256), dim3(64), ac, 0, mode == 2 ? s1 : s0, mode == 1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hp.data(), sp, 256 * sizeof(Stamp), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hc.data(), sc, 256 * sizeof(Stamp), hipMemcpyDeviceToHost));
      unsigned long long p0 = ~0ull, p1 = 0, c0 = ~0ull;
      for (int i = 0; i < 256; ++i) { p0 = std::min(p0, hp[i].t0); p1 = std::max(p1, hp[i].t1); c0 = std::min(c0, hc[i].t0); }
      printf("(1) %-28s rep %d: producer ran %7.1f us (budget 200), consumer's first workgroup started %+8.1f us after the producer's start -> %s\n",
             mode == 0 ? "in order" : mode == 1 ? "hipExtAnyOrderLaunch" : "consumer on a second stream", rep, (p1 - p0) / 100.0, ((double)c0 - (double)p0) / 100.0,
             c0 < p1 ? "OVERLAP" : "serialised");
    }
  }
  // (2) placement
  {
    int per[2][8] = {};
    int same = 0;
    for (int i = 0; i < 256; ++i) { per[0][hp[i].xcc & 7]++; per[1][hc[i].xcc & 7]++; same += hp[i].xcc == hc[i].xcc; }
    printf("(2) workgroups per XCC, producer: "); for (int x = 0; x < 8; ++x) printf("%d ", per[0][x]);
    printf(" consumer: "); for (int x = 0; x < 8; ++x) printf("%d ", per[1][x]);
    printf("\n    xcc(b) of the first 16 producer workgroups: "); for (int i = 0; i < 16; ++i) printf("%u ", hp[i].xcc);
    printf("\n    xcc(b) of the first 16 consumer workgroups: "); for (int i = 0; i < 16; ++i) printf("%u ", hc[i].xcc);
    int rr = 0; for (int i = 8; i < 256; ++i) rr += hp[i].xcc == hp[i - 8].xcc;
    printf("\n    producer b and b - 8 on the same XCC: %d / 248; workgroup b of both launches on the same XCC: %d / 256\n", rr, same);
  }
  // (3) chained tiles
  {
    const int K = 40;
    unsigned* flags;
    int* ticks;
    unsigned* timeouts;
    CK(hipMalloc(&flags, (size_t)(K + 1) * 256 * 4));
    CK(hipMalloc(&ticks, 256 * 4));
    CK(hipMalloc(&timeouts, 4));
    std::vector<int> ht(256);
    srand(1);
    double meanT = 0, maxT = 0;
    for (int i = 0; i < 256; ++i) { ht[i] = 2000 + rand() % 800; meanT += ht[i] / 256.0; maxT = std::max<double>(maxT, ht[i]); }  // 20 - 28 us
    CK(hipMemcpy(ticks, ht.data(), 256 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lds_kb : {0, 150}) {
      if (lds_kb) CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
      for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemsetAsync(flags, 0, (size_t)(K + 1) * 256 * 4, s0));
          CK(hipMemsetAsync(timeouts, 0, 4, s0));
          CK(hipStreamSynchronize(s0));
          CK(hipEventRecord(e0, s0));
          for (int k = 0; k < K; ++k) {
            const unsigned* prev = k ? flags + (size_t)(k - 1) * 256 : nullptr;
            unsigned* mine = flags + (size_t)k * 256;
            Stamp* stn = k == K - 1 ? sp : nullptr;
            void* a[] = {&prev, &mine, &ticks, &k, &timeouts, &stn};
            launch((const void*)k_chain, dim3(256), dim3(512), a, (size_t)lds_kb * 1024, s0, mode == 1 && k > 0);
          }
          CK(hipEventRecord(e1, s0));
          CK(hipStreamSynchronize(s0));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          unsigned to;
          CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
          printf("(3) %d KB LDS, %-22s rep %d: %d chained launches %8.1f us = %6.2f us per launch (work per workgroup mean %.1f max %.1f us)  timeouts %u\n",
                 lds_kb, mode ? "hipExtAnyOrderLaunch" : "in order", rep, K, ms * 1e3, ms * 1e3 / K, meanT / 100.0, maxT / 100.0, to);
        }
      }
    }
  }
  return 0;
}
