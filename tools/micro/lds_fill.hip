// GPU microbenchmark: how fast can ONE CU pull L2-resident bytes into its LDS?  Every GEMM of the bf16 mode is bound by this path
// (74 GB/s per CU in the 128 x 128 K loops), so its ceiling decides what a restaging of the bf16 GEMMs can win.
//   grid = one workgroup per CU (256) x NW waves; every workgroup streams the SAME `shared_kb` KiB (a weight panel: L2 hits after the
//   first touch) or its OWN slice of a big buffer, `reps` times, in stages of `stage_kb` with `depth` stages in flight:
//   mode 0  LDS-DMA (global_load_lds_dwordx4), counted vmcnt, nothing reads the LDS
//   mode 1  register staging (global_load_dwordx4 -> ds_write_b128), `depth` x 16 B in flight per lane
//   mode 2  global_load_dwordx4 into registers only (no LDS): the L1/L2 -> CU path itself
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/lds_fill tools/micro/lds_fill.hip ; run: tools/micro/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int DEPTH>
__device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_s_waitcnt((DEPTH & 15) | (7 << 4) | (15 << 8) | ((DEPTH >> 4) << 14));
}

// mode 0: every wave issues `ppw` pieces (1 KiB each) per stage; DEPTH stages in flight per wave
template <int PPW, int DEPTH>
__global__ __launch_bounds__(1024) void k_dma(const char* src, long wg_stride, long bytes, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  const long stage_bytes = (long)nw * PPW * 1024;
  const long nst = bytes / stage_bytes;
  constexpr int NS = DEPTH + 1;
  long issued = 0;
  for (int r = 0; r < reps; ++r) {
    for (long s = 0; s < nst; ++s, ++issued) {
      char* dst = lds + (issued % NS) * stage_bytes + (long)wave * PPW * 1024;
      const char* p = base + s * stage_bytes + (long)wave * PPW * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void*)(p + i * 1024), (lds_void*)(dst + i * 1024), 16, 0, 0);
      if (issued >= DEPTH) wait_vm<DEPTH * PPW>();  // everything but the newest DEPTH stages has landed
    }
  }
  wait_vm<0>();
  __syncthreads();
  if (lds[threadIdx.x] == 123 && reps < 0) sink[0] = 1.f;
}

// mode 1 / 2: register staging; U = 16-byte loads in flight per lane
template <int U, bool TO_LDS>
__global__ __launch_bounds__(1024) void k_reg(const char* src, long wg_stride, long bytes, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const char* base = src + (long)blockIdx.x * wg_stride;
  const long per_iter = (long)blockDim.x * 16 * U;
  const long nit = bytes / per_iter;
  int4 acc = make_int4(0, 0, 0, 0);
  for (int r = 0; r < reps; ++r) {
    for (long it = 0; it < nit; ++it) {
      int4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *(const int4*)(base + it * per_iter + ((long)u * blockDim.x + threadIdx.x) * 16);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (TO_LDS) *(int4*)(lds + (((it & 1) * U + u) * (long)blockDim.x + threadIdx.x) * 16) = v[u];
        else { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
      }
    }
  }
  __syncthreads();
  if (TO_LDS) { if (lds[threadIdx.x] == 123 && reps < 0) sink[0] = 1.f; }
  else if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678 && reps < 0) sink[0] = 1.f;
}

template <class F>
static float time_ms(F f, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const long BIG = 1L << 30;
  char* buf;
  float* sink;
  hipMalloc(&buf, BIG);
  hipMemset(buf, 1, BIG);
  hipMalloc(&sink, 64);
  const int reps = 8;
  printf("CUs %d; GB/s per CU = bytes pulled by one workgroup / kernel time (one workgroup per CU)\n", cus);
#define RUN_DMA(NW, PPW, DEPTH, SHARED_KB)                                                                                    \
  {                                                                                                                           \
    const long bytes = (long)(SHARED_KB) * 1024, stride = 0;                                                                  \
    hipFuncSetAttribute((const void*)k_dma<PPW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    const int lds = (DEPTH + 1) * NW * PPW * 1024;                                                                            \
    if (lds <= 160 * 1024) {                                                                                                  \
      float ms = time_ms([&] { hipLaunchKernelGGL((k_dma<PPW, DEPTH>), dim3(cus), dim3(NW * 64), lds, 0, buf, stride, bytes, reps, sink); }, 20); \
      printf("dma   shared %5d KB  waves %2d  pieces/wave/stage %2d  stage %3d KB  depth %d (LDS %3d KB): %6.1f GB/s per CU  %5.2f TB/s chip\n", \
             SHARED_KB, NW, PPW, NW * PPW, DEPTH, lds / 1024, bytes * reps / (ms * 1e-3) / 1e9, bytes * reps * (double)cus / (ms * 1e-3) / 1e12); \
    }                                                                                                                         \
  }
  for (int kb : {2048, 512}) {
    RUN_DMA(8, 4, 1, kb) RUN_DMA(8, 4, 2, kb) RUN_DMA(8, 4, 3, kb) RUN_DMA(8, 8, 1, kb) RUN_DMA(4, 8, 1, kb) RUN_DMA(4, 8, 2, kb) RUN_DMA(4, 4, 3, kb)
    RUN_DMA(16, 2, 2, kb) RUN_DMA(16, 4, 1, kb) RUN_DMA(2, 8, 3, kb) RUN_DMA(1, 16, 3, kb) RUN_DMA(8, 2, 4, kb) RUN_DMA(8, 1, 8, kb)
  }
  // distinct data per workgroup, 4 MB each (Infinity Cache / HBM side)
  {
    const long bytes = 4L << 20, stride = 4L << 20;
    hipFuncSetAttribute((const void*)k_dma<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = time_ms([&] { hipLaunchKernelGGL((k_dma<4, 3>), dim3(cus), dim3(512), 128 * 1024, 0, buf, stride, bytes, 1, sink); }, 10);
    printf("dma   distinct 4 MB per CU (1 GB total), 8 waves, depth 3: %6.1f GB/s per CU  %5.2f TB/s chip\n", bytes / (ms * 1e-3) / 1e9, bytes * (double)cus / (ms * 1e-3) / 1e12);
  }
#define RUN_REG(NW, U, TOLDS, SHARED_KB)                                                                                      \
  {                                                                                                                           \
    const long bytes = (long)(SHARED_KB) * 1024;                                                                              \
    hipFuncSetAttribute((const void*)k_reg<U, TOLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
    const int lds = TOLDS ? 2 * U * NW * 1024 : 1024;                                                                         \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_reg<U, TOLDS>), dim3(cus), dim3(NW * 64), lds, 0, buf, 0L, bytes, reps, sink); }, 20); \
    printf("%s shared %5d KB  waves %2d  loads in flight per lane %2d: %6.1f GB/s per CU  %5.2f TB/s chip\n", TOLDS ? "reg->lds" : "reg only", \
           SHARED_KB, NW, U, bytes * reps / (ms * 1e-3) / 1e9, bytes * reps * (double)cus / (ms * 1e-3) / 1e12);            \
  }
  for (int kb : {2048, 512}) {
    RUN_REG(8, 4, true, kb) RUN_REG(8, 8, true, kb) RUN_REG(16, 4, true, kb) RUN_REG(16, 8, true, kb)
    RUN_REG(8, 4, false, kb) RUN_REG(8, 8, false, kb) RUN_REG(16, 4, false, kb) RUN_REG(16, 8, false, kb) RUN_REG(4, 8, false, kb)
  }
  return 0;
}
