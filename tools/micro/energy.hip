// GPU microbenchmark for the ENERGY model of a DDPM step ("power"): each case keeps ONE resource of the chip busy
// for a few seconds while tools/energy_model.sh samples rocm-smi (package power, sclk) once per second beside it; tools/energy_model.py
// turns (power - idle power) / rate into joules per unit: J per MFMA, per fabric byte written / read (HBM side, Infinity-Cache side),
// per byte staged L2 -> LDS, per LDS fragment byte read, per VALU instruction.  This program starts no child process.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/energy tools/micro/energy.hip ; run: tools/micro/energy [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline float rnd(unsigned seed) { return ((int)(hash32(seed) >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }

// ---- MFMA only: MODE 0 f16 16x16x32, 1 bf16 16x16x32, 2 f32 16x16x4; data 1 = random, 2 = hi / lo pairs as the split modes multiply them
template <int MODE>
__global__ __launch_bounds__(512) void k_mfma(float* out, int iters, int data) {
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  if (MODE == 2) {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(id * 8 + i); b[i] = rnd(id * 8 + 4 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
  } else {
    h8 ah[4], bh[4];
    b8 ab[4], bb[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) {
        float va = rnd(id * 64 + i * 8 + j), vb = rnd(id * 64 + 32 + i * 8 + j);
        if (data == 2 && (i & 1)) { va *= (MODE == 0 ? 4.8828125e-4f : 3.90625e-3f); vb *= (MODE == 0 ? 4.8828125e-4f : 3.90625e-3f); }
        ah[i][j] = (_Float16)va; bh[i][j] = (_Float16)vb; ab[i][j] = (__bf16)va; bb[i][j] = (__bf16)vb;
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 3], bh[(i >> 1) & 3], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i & 3], bb[(i >> 1) & 3], acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5678f) out[id] = s;
}
// ---- MFMAs fed from LDS the way a K loop feeds them: per "K tile" a wave reads 10 row-tile fragments (20 x ds_read_b128) of random split-fp16
// data and issues 6 x 4 products x 3 MFMAs on them - the operands of consecutive MFMAs change, which a register-only loop never exercises
__global__ __launch_bounds__(512) void k_mfma_lds(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 128 * 1024 / 4; i += blockDim.x) {
    const float v = rnd(i * 7 + blockIdx.x);
    const _Float16 h = (_Float16)((i & 16) ? v * 4.8828125e-4f : v);  // (hi | lo planes alternate per 64 bytes as in the operand layout)
    const _Float16 h2 = (_Float16)((i & 16) ? rnd(i * 13 + 5) * 4.8828125e-4f : rnd(i * 13 + 5));
    ((unsigned*)lds)[i] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 acc[6][4];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const char* base = lds + ((it & 3) * 24 * 1024) + (wave & 1) * 12 * 1024 + lane * 16;  // 4 stages, conflict-free 16 B per lane
    h8 a[6][2], b[4][2];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i][0] = *(const h8*)(base + i * 2048); a[i][1] = *(const h8*)(base + i * 2048 + 1024); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j][0] = *(const h8*)(base + (6 + (j & 1)) * 2048 - (j >> 1) * 1024); b[j][1] = *(const h8*)(base + (j & 3) * 2048 + 1024); }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][1], a[i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][0], acc[i][j], 0, 0, 0);
      }
  }
  float sum = 0.f;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  if (sum == 1234.5678f) out[threadIdx.x] = sum;
}
// ---- scalar ALU only
__global__ __launch_bounds__(512) void k_salu(float* out, int iters) {
  unsigned a = blockIdx.x, b = 0x9e3779b9u;
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0"
                 : "+s"(a), "+s"(b));
  }
  if (a == 0x12345678u && iters < 0) out[0] = (float)b;
}
// ---- VALU only: 8 independent fma chains per lane
__global__ __launch_bounds__(512) void k_valu(float* out, int iters) {
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = rnd(id * 8 + i);
  const float m = 1.0000001f, c = rnd(id) * 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 1234.5678f) out[id] = s;
}
// ---- nothing: every CU holds 8 resident waves that (a) sleep (s_sleep: clock-gated issue), (b) spin on s_nop, (c) wait at s_barrier-free
// s_waitcnt-like stalls are approximated by (a).  The power of a chip whose CUs are occupied but not computing: what every stall of a
// real kernel costs while the clocks run.
template <int MODE>
__global__ __launch_bounds__(512) void k_spin(float* out, int iters) {
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) __builtin_amdgcn_s_sleep(127);
    else { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
  }
  if (iters < 0) out[threadIdx.x] = 1.f;
}
// ---- LDS fragment reads only: every lane reads 16 bytes per instruction, conflict-free (consecutive lanes, consecutive chunks)
__global__ __launch_bounds__(512) void k_ldsread(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) ((unsigned*)lds)[i] = hash32(i + blockIdx.x);
  __syncthreads();
  int4 acc = make_int4(0, 0, 0, 0);
  const char* p = lds + threadIdx.x * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int4 v = *(const int4*)(p + ((it * 8 + u) & 7) * 8192);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1.f;
}
// ---- L2 -> LDS by LDS-DMA: every workgroup streams the same `bytes` (L2 hits), 4 pieces per wave and stage, 3 stages in flight
__global__ __launch_bounds__(512) void k_dma(const char* src, long wg_stride, long bytes, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int PPW = 4, DEPTH = 3, NS = DEPTH + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  const long stage_bytes = (long)nw * PPW * 1024, nst = bytes / stage_bytes;
  long issued = 0;
  for (int r = 0; r < reps; ++r)
    for (long s = 0; s < nst; ++s, ++issued) {
      char* dst = lds + (issued % NS) * stage_bytes + (long)wave * PPW * 1024;
      const char* p = base + s * stage_bytes + (long)wave * PPW * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < PPW; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(p + i * 1024), (lds_void*)(dst + i * 1024), 16, 0, 0);
      if (issued >= DEPTH) __builtin_amdgcn_s_waitcnt(((DEPTH * PPW) & 15) | (7 << 4) | (15 << 8) | (((DEPTH * PPW) >> 4) << 14));
    }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
  __syncthreads();
  if (lds[threadIdx.x] == 123 && reps < 0) sink[0] = 1.f;
}
// ---- fabric: stores / loads of distinct bytes per workgroup
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_write(float4* out, long n_per_wg) {
  v4f* o = (v4f*)out + (long)blockIdx.x * n_per_wg;
  const v4f v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (long i = threadIdx.x; i < n_per_wg; i += blockDim.x) o[i] = v;
}
__global__ void k_read(const float4* in, long n_per_wg, float* sink) {
  const float4* p = in + (long)blockIdx.x * n_per_wg;
  float s = 0.f;
  for (long i = threadIdx.x; i < n_per_wg; i += blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) sink[0] = s;
}

static double now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

// run `launch` back to back for `secs` seconds; prints "CASE name t0 t1 units_per_second unit"
template <class F>
static void run_case(const char* name, double secs, double units_per_launch, const char* unit, F launch) {
  launch(); hipDeviceSynchronize();
  const double t0 = now();
  long n = 0;
  while (now() - t0 < secs) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    n += 20;
  }
  const double t1 = now();
  printf("CASE %s %.3f %.3f %.6e %s\n", name, t0, t1, units_per_launch * n / (t1 - t0), unit);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 5.0;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* sink; hipMalloc(&sink, (size_t)2 * cus * 512 * 4);
  char* big; const long BIG = 3L << 30; hipMalloc(&big, BIG); hipMemset(big, 1, BIG);
  hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_ldsread, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  { const double t0 = now(); sleep((unsigned)secs); printf("CASE idle %.3f %.3f 0 none\n", t0, now()); fflush(stdout); }
  run_case("spin_sleep", secs, 1.0, "launch", [&] { hipLaunchKernelGGL(k_spin<0>, dim3(2 * cus), dim3(512), 0, 0, sink, 2000); });
  run_case("spin_nop", secs, 1.0, "launch", [&] { hipLaunchKernelGGL(k_spin<1>, dim3(2 * cus), dim3(512), 0, 0, sink, 20000); });
  const int it_m = 4000;
  const double mfmas = (double)2 * cus * 8 * it_m * 8;  // wave-level MFMA instructions per launch
  run_case("mfma_f16_random", secs, mfmas, "MFMA", [&] { hipLaunchKernelGGL(k_mfma<0>, dim3(2 * cus), dim3(512), 0, 0, sink, it_m, 1); });
  run_case("mfma_f16_hilo", secs, mfmas, "MFMA", [&] { hipLaunchKernelGGL(k_mfma<0>, dim3(2 * cus), dim3(512), 0, 0, sink, it_m, 2); });
  run_case("mfma_bf16_random", secs, mfmas, "MFMA", [&] { hipLaunchKernelGGL(k_mfma<1>, dim3(2 * cus), dim3(512), 0, 0, sink, it_m, 1); });
  run_case("mfma_bf16_hilo", secs, mfmas, "MFMA", [&] { hipLaunchKernelGGL(k_mfma<1>, dim3(2 * cus), dim3(512), 0, 0, sink, it_m, 2); });
  run_case("mfma_f32_random", secs, mfmas, "MFMA", [&] { hipLaunchKernelGGL(k_mfma<2>, dim3(2 * cus), dim3(512), 0, 0, sink, it_m, 1); });
  hipFuncSetAttribute((const void*)k_mfma_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run_case("mfma_f16_from_lds", secs, (double)cus * 8 * 1500.0 * 72, "MFMA", [&] { hipLaunchKernelGGL(k_mfma_lds, dim3(cus), dim3(512), 128 * 1024, 0, sink, 1500); });
  run_case("salu", secs, (double)2 * cus * 8 * 40000.0 * 8, "SALUinst", [&] { hipLaunchKernelGGL(k_salu, dim3(2 * cus), dim3(512), 0, 0, sink, 40000); });
  run_case("valu_fma", secs, (double)2 * cus * 8 * 20000.0 * 8, "VALUinst", [&] { hipLaunchKernelGGL(k_valu, dim3(2 * cus), dim3(512), 0, 0, sink, 20000); });
  run_case("lds_read_b128", secs, (double)cus * 512 * 16 * 8 * 20000.0, "B", [&] { hipLaunchKernelGGL(k_ldsread, dim3(cus), dim3(512), 64 * 1024, 0, sink, 20000); });
  run_case("dma_l2_to_lds", secs, (double)cus * (2 << 20) * 64.0, "B", [&] { hipLaunchKernelGGL(k_dma, dim3(cus), dim3(512), 128 * 1024, 0, big, 0L, 2L << 20, 64, sink); });
  // distinct 8 MB per CU out of 2 GB: the Infinity Cache cannot hold it -> HBM; the same 256 KB per CU (64 MB in all) again and again -> Infinity Cache
  run_case("dma_hbm_to_lds", secs, (double)cus * (8 << 20), "B", [&] { hipLaunchKernelGGL(k_dma, dim3(cus), dim3(512), 128 * 1024, 0, big, 8L << 20, 8L << 20, 1, sink); });
  {
    const long per = 109L << 20, n_per_wg = per / 16 / cus;
    long slot = 0;
    const long nslots = BIG / per;
    run_case("write_hbm", secs, (double)per, "B", [&] { hipLaunchKernelGGL(k_write, dim3(cus), dim3(512), 0, 0, (float4*)(big + (slot++ % nslots) * per), n_per_wg); });
    run_case("write_same_109MB", secs, (double)per, "B", [&] { hipLaunchKernelGGL(k_write, dim3(cus), dim3(512), 0, 0, (float4*)big, n_per_wg); });
    slot = 0;
    run_case("read_hbm", secs, (double)per, "B", [&] { hipLaunchKernelGGL(k_read, dim3(cus), dim3(512), 0, 0, (const float4*)(big + (slot++ % nslots) * per), n_per_wg, sink); });
    run_case("read_same_109MB", secs, (double)per, "B", [&] { hipLaunchKernelGGL(k_read, dim3(cus), dim3(512), 0, 0, (const float4*)big, n_per_wg, sink); });
    run_case("write_then_read_109MB", secs, 2.0 * per, "B", [&] {
      hipLaunchKernelGGL(k_write, dim3(cus), dim3(512), 0, 0, (float4*)big, n_per_wg);
      hipLaunchKernelGGL(k_read, dim3(cus), dim3(512), 0, 0, (const float4*)big, n_per_wg, sink); });
  }
  { const double t0 = now(); sleep((unsigned)secs); printf("CASE idle2 %.3f %.3f 0 none\n", t0, now()); fflush(stdout); }
  return 0;
}
