// GPU microbenchmark: what the matrix pipe sustains under the board's power cap.  Every wave of a full-chip launch (256 CUs x 8 waves)
// issues only v_mfma_f32_16x16x32_f16 (or _bf16, or 16x16x4_f32) on register operands - no LDS, no memory - for a few seconds per case;
// tools/mfma_power.sh runs it in the background and samples rocm-smi (package power, sclk) once per second beside it (this program
// itself starts no child process: a GPU process must not fork + exec on this pool).
//   operands: (r) random normal-ish values with full mantissas   (z) zeros   (s) random hi + small random "lo" pairs as the split modes multiply them
// Prints TFLOP/s (2 * 16*16*32 per MFMA), the implied cycles per MFMA at the sampled clock, power.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_power tools/micro/mfma_power.hip ; run: tools/micro/mfma_power [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <ctime>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ inline float rnd(unsigned seed) { return ((int)(hash32(seed) >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }  // uniform in [-2, 2)

template <int MODE>  // 0 f16, 1 bf16, 2 f32 (16x16x4)
__global__ __launch_bounds__(512) void k_mfma(float* out, int iters, int data) {
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) {
        float va = data == 0 ? 0.f : rnd(id * 64 + i * 8 + j), vb = data == 0 ? 0.f : rnd(id * 64 + 32 + i * 8 + j);
        if (data == 2 && (i & 1)) { va *= 4.8828125e-4f; vb *= 4.8828125e-4f; }  // the "lo" halves: 2^-11 of the hi values
        a[i][j] = (_Float16)va; b[i][j] = (_Float16)vb;
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
  } else if (MODE == 1) {
    b8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) {
        float va = data == 0 ? 0.f : rnd(id * 64 + i * 8 + j), vb = data == 0 ? 0.f : rnd(id * 64 + 32 + i * 8 + j);
        if (data == 2 && (i & 1)) { va *= 3.90625e-3f; vb *= 3.90625e-3f; }
        a[i][j] = (__bf16)va; b[i][j] = (__bf16)vb;
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
  } else if (MODE == 4 || MODE == 5 || MODE == 6) {  // operand reuse between consecutive MFMAs: 4 = none (8 distinct A and B), 5 = same A and B, 6 = same A
    h8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) {
        a[i][j] = (_Float16)(data == 0 ? 0.f : rnd(id * 128 + i * 8 + j)); b[i][j] = (_Float16)(data == 0 ? 0.f : rnd(id * 128 + 64 + i * 8 + j));
      }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[MODE == 4 ? i : 0], b[MODE == 5 ? 0 : i], acc[i], 0, 0, 0);
    }
  } else if (MODE == 7) {  // v_mfma_i32_16x16x64_i8: twice the multiply-adds per instruction of the 16-bit shapes (an Ozaki-style int8 slicing of fp32 would need 6 of them per product)
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 a[4], b[4], c[8];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        a[i][j] = data == 0 ? 0 : (int)hash32(id * 64 + i * 4 + j); b[i][j] = data == 0 ? 0 : (int)hash32(id * 64 + 16 + i * 4 + j);
      }
    for (int i = 0; i < 8; ++i) c[i] = i4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 1) & 3], c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) acc[i][j] = (float)c[i][j];
  } else if (MODE == 3) {  // v_mfma_f32_32x32x16_f16: 4 independent 32 x 32 accumulators (64 registers), the same flops per instruction-cycle
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) {
        a[i][j] = (_Float16)(data == 0 ? 0.f : rnd(id * 64 + i * 8 + j)); b[i][j] = (_Float16)(data == 0 ? 0.f : rnd(id * 64 + 32 + i * 8 + j));
      }
    f16v c[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 3], c[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j & 3] += c[i][j];
  } else {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = data == 0 ? 0.f : rnd(id * 8 + i); b[i] = data == 0 ? 0.f : rnd(id * 8 + 4 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5678f) out[id] = s;  // keep the chain alive without traffic
}

int main(int argc, char** argv) {
  const int secs = argc > 1 ? atoi(argv[1]) : 5;
  const int ncase = 15;
  hipStream_t st;
  hipStreamCreate(&st);
  float* out;
  hipMalloc(&out, 256 * 8 * 64 * 4 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { int mode, data; const char* name; double flop_per_mfma; } cases[ncase] = {
      {0, 1, "f16 16x16x32 random", 16384.}, {0, 2, "f16 16x16x32 hi/lo pairs", 16384.}, {0, 0, "f16 16x16x32 zeros", 16384.},
      {1, 1, "bf16 16x16x32 random", 16384.}, {1, 2, "bf16 16x16x32 hi/lo pairs", 16384.}, {1, 0, "bf16 16x16x32 zeros", 16384.},
      {4, 1, "f16 16x16x32 no operand shared", 16384.}, {6, 1, "f16 16x16x32 A shared", 16384.}, {5, 1, "f16 16x16x32 A and B shared", 16384.},
      {7, 1, "i8 16x16x64 random", 32768.}, {7, 0, "i8 16x16x64 zeros", 32768.},
      {3, 1, "f16 32x32x16 random", 32768.}, {3, 0, "f16 32x32x16 zeros", 32768.},
      {2, 1, "f32 16x16x4 random", 2048.},   {2, 0, "f32 16x16x4 zeros", 2048.}};
  const int iters = 20000, grid = 256 * 2, block = 512;  // 2 workgroups x 8 waves per CU = 4 waves per SIMD
  for (int c = 0; c < ncase; ++c) {
    auto launch = [&]() {
      if (cases[c].mode == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 5) hipLaunchKernelGGL(k_mfma<5>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 6) hipLaunchKernelGGL(k_mfma<6>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 7) hipLaunchKernelGGL(k_mfma<7>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else if (cases[c].mode == 3) hipLaunchKernelGGL(k_mfma<3>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
      else hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(block), 0, st, out, iters, cases[c].data);
    };
    launch();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st); launch(); hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms1 = 0.f;
    hipEventElapsedTime(&ms1, e0, e1);
    const int n = (int)(secs * 1000.0 / ms1) + 1;
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)n * grid * (block / 64) * (double)iters * (cases[c].mode == 3 ? 4 : 8);
    const double tf = mfmas * cases[c].flop_per_mfma / (ms * 1e-3) / 1e12;
    const double per_simd_per_s = mfmas / 1024.0 / (ms * 1e-3);
    printf("[t=%ld] %-32s %8.1f TFLOP/s  (%.0f M MFMA/s per SIMD; first launch %.2f ms, %d launches in %.0f ms)\n", (long)time(nullptr), cases[c].name, tf, per_simd_per_s / 1e6, ms1, n, ms);
    fflush(stdout);
    sleep(1);
  }
  return 0;
}
