// GPU microbenchmark: do MFMA work of one wave and VALU work of ANOTHER wave on the same SIMD overlap?
// build: hipcc -O3 --offload-arch=gfx950 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(ad))
#define M(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

// mode bit0: waves 0-3 run MFMA loop; bit1: waves 4-7 run VALU loop; bit2: waves 0-3 run MFMA+VALU interleaved
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = wall_clock64(); const long long c0_ = clock64();
  float r = 0.f;
  if ((wave < 4 && (mode & 1)) || (mode & 8)) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      M(c0);
      M(c1);
      M(c2);
      M(c3);
    }
    r = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (wave >= 4 && (mode & 2)) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float m = 1.0001f, ad = 0.5f;
    for (int i = 0; i < iters; ++i) {
      // 12 VALU per iteration (3 per MFMA of the other wave's iteration)
      F(x0); F(x1); F(x2); F(x3);
      F(x4); F(x5); F(x6); F(x7);
      F(x0); F(x1); F(x2); F(x3);
    }
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (wave < 4 && (mode & 4)) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float m = 1.0001f, ad = 0.5f;
    for (int i = 0; i < iters; ++i) {
      M(c0);
      F(x0); F(x1); F(x2);
      M(c1);
      F(x3); F(x4); F(x5);
      M(c2);
      F(x6); F(x7); F(x0);
      M(c3);
      F(x1); F(x2); F(x3);
    }
    r = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  const unsigned long long t1 = wall_clock64(); const long long c1_ = clock64();
  if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; cyc[2048 + blockIdx.x * 8 + wave] = (unsigned long long)(c1_ - c0_); }
  if (r == 12345.678f) out[0] = r;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 2 * 256 * 8 * 8);
  const int iters = 20000;
  const char* names[] = {"", "MFMA waves alone", "VALU waves alone", "MFMA waves + VALU waves (different waves, same SIMDs)", "MFMA+VALU interleaved in one wave",
                         "", "interleaved wave + VALU waves", "", "all 8 waves MFMA (2 per SIMD)"};
  for (int mode : {1, 8, 2, 3, 4}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, 2 * 256 * 8 * 8);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
      hipDeviceSynchronize();
    }
    unsigned long long h[8], hc[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hc, cyc + 2048, sizeof(hc), hipMemcpyDeviceToHost);
    printf("   s_memtime ticks: wave0 %llu (%.3f GHz-equivalent)  wave4 %llu (%.3f)\n", hc[0], h[0] ? hc[0] / (h[0] * 10.0) : 0.0, hc[4], h[4] ? hc[4] / (h[4] * 10.0) : 0.0);
    printf("mode %d %-60s: wave0 (MFMA) %.1f us  wave4 (VALU) %.1f us   [per iter: 4 MFMA = 64 MFMA-cycles, 12 VALU = 48 VALU-cycles; ideal MFMA %.1f us, VALU %.1f us at 2.4 GHz]\n",
           mode, names[mode], h[0] / 100.0, h[4] / 100.0, iters * 64 / 2400.0, iters * 48 / 2400.0);
  }
  return 0;
}
