// GPU microbenchmark: which instruction classes of ANOTHER wave overlap with a wave issuing MFMAs back-to-back on the
// same SIMD?  waves 0-3: MFMA loop; waves 4-7: filler loop of one instruction class.
// build: hipcc -O3 --offload-arch=gfx950 -o overlap2 overlap2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define M(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int KIND>
__device__ __forceinline__ float filler(int iters, float* gout) {
  __shared__ float4 lds[512];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  f32x2 p0 = {x0, x1}, p1 = {x2, x3}, pm = {1.0001f, 1.0001f}, pa = {0.5f, 0.5f};
  const float m = 1.0001f, ad = 0.5f;
  float4 l0 = {0, 0, 0, 0};
  unsigned u0 = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if constexpr (KIND == 0) {  // 4 x v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(ad));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(ad));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(ad));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(ad));
      } else if constexpr (KIND == 1) {  // 2 x v_pk_fma_f32 (same flops)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pa));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pm), "v"(pa));
      } else if constexpr (KIND == 2) {  // 4 x v_exp_f32
        asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x2));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
      } else if constexpr (KIND == 3) {  // 4 x integer VALU
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u0));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(x1));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u0));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(x2));
      } else if constexpr (KIND == 4) {  // 2 x ds_read_b128
        l0.x += lds[(threadIdx.x + i) & 511].x;
        l0.y += lds[(threadIdx.x + i + 64) & 511].y;
      } else if constexpr (KIND == 5) {  // 1 x global_store_dwordx4 (streaming)
        ((float4*)gout)[((size_t)blockIdx.x * 512 + threadIdx.x) + (size_t)((i * 3 + r) & 1023) * 256 * 512] = l0;
      } else if constexpr (KIND == 6) {  // 4 x v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x0), "v"(x1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x2), "v"(x3));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x1), "v"(x2));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x3), "v"(x0));
      }
    }
  }
  return x0 + x1 + x2 + x3 + p0[0] + p0[1] + p1[0] + p1[1] + l0.x + l0.y + (float)u0;
}

template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, float* gout, unsigned long long* tm) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = wall_clock64();
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
      f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      for (int i = 0; i < iters; ++i) { M(c0); M(c1); M(c2); M(c3); }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else if (mode & 2) {
    r = filler<KIND>(iters, gout);
  }
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) tm[blockIdx.x * 8 + wave] = t1 - t0;
  if (r == 12345.678f) out[0] = r;
}

template <int KIND>
void run(const char* name, float* out, float* gout, unsigned long long* tm) {
  const int iters = 20000;
  double res[4][2];
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(tm, 0, 256 * 8 * 8);
      hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out, gout, tm);
      hipDeviceSynchronize();
    }
    unsigned long long h[8];
    hipMemcpy(h, tm, sizeof(h), hipMemcpyDeviceToHost);
    res[mode][0] = h[0] / 100.0; res[mode][1] = h[4] / 100.0;
  }
  printf("%-28s MFMA alone %7.1f us | filler alone %7.1f us | together: MFMA %7.1f us, filler %7.1f us  (sum %7.1f, max %7.1f)\n", name,
         res[1][0], res[2][1], res[3][0], res[3][1], res[1][0] + res[2][1], res[1][0] > res[2][1] ? res[1][0] : res[2][1]);
}

int main() {
  float *out, *gout; unsigned long long* tm;
  hipMalloc(&out, 4); hipMalloc(&tm, 256 * 8 * 8); hipMalloc(&gout, (size_t)1024 * 256 * 512 * 16);
  run<0>("4 v_fma_f32", out, gout, tm);
  run<1>("2 v_pk_fma_f32", out, gout, tm);
  run<2>("4 v_exp_f32", out, gout, tm);
  run<3>("4 int VALU", out, gout, tm);
  run<4>("2 ds_read", out, gout, tm);
  run<5>("1 global_store_dwordx4", out, gout, tm);
  run<6>("4 v_cvt_pk_bf16_f32", out, gout, tm);
  return 0;
}
