// GPU microbenchmark: what can a wave issue while ANOTHER wave of its SIMD streams MFMAs back to back?
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD, as the X / Y waves of the clip GEMM).  Waves 4-7 ("Y") issue N
// independent MFMAs in a row; waves 0-3 ("X") run a probe of 256 instructions of one kind - VALU adds, SALU adds, LDS reads, global loads -
// and time it with s_memtime; the probe alone (Y idle) is the baseline.  Priorities: X 0 / Y 2 (the clip kernel's), X 3 / Y 2, both 0.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/issue_overlap tools/micro/issue_overlap.hip ; run: tools/micro/issue_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND, int MF, int PACE = 0>  // PACE: s_nop wait states behind every MFMA of Y ; KIND: 0 VALU, 1 SALU, 2 LDS read, 3 global load ; MF: 0 f32 16x16x4, 1 f16 16x16x32
__global__ __launch_bounds__(512) void k(unsigned long long* out, const float* g, int y_on, int xprio, int yprio) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) lds[i] = (float)i;
  __syncthreads();
  if (wave >= 4) {
    if (!y_on) return;
    if (yprio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)lane * 0.01f, b = (float)(lane + 1) * 0.02f;
    h8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a + j); bh[j] = (_Float16)(b - j); }
    float fv[4] = {a, b, a + b, a - b};
    f4 fr = f4{0.f, 0.f, 0.f, 0.f};
    int fs = __builtin_amdgcn_readfirstlane(wave);
    unsigned long long ty0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < 40; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MF == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
        if (PACE == 101) { asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3" : "+v"(fv[0]), "+v"(fv[1]), "+v"(fv[2]), "+v"(fv[3])); }
        else if (PACE == 104) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(fv[0])); }
        else if (PACE >= 24) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
        else if (PACE >= 16) asm volatile("s_nop 7\n\ts_nop 7");
        else if (PACE >= 12) asm volatile("s_nop 7\n\ts_nop 3");
        else if (PACE >= 8) asm volatile("s_nop 7");
        else if (PACE >= 4) asm volatile("s_nop 3");
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    s += fv[0] + fv[1] + fv[2] + fv[3] + fr[0] + fr[1] + (float)fs;
    asm volatile("s_nop 0" ::"v"(s));
    unsigned long long ty1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) out[2048 + blockIdx.x * 4 + (wave - 4)] = ty1 - ty0;
    if (s == 1234.5f) out[4096 + tid] = 1;
    return;
  }
  if (xprio == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_sleep(20);  // (let Y get going: ~1300 cycles)
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float v = (float)lane;
  int sv = __builtin_amdgcn_readfirstlane(wave);
  if (KIND == 0) {
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
  } else if (KIND == 1) {
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sv));
  } else if (KIND == 2) {
    f4 r = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; ++i) { f4 q = *(volatile f4*)&lds[((lane + i * 64) & 1023) * 4]; r += q; }
    v = r[0] + r[1];
  } else {
    f4 r = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; ++i) { f4 q = *(const volatile f4*)&g[(blockIdx.x * 64 + i) * 256 + lane * 4]; r += q; }
    v = r[0] + r[1];
  }
  asm volatile("s_nop 0" ::"v"(v), "s"(sv));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (v == 1234.5f && sv == 77) out[8192] = 1;
}

template <int KIND, int MF, int PACE = 0>
static void run(const char* name, unsigned long long* out, const float* g, hipStream_t st) {
  static unsigned long long h[3072];
  const int cfg[3][3] = {{0, 0, 0}, {1, 0, 2}, {1, 3, 2}};
  const char* cn[3] = {"probe alone", "beside MFMAs, X prio 0 / Y 2", "beside MFMAs, X prio 3 / Y 2"};
  for (int c = 0; c < 3; ++c) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k<KIND, MF, PACE>), dim3(256), dim3(512), 0, st, out, g, cfg[c][0], cfg[c][1], cfg[c][2]);
      (void)hipStreamSynchronize(st);
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, sy = 0;
    for (int i = 0; i < 1024; ++i) { s += (double)h[i]; sy += (double)h[2048 + i]; }
    printf("%-44s pace %2d  %-30s probe %8.0f ticks   Y's 640 MFMAs %8.0f ticks\n", name, PACE, cn[c], s / 1024, cfg[c][0] ? sy / 1024 : 0.0);
  }
}

int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  unsigned long long* out; (void)hipMalloc(&out, 16384 * 8); (void)hipMemset(out, 0, 16384 * 8);
  float* g; (void)hipMalloc(&g, 256 * 64 * 256 * 4); (void)hipMemset(g, 0, 256 * 64 * 256 * 4);
  printf("== a probe wave beside another wave's back-to-back MFMA stream (same SIMD)\n");
  run<0, 0>("256 VALU adds | f32 16x16x4 MFMAs", out, g, st);
  run<1, 0>("256 SALU adds | f32 16x16x4 MFMAs", out, g, st);
  run<2, 0>("64 ds_read_b128 | f32 16x16x4 MFMAs", out, g, st);
  run<3, 0>("64 global_load_dwordx4 | f32 16x16x4 MFMAs", out, g, st);
  run<0, 1>("256 VALU adds | f16 16x16x32 MFMAs", out, g, st);
  run<2, 1>("64 ds_read_b128 | f16 16x16x32 MFMAs", out, g, st);
  run<3, 1>("64 global_load_dwordx4 | f16 16x16x32 MFMAs", out, g, st);
  printf("== the MFMA stream paced with s_nop behind every MFMA (the probe gets through, the stream pays the full nop)\n");
  run<0, 0, 4>("256 VALU adds | f32 MFMAs + s_nop 3", out, g, st);
  run<0, 0, 8>("256 VALU adds | f32 MFMAs + s_nop 7", out, g, st);
  run<2, 0, 8>("64 ds_read_b128 | f32 MFMAs + s_nop 7", out, g, st);
  run<0, 1, 4>("256 VALU adds | f16 MFMAs + s_nop 3", out, g, st);
  printf("== other instructions inside the MFMA wave's own stream (column Y: what they cost the stream)\n");
  run<1, 0, 0>("f32 MFMAs, nothing between", out, g, st);
  run<1, 0, 104>("f32 MFMAs, 1 v_add behind each (same wave)", out, g, st);
  run<1, 0, 101>("f32 MFMAs, 4 independent v_add behind each", out, g, st);
  run<1, 1, 0>("f16 MFMAs, nothing between", out, g, st);
  run<1, 1, 104>("f16 MFMAs, 1 v_add behind each (same wave)", out, g, st);
  return 0;
}
