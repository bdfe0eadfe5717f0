// GPU microbenchmark: does the SHAPE of the matrix instruction change the energy of a multiply-accumulate?  The split modes run at the
// board's power cap with the matrix pipe taking about half of the dynamic joules (profiles/r04/energy_model.json), so joules per MAC
// is what the step time follows.  Register-only loops on every SIMD, random fp16 / bf16 operands, each case for a few seconds while
// tools/power_sampler.py samples package power and shader clock beside it (tools/mfma_shapes.sh):
//   v_mfma_f32_16x16x32_{f16,bf16}: 8 passes, 8 192 MACs per wave-instruction - the instruction every kernel here uses
//   v_mfma_f32_32x32x16_{f16,bf16}: 16 passes, 16 384 MACs - the same MACs per pass, half the operand-register reads per MAC
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_shapes tools/micro/mfma_shapes.hip ; run: tools/micro/mfma_shapes [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline float rnd(unsigned seed) { return ((int)(hash32(seed) >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }

// MODE 0: f16 16x16x32, 1: f16 32x32x16, 2: bf16 16x16x32, 3: bf16 32x32x16.  8 (16x16) / 4 (32x32) independent accumulator tiles per wave.
template <int MODE>
__global__ __launch_bounds__(512) void k_mfma(float* out, int iters) {
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  h8 ah[4], bh[4];
  b8 ab[4], bb[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      const float va = rnd(id * 64 + i * 8 + j), vb = rnd(id * 64 + 32 + i * 8 + j);
      ah[i][j] = (_Float16)va; bh[i][j] = (_Float16)vb; ab[i][j] = (__bf16)va; bb[i][j] = (__bf16)vb;
    }
  float s = 0.f;
  if constexpr (MODE == 0 || MODE == 2) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 3], bh[(i >> 1) & 3], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i & 3], bb[(i >> 1) & 3], acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 3], bh[(i >> 1) & 1], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[i & 3], bb[(i >> 1) & 1], acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  if (s == 1234.5678f) out[id] = s;
}

static double now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

template <class F>
static void run_case(const char* name, double secs, double macs_per_launch, F launch) {
  launch(); hipDeviceSynchronize();
  const double t0 = now();
  long n = 0;
  while (now() - t0 < secs) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    n += 20;
  }
  const double t1 = now();
  printf("CASE %s %.3f %.3f %.6e MAC/s\n", name, t0, t1, macs_per_launch * n / (t1 - t0));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 5.0;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* sink; hipMalloc(&sink, (size_t)2 * cus * 512 * 4);
  { const double t0 = now(); sleep((unsigned)secs); printf("CASE idle %.3f %.3f 0 none\n", t0, now()); fflush(stdout); }
  const int it = 4000;
  const double waves = (double)2 * cus * 8;
  for (int rep = 0; rep < 2; ++rep) {
    run_case("f16_16x16x32", secs, waves * it * 8 * 8192.0, [&] { hipLaunchKernelGGL(k_mfma<0>, dim3(2 * cus), dim3(512), 0, 0, sink, it); });
    run_case("f16_32x32x16", secs, waves * it * 4 * 16384.0, [&] { hipLaunchKernelGGL(k_mfma<1>, dim3(2 * cus), dim3(512), 0, 0, sink, it); });
    run_case("bf16_16x16x32", secs, waves * it * 8 * 8192.0, [&] { hipLaunchKernelGGL(k_mfma<2>, dim3(2 * cus), dim3(512), 0, 0, sink, it); });
    run_case("bf16_32x32x16", secs, waves * it * 4 * 16384.0, [&] { hipLaunchKernelGGL(k_mfma<3>, dim3(2 * cus), dim3(512), 0, 0, sink, it); });
  }
  return 0;
}
