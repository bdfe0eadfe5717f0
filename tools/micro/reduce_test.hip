// GPU check of the DPP / permlane-swap reductions in csrc/tamf_device.h against plain loops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../oakink2-tamf_amd/csrc/tamf_device.h"
__global__ void k(const float* in, float* out) {
  const float v = in[threadIdx.x];
  out[threadIdx.x] = wave_reduce<RedSum>(v);
  out[64 + threadIdx.x] = wave_reduce<RedMax>(v);
  out[128 + threadIdx.x] = wave_reduce<RedMin>(v);
  out[192 + threadIdx.x] = groups_reduce<RedSum>(v);
  out[256 + threadIdx.x] = groups_reduce<RedMin>(v);
}
int main() {
  float h[64], o[320], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 10.f + i * 0.01f;
  hipMalloc(&di, 256); hipMalloc(&dout, 320 * 4);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 320 * 4, hipMemcpyDeviceToHost);
  float s = 0, mx = -1e30f, mn = 1e30f;
  for (int i = 0; i < 64; ++i) { s += h[i]; mx = fmaxf(mx, h[i]); mn = fminf(mn, h[i]); }
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float gs = 0, gm = 1e30f;
    for (int g = 0; g < 4; ++g) { gs += h[(i & 15) + 16 * g]; gm = fminf(gm, h[(i & 15) + 16 * g]); }
    if (fabsf(o[i] - s) > 1e-3f || o[64 + i] != mx || o[128 + i] != mn || fabsf(o[192 + i] - gs) > 1e-4f || o[256 + i] != gm) {
      if (bad < 8) printf("lane %d: sum %f/%f max %f/%f min %f/%f gsum %f/%f gmin %f/%f\n", i, o[i], s, o[64 + i], mx, o[128 + i], mn, o[192 + i], gs, o[256 + i], gm);
      ++bad;
    }
  }
  printf("bad lanes: %d\n", bad);
  return bad != 0;
}
