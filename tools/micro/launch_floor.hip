// GPU microbenchmark: what one kernel boundary costs on MI355X, as the launch-to-launch period of back-to-back launches on one
// stream (plain launches and a captured hipGraph of 50 launches), for
//   (a) an empty kernel with the workgroup shapes of the step's kernels (256 workgroups x {64, 512, 1024} threads x {0, 64, 155} KB LDS)
//   (b) a kernel that only writes N MB (plain stores / nt stores): the dirty lines of the XCD L2s are written back when it ends
//   (c) a kernel that busy-waits T us in every workgroup, followed by (b): does the flush overlap anything?
// build: hipcc -O3 --offload-arch=gfx950 -o launch_floor tools/micro/launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(int* p) {
  extern __shared__ char lds[];
  if (p && threadIdx.x == 9999) p[0] = lds[0];
}
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void k_write(float4* out, long n_per_wg) {  // every workgroup writes n_per_wg float4 (contiguous)
  v4f* o = (v4f*)out + (long)blockIdx.x * n_per_wg;
  const v4f v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (long i = threadIdx.x; i < n_per_wg; i += blockDim.x) {
    if (NT) __builtin_nontemporal_store(v, o + i);
    else o[i] = v;
  }
}
__global__ void k_read(const float4* in, long n_per_wg, float* sink) {
  const float4* p = in + (long)blockIdx.x * n_per_wg;
  float s = 0.f;
  for (long i = threadIdx.x; i < n_per_wg; i += blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) sink[0] = s;
}

// every XCD reads the whole buffer of n float4 (workgroup b runs on XCD b % 8 and reads slice b / 8 of 32): what a weight matrix that
// all eight L2s pull from the Infinity Cache costs, against the same number of fabric bytes of distinct data (k_read)
__global__ void k_read_shared(const float4* in, long n, float* sink) {
  const long per = n / 32;
  const float4* p = in + (long)(blockIdx.x / 8) * per;
  float s = 0.f;
  for (long i = threadIdx.x; i < per; i += blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) sink[0] = s;
}

template <class F>
static float per_launch_us(F launch, hipStream_t st, int iters, bool graph) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  if (!graph) {
    for (int i = 0; i < 10; ++i) launch(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch(st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
  } else {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < iters; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f / iters;
}

int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 50;
  for (int graph = 0; graph < 2; ++graph) {
    printf("--- %s\n", graph ? "hipGraph of 50 kernel nodes" : "plain launches");
    for (int nt : {64, 512, 1024})
      for (int lds_kb : {0, 64, 155}) {
        const float us = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(nt), lds_kb * 1024, s, (int*)nullptr); }, st, iters, graph);
        printf("empty 256 x %4d threads, %3d KB LDS: %6.2f us per launch\n", nt, lds_kb, us);
      }
    for (int wgs : {2048}) {
      const float us = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, s, (int*)nullptr); }, st, iters, graph);
      printf("empty %d x 256 threads: %6.2f us per launch\n", wgs, us);
    }
    float4* buf; hipMalloc(&buf, 512l << 20);
    float* sink; hipMalloc(&sink, 64);
    for (int mb : {1, 8, 27, 54, 109}) {
      const long n_per_wg = ((long)mb << 20) / 16 / 256;
      const float w = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_write<0>, dim3(256), dim3(512), 0, s, buf, n_per_wg); }, st, iters, graph);
      const float wnt = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_write<1>, dim3(256), dim3(512), 0, s, buf, n_per_wg); }, st, iters, graph);
      const float wr = per_launch_us([&](hipStream_t s) {
        hipLaunchKernelGGL(k_write<0>, dim3(256), dim3(512), 0, s, buf, n_per_wg);
        hipLaunchKernelGGL(k_read, dim3(256), dim3(512), 0, s, buf, n_per_wg, sink); }, st, iters, graph);
      const float wrnt = per_launch_us([&](hipStream_t s) {
        hipLaunchKernelGGL(k_write<1>, dim3(256), dim3(512), 0, s, buf, n_per_wg);
        hipLaunchKernelGGL(k_read, dim3(256), dim3(512), 0, s, buf, n_per_wg, sink); }, st, iters, graph);
      printf("write %3d MB: plain %6.2f us (%.2f TB/s)  nt %6.2f us | write + read-back pair: plain %6.2f us  nt %6.2f us\n", mb, w, mb * 1.048576e6 / w / 1e6,
             wnt, wr, wrnt);
    }
    for (int mb : {4, 8, 13}) {  // W1 (4.2 MB), W1 + W2, all weights of a layer (12.6 MB) in the split modes
      const long n = ((long)mb << 20) / 16;
      const float sh = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_read_shared, dim3(256), dim3(512), 0, s, buf, n, sink); }, st, iters, graph);
      const float di = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_read, dim3(256), dim3(512), 0, s, buf, n * 8 / 256, sink); }, st, iters, graph);
      printf("read %2d MB by each of the 8 XCDs (%3d MB through the fabric): %6.2f us | %3d MB of distinct data: %6.2f us\n", mb, mb * 8, sh, mb * 8, di);
    }
    // is 6 TB/s a limit of the fabric or of 256 CUs x 24 GB/s?  the same 109 MB moved by 64 / 128 / 256 workgroups (one per CU)
    for (int wgs : {64, 128, 256}) {
      const long n_per_wg = (109l << 20) / 16 / wgs;
      const float w = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_write<0>, dim3(wgs), dim3(1024), 0, s, buf, n_per_wg); }, st, iters, graph);
      const float r = per_launch_us([&](hipStream_t s) { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(1024), 0, s, buf, n_per_wg, sink); }, st, iters, graph);
      printf("109 MB by %3d workgroups x 1024 threads: write %6.2f us (%.2f TB/s)  read %6.2f us (%.2f TB/s)\n", wgs, w, 114.3 / w, r, 114.3 / r);
    }
    hipFree(buf); hipFree(sink);
  }
  return 0;
}
