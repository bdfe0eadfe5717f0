// Small tiles with a DEEP K pipeline, for launches of a few tiles (a few clips per call - one, in the reference's own launcher,
// launch/sample.py:202-229).  Such a launch cannot fill the chip whatever the tile shape; what it pays is LATENCY: in the
// double-buffered gemm_tile (tamf_gemm.h) one K tile is in flight while one is multiplied, so a K interval costs a whole
// L2 -> LDS round trip - 0.55 us, 64 of them in FFN2 (K = 2048 in the split modes): 35 us of a 85-us layer at one clip
// (profiles/r05/small_batch_resid_c25.txt), with every MFMA of the workgroup done in a tenth of that.  Here NSTG stages hold
// NSTG - 1 K tiles in flight; the wait in front of a barrier is counted (s_waitcnt vmcnt(N): everything but the NSTG - 2 youngest
// tiles) the way the clip kernel counts (tamf_gemm_clip.h), the K loop issues the SAME number of requests in every interval
// (past the end of K: the last K tile again, into a stage nobody reads any more), and the fragment reads / requests sit in a helper
// whose __restrict__ stage pointers keep hipcc from putting a vmcnt(0) in front of LDS reads that follow an LDS-DMA.
// Same K order per output element as every other tile shape here (K tiles in order, the MFMAs of a K tile in order): the same bits.
#pragma once
#include "tamf_gemm.h"
#include "tamf_gemm_clip.h"

template <int BM, int BN, int NSTG>
struct GemmSmemDeep {
  static constexpr int LDC = BN + 4;
  static constexpr int STAGE = (BM + BN) * GEMM_BKB;
  static constexpr int CBYTES = BM * LDC * 4;
  static constexpr int BYTES = (NSTG * STAGE > CBYTES) ? NSTG * STAGE : CBYTES;
  static constexpr int STATS_OFF = BYTES, TOTAL = BYTES + BM * 8;
};

// one K interval: all fragments of the K tile in `cur`, then this wave's pieces of a later K tile into `nxt`, then the MFMAs
template <class Op, int MI, int NI, int A_PW, int W_PW, int NWV, int A_PIECES, int A_BYTES>
TAMF_DEV void deep_ktile(const char* __restrict__ cur, char* __restrict__ nxt, const char* Ab, const char* Wb, const unsigned (&a_off)[A_PW],
                         const unsigned (&w_off)[W_PW], long kbyte_next, int wave, int a_frag, int w_frag, int c0, int c1, f32x4 (&acc)[MI][NI]) {
  constexpr int BKB = GEMM_BKB;
  int4 af[MI][2], wf[NI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    af[mi][0] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c0);
    af[mi][1] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c1);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    wf[ni][0] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c0);
    wf[ni][1] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c1);
  }
#pragma unroll
  for (int ii = 0; ii < A_PW; ++ii) {
    const int q = wave + ii * NWV;
    if (A_PIECES % NWV == 0 || q < A_PIECES) glds16<0>(Ab + a_off[ii] + kbyte_next, nxt + q * 1024);
  }
#pragma unroll
  for (int ii = 0; ii < W_PW; ++ii) glds16<0>(Wb + w_off[ii] + kbyte_next, nxt + A_BYTES + (wave + ii * NWV) * 1024);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) Op::mma(acc[mi][ni], wf[ni], af[mi]);  // D rows = n (4g + reg), cols = m (lr)
  __builtin_amdgcn_sched_barrier(0);  // (nothing sinks below the counted wait that follows: tamf_gemm.h, gemm_tile)
}

template <class Op, int BM, int BN, int WGM, int WGN, int NSTG, class Epi>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_deep_kernel(const GemmArgs<Op> ga, const Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = GEMM_BKB;
  constexpr int NT = WGM * WGN * 64, NWV = WGM * WGN;
  constexpr int CPR = BKB / 16, RPI = 1024 / BKB;
  constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
  constexpr int A_PIECES = BM / RPI, W_PIECES = BN / RPI;
  constexpr int A_PW = (A_PIECES + NWV - 1) / NWV, W_PW = W_PIECES / NWV;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && W_PIECES % NWV == 0 && NSTG >= 3 && NSTG <= 6, "tile shape");
  typedef GemmSmemDeep<BM, BN, NSTG> SM;
  constexpr int A_BYTES = BM * BKB;
  constexpr int LA = NSTG - 1;  // K tiles in flight

  const int ntn = ga.N / BN;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntn) * BM, n0 = (lb % ntn) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int M = ga.M;
  const int KT = (ga.K * Op::EB) / BKB;
  const char* Ab = (const char*)ga.A;
  const char* Wb = (const char*)ga.W;

  unsigned a_off[A_PW], w_off[W_PW];
  const int prow = lane / CPR, pch = lane % CPR;
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave + i * NWV) * RPI + prow;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    a_off[i] = (unsigned)((long)gr * ga.lda * Op::EB + ((pch ^ swz_chunk<BKB>(row)) << 4));
  }
#pragma unroll
  for (int i = 0; i < W_PW; ++i) {
    const int row = (wave + i * NWV) * RPI + prow;
    w_off[i] = (unsigned)((long)(n0 + row) * ga.ldw * Op::EB + ((pch ^ swz_chunk<BKB>(row)) << 4));
  }
  // pieces this wave requests per K tile (wave-uniform): the A pieces go to the first A_PIECES % NWV waves when they do not divide
  constexpr int A_REM = A_PIECES % NWV;
  const bool a_hi = A_REM == 0 || wave < A_REM;
  constexpr int PW_HI = A_PW + W_PW, PW_LO = A_PW - 1 + W_PW;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = (wm0 + lr) * BKB;
  const int w_frag = A_BYTES + (wn0 + lr) * BKB;

  // prologue: K tiles 0 .. LA - 1 into stages 0 .. LA - 1 (past the end of K: the last K tile again - constant request counts)
#pragma unroll
  for (int p = 0; p < LA; ++p) {
    const long kb = (long)(p < KT ? p : KT - 1) * BKB;
    char* st = smem + p * SM::STAGE;
#pragma unroll
    for (int ii = 0; ii < A_PW; ++ii) {
      const int q = wave + ii * NWV;
      if (A_PIECES % NWV == 0 || q < A_PIECES) glds16<0>(Ab + a_off[ii] + kb, st + q * 1024);
    }
#pragma unroll
    for (int ii = 0; ii < W_PW; ++ii) glds16<0>(Wb + w_off[ii] + kb, st + A_BYTES + (wave + ii * NWV) * 1024);
  }
  float2* const rstat = (float2*)(smem + SM::STATS_OFF);
  if constexpr (Epi::ROWSTATS) ln_stage<NT, Epi::STAGE_AFF, (BM + NT / 4 - 1) / (NT / 4)>(epi.ln, epi.ctl.wscale, m0, BM, M, rstat, tid);
  // the whole prologue has landed: an EXPLICIT vmcnt(0) in front of the barrier (a plain __syncthreads() left it to hipcc's
  // fence lowering, and with no LayerNorm staged - layer 0's out-proj - nothing else forces a wait in front of the first LDS reads;
  // _isa_check.check_deep asserts the wait on the build's own assembly)
  clip_wait_vm<0>();
  clip_barrier_lds();

  int sc = 0, sn = LA % NSTG;
  for (int kt = 0; kt < KT; ++kt) {
    const int ktn = kt + LA < KT ? kt + LA : KT - 1;
    deep_ktile<Op, MI, NI, A_PW, W_PW, NWV, A_PIECES, A_BYTES>(smem + sc * SM::STAGE, smem + sn * SM::STAGE, Ab, Wb, a_off, w_off, (long)ktn * BKB,
                                                               wave, a_frag, w_frag, c0, c1, acc);
    // K tile kt + 1 has landed when all but the LA - 1 youngest batches of this wave's requests have (vmcnt retires in order)
    if (a_hi) clip_wait_vm<PW_HI*(LA - 1)>(); else clip_wait_vm<PW_LO*(LA - 1)>();
    clip_barrier_lds();
    sc = sc + 1 == NSTG ? 0 : sc + 1;
    sn = sn + 1 == NSTG ? 0 : sn + 1;
  }
  clip_wait_vm<0>();  // the redundant tail requests must not land in the C tile below
  clip_barrier_lds();

  float* Ct = (float*)smem;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const f32x4 v = acc[mi][ni];
      *(float4*)(Ct + (wm0 + mi * 16 + lr) * SM::LDC + wn0 + ni * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
  __syncthreads();
  if constexpr (Epi::ROWSTATS) epi.template run<BM, BN, NT>(Ct, SM::LDC, m0, n0, M, tid, rstat);
  else epi.template run<BM, BN, NT>(Ct, SM::LDC, m0, n0, M, tid);
}
