// Persistent, software-pipelined GEMM for the wide 128 x 128-tile launches (QKV, FFN1, FFN2 in bf16x3).
//
// The 2-barrier kernel of tamf_gemm.h runs every tile as prologue -> K loop -> epilogue, and the two workgroups of a CU are
// in the same phase because they start together: the matrix pipe idles through the VALU/store-bound epilogue (28 % of a
// bf16x3 FFN1 tile, 40 % in bf16).  Here ONE workgroup of 4 waves (one per SIMD, the whole 512-register file each) owns a CU and walks its tiles:
//   * K tiles stream through a 3-stage LDS ring: while K tile q is multiplied from registers, the fragments of q+1 are read
//     from LDS into a second register set, q+2 is in flight and the LDS-DMA of q+3 is issued; the wait before the
//     per-K-tile barrier is a counted `s_waitcnt vmcnt` (never 0 inside the loop), and the stream runs across tile
//     boundaries, so a tile has no cold prologue;
//   * the finished accumulators are parked in an fp32 C tile that does NOT alias the ring (96 + 64 = 160 KiB of LDS), and
//     the epilogue of tile i is executed in 8 row slices per thread BETWEEN the MFMAs of tile i+1's K loop.
// Epilogues plug in through  pipe_ok() / pipe_bias() / item8()  (8 consecutive columns of one row, bias already added).
//
// STATUS (round 1): correct (tests/test_hip_forward.py passes with TAMF_GEMM_PIPE=1) but NOT faster, therefore off by
// default.  FFN1 bf16x3, M = 13312: 2-barrier kernel 90 us; this kernel 168 us with 4 waves (below) and 133 us as 8 waves x
// (64 x 32) without the register double buffer.  With one workgroup per CU nothing covers a wave's own issue costs: the 8
// LDS-DMA pieces per wave and K tile cost about as many issue cycles as the 48 MFMAs they feed, and every LDS latency and
// barrier skew is exposed, where the two independent workgroups of the 2-barrier kernel fill each other's gaps.  The
// epilogue overlap itself works (the matrix pipe does not wait for it), so the next step is instruction-level placement
// of the DMA issues between the MFMAs (sched_group_barrier / asm), not a different structure.
#pragma once
#include "tamf_gemm.h"

constexpr int PIPE_BM = 128, PIPE_BN = 128, PIPE_NW = 4, PIPE_NST = 3;
constexpr int PIPE_A_BYTES = PIPE_BM * GEMM_BKB;                  // 16 KiB
constexpr int PIPE_STAGE = (PIPE_BM + PIPE_BN) * GEMM_BKB;       // 32 KiB
constexpr int PIPE_C_OFF = PIPE_NST * PIPE_STAGE;                // 96 KiB
constexpr int PIPE_SMEM = PIPE_C_OFF + PIPE_BM * PIPE_BN * 4;    // 160 KiB
constexpr int PIPE_PPG = 8;                                      // LDS-DMA pieces per wave per K tile (4 A + 4 W)

// C tile: fp32 [128][128]; the 16-byte chunk c of row r lives at chunk position c ^ (r & 15) (conflict-free float4 parking
// of the swapped-product accumulators: the 16 lanes of a lane group hold 16 different rows of one chunk column)
TAMF_DEV int pipe_c_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

#define TAMF_PIPE_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One K tile for one wave.  `cur` / `nxt` / `ct` are three disjoint LDS regions; the __restrict__ qualifiers on this inlined
// helper give hipcc the alias scopes without which it would put an s_waitcnt vmcnt(0) in front of every LDS access that
// follows an LDS-DMA in program order (tamf_attn.h AttnBlock uses the same device).  ITEM: also run one epilogue item of the
// pending tile; LAST: last K tile of the current tile - request its bias vector (asm load: the compiler must not wait for
// it with vmcnt(0)), and after the MFMAs park the accumulators in the C tile (behind a barrier: every wave is done reading
// the previous tile's C values by then).
struct PipeItem {
  int m0, n0;    // origin of the pending tile
  f32x4 b0, b1;  // its bias for this thread's 8 columns
};
struct PipeFrags {
  int4 a[4][2], w[4][2];
};
// the 8 LDS-DMA pieces of a wave for one K tile: pieces wave + 4 i of the A and of the W tile
TAMF_DEV void pipe_issue(char* st, const char* const (&pa)[4], const char* const (&pw)[4], long ko, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16<0>(pa[i] + ko, st + (wave + PIPE_NW * i) * 1024);
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16<0>(pw[i] + ko, st + PIPE_A_BYTES + (wave + PIPE_NW * i) * 1024);
}
TAMF_DEV void pipe_read_frags(const char* st, PipeFrags& f, int a_frag, int w_frag, int c0, int c1) {
  constexpr int BKB = GEMM_BKB;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    f.a[mi][0] = *(const int4*)(st + a_frag + mi * 16 * BKB + c0);
    f.a[mi][1] = *(const int4*)(st + a_frag + mi * 16 * BKB + c1);
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    f.w[ni][0] = *(const int4*)(st + w_frag + ni * 16 * BKB + c0);
    f.w[ni][1] = *(const int4*)(st + w_frag + ni * 16 * BKB + c1);
  }
}
template <class Op, class Epi, bool ITEM, bool LAST>
TAMF_DEV void pipe_ktile(const char* __restrict__ rd, char* __restrict__ wr, char* __restrict__ ct, const bool do_read,
                         const bool issue, const char* const (&pa)[4], const char* const (&pw)[4], const long ko, const int wave,
                         const int a_frag, const int w_frag, const int c0, const int c1, f32x4 (&acc)[4][4], const PipeFrags& fc,
                         PipeFrags& fn, const Epi& epi, const PipeItem& pt, const int item_row, const int cg, const int M,
                         const float* bias_next, f32x4& nb0, f32x4& nb1, const int (&park)[4]) {
  if constexpr (LAST) {
    if (bias_next) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nb0) : "v"(bias_next));
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(nb1) : "v"(bias_next));
    }
  }
  // fragments of the NEXT K tile: their LDS latency is covered by this tile's MFMAs
  if (do_read) pipe_read_frags(rd, fn, a_frag, w_frag, c0, c1);
  float v[8];
  if constexpr (ITEM) {
    const float4 x0 = *(const float4*)(ct + pipe_c_off(item_row, 2 * cg));
    const float4 x1 = *(const float4*)(ct + pipe_c_off(item_row, 2 * cg + 1));
    v[0] = x0.x + pt.b0[0]; v[1] = x0.y + pt.b0[1]; v[2] = x0.z + pt.b0[2]; v[3] = x0.w + pt.b0[3];
    v[4] = x1.x + pt.b1[0]; v[5] = x1.y + pt.b1[1]; v[6] = x1.z + pt.b1[2]; v[7] = x1.w + pt.b1[3];
  }
  // the LDS-DMA pieces of K tile q+3 are issued between the MFMA groups: a piece costs ~100 issue cycles of this wave,
  // which the matrix pipe spends on the 12 MFMAs queued before it
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) Op::mma(acc[mi][ni], fc.w[ni], fc.a[mi]);
    if (issue) {
      glds16<0>(pa[ni] + ko, wr + (wave + PIPE_NW * ni) * 1024);
      glds16<0>(pw[ni] + ko, wr + PIPE_A_BYTES + (wave + PIPE_NW * ni) * 1024);
    }
  }
  if constexpr (ITEM) {
    const int gr = pt.m0 + item_row;
    if (gr < M) epi.item8(v, gr, pt.n0 + cg * 8);
  }
  if constexpr (LAST) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f32x4 a = acc[mi][ni];
        *(float4*)(ct + park[ni] + mi * 16 * 512) = make_float4(a[0], a[1], a[2], a[3]);
        acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
}

template <class Op, class Epi>
__global__ __launch_bounds__(PIPE_NW * 64, 1) void gemm_pipe_kernel(const GemmArgs<Op> ga, const Epi epi) {
  constexpr int BKB = GEMM_BKB, BM = PIPE_BM, BN = PIPE_BN, NST = PIPE_NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int M = ga.M, ntn = ga.N / BN, ntm = (M + BM - 1) / BM, n_tiles = ntn * ntm;
  const int KT = (ga.K * Op::EB) / BKB;  // even, >= 8 (launcher)
  const int G = gridDim.x, xr = xcd_remap(blockIdx.x, G);
  const int n_my = xr < n_tiles ? (n_tiles - xr + G - 1) / G : 0;
  const int Q = n_my * KT;
  if (Q == 0) return;
  const char* Ab = (const char*)ga.A;
  const char* Wb = (const char*)ga.W;

  // fragment addressing (as tamf_gemm.h): lane (lr, g) reads chunks g and 4+g of tile row lr (+16 per MFMA tile)
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = (wm0 + lr) * BKB, w_frag = PIPE_A_BYTES + (wn0 + lr) * BKB;
  // LDS-DMA pieces of this lane: piece p = wave + 4 i covers tile rows 8p .. 8p+7; the swizzle goes on the source chunk
  const int prow = lane >> 3, pch = lane & 7;
  int prw[4], psc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    prw[i] = (wave + PIPE_NW * i) * 8 + prow;
    psc[i] = (pch ^ swz_chunk<BKB>(prw[i])) << 4;
  }
  // parking positions of acc[mi][ni]: row wm0 + 16 mi + lr, chunk wn0/4 + 4 ni + g, swizzled by row & 15 == lr
  int park[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) park[ni] = (wm0 + lr) * 512 + (((wn0 >> 2) + ((4 * ni + g) ^ lr)) << 4);

  // ---- prefetch side of the K-tile stream (runs three K tiles ahead of the multiply, across tile boundaries)
  int pf_q = 0, pf_kt = 0, pf_s = 0;
  const char *pa[4], *pw[4];
#define TAMF_PF_TILE()                                                          \
  {                                                                             \
    const int t_ = pf_s * G + xr, m0_ = (t_ / ntn) * BM, n0_ = (t_ % ntn) * BN; \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                          \
      int gr_ = m0_ + prw[i_];                                                  \
      gr_ = gr_ < M ? gr_ : M - 1;                                              \
      pa[i_] = Ab + (long)gr_ * ga.lda * Op::EB + psc[i_];                      \
      pw[i_] = Wb + (long)(n0_ + prw[i_]) * ga.ldw * Op::EB + psc[i_];          \
    }                                                                           \
  }
#define TAMF_PF_ADVANCE()          \
  {                                \
    ++pf_q;                        \
    if (++pf_kt == KT) {           \
      pf_kt = 0;                   \
      ++pf_s;                      \
      if (pf_q < Q) TAMF_PF_TILE() \
    }                              \
  }
  TAMF_PF_TILE()
#pragma unroll
  for (int i = 0; i < NST; ++i) {  // the first three groups (Q >= 8)
    pipe_issue(smem + i * PIPE_STAGE, pa, pw, (long)pf_kt * BKB, wave);
    TAMF_PF_ADVANCE()
  }
  // fragments of K tile 0: group 0 has landed when at most groups 1 and 2 (16 loads) are outstanding
  TAMF_PIPE_WAIT_BARRIER(16);
  PipeFrags fA, fB;
  pipe_read_frags(smem, fA, a_frag, w_frag, c0, c1);

  // ---- epilogue items: thread = 8 columns (cg) of rows 16 j + tid/16, j = 0..7, taken in K-tile iteration
  // ((2 j + o) * KT) >> 4 of the NEXT tile's K loop (o = wave & 1 spreads the stores of the four waves)
  const int cg = tid & 15, rsub = tid >> 4;
  const int o = wave & 1;
  const float* bias = epi.pipe_bias();
  PipeItem pt;
  pt.m0 = pt.n0 = 0;
  pt.b0 = pt.b1 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 nb0 = pt.b0, nb1 = pt.b0;
  int nm0 = 0, nn0 = 0;
  bool pending = false, fresh = false;
  char* ct = smem + PIPE_C_OFF;
  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  int q = 0, st = 0;  // st = stage holding K tile q
  for (int s = 0; s < n_my; ++s) {
    const int t = s * G + xr, m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
    const float* bias_next = bias ? bias + n0 + cg * 8 : nullptr;
    int jn = 0;                       // next item of the pending tile
    int jk = (o * KT) >> 4;           // ... and the K-tile iteration it is due in
    for (int kt = 0; kt < KT; kt += 2) {
#define TAMF_PIPE_STEP(KT_, FC, FN)                                                                                          \
  {                                                                                                                          \
    /* K tile q+1 has landed once at most the 8 loads of group q+2 are outstanding (loads return in order); the bias loads */ \
    /* of the tile parked last were issued before group q+2 and have landed too */                                           \
    if (q + 2 < Q) TAMF_PIPE_WAIT_BARRIER(8);                                                                                \
    else TAMF_PIPE_WAIT_BARRIER(0);                                                                                          \
    if (fresh) { /* first K tile after a park: the bias registers requested by asm are valid now */                          \
      asm volatile("" : "+v"(nb0), "+v"(nb1));                                                                               \
      pt.b0 = nb0;                                                                                                           \
      pt.b1 = nb1;                                                                                                           \
      pt.m0 = nm0;                                                                                                           \
      pt.n0 = nn0;                                                                                                           \
      fresh = false;                                                                                                         \
      pending = true;                                                                                                        \
    }                                                                                                                        \
    const bool issue_ = pf_q < Q, read_ = q + 1 < Q;                                                                         \
    const int st1_ = st + 1 >= NST ? 0 : st + 1;                                                                             \
    const char* rd_ = smem + st1_ * PIPE_STAGE;                                                                              \
    char* wr_ = smem + st * PIPE_STAGE;                                                                                      \
    const long ko_ = (long)pf_kt * BKB;                                                                                      \
    const bool item_ = pending && jn < 8 && (KT_) == jk;                                                                     \
    const bool last_ = (KT_) == KT - 1;                                                                                      \
    const int irow_ = jn * 16 + rsub;                                                                                        \
    if (item_) {                                                                                                             \
      if (last_) TAMF_PIPE_CALL(true, true, FC, FN);                                                                         \
      else TAMF_PIPE_CALL(true, false, FC, FN);                                                                              \
      ++jn;                                                                                                                  \
      jk = ((2 * jn + o) * KT) >> 4;                                                                                         \
    } else {                                                                                                                 \
      if (last_) TAMF_PIPE_CALL(false, true, FC, FN);                                                                        \
      else TAMF_PIPE_CALL(false, false, FC, FN);                                                                             \
    }                                                                                                                        \
    if (issue_) TAMF_PF_ADVANCE()                                                                                            \
    st = st1_;                                                                                                               \
    ++q;                                                                                                                     \
  }
#define TAMF_PIPE_CALL(IT, LA, FC, FN)                                                                                          \
  pipe_ktile<Op, Epi, IT, LA>(rd_, wr_, ct, read_, issue_, pa, pw, ko_, wave, a_frag, w_frag, c0, c1, acc, FC, FN, epi, pt, irow_, \
                              cg, M, bias_next, nb0, nb1, park)
      TAMF_PIPE_STEP(kt, fA, fB)
      TAMF_PIPE_STEP(kt + 1, fB, fA)
#undef TAMF_PIPE_CALL
#undef TAMF_PIPE_STEP
    }
    // the tile is parked; its items start after the next K tile's barrier (or in the drain below)
    nm0 = m0;
    nn0 = n0;
    pending = false;
    fresh = true;
  }
  // the last tile's epilogue
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  asm volatile("" : "+v"(nb0), "+v"(nb1));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = j * 16 + rsub, gr = nm0 + row;
    const float4 x0 = *(const float4*)(ct + pipe_c_off(row, 2 * cg));
    const float4 x1 = *(const float4*)(ct + pipe_c_off(row, 2 * cg + 1));
    float v[8] = {x0.x + nb0[0], x0.y + nb0[1], x0.z + nb0[2], x0.w + nb0[3], x1.x + nb1[0], x1.y + nb1[1], x1.z + nb1[2], x1.w + nb1[3]};
    if (gr < M) epi.item8(v, gr, nn0 + cg * 8);
  }
#undef TAMF_PF_TILE
#undef TAMF_PF_ADVANCE
}
