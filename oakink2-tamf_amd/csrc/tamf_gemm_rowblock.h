// Row-block GEMM with the weights streamed L2 -> REGISTERS (no LDS staging of W) and LayerNorm in the epilogue:
//
//   X[m][:] = LayerNorm(resid[m][:] + bias + sum_k A[m][k] W[:][k]) * gamma + beta      N = 512 output columns = the whole row
//
// for the two LayerNorm-fused GEMMs of an encoder layer (attention out-proj, FFN2; interaction_segment_mdm.py:63-70 ->
// nn.TransformerEncoderLayer: x = norm1(x + sa(x)); x = norm2(x + ff(x))).
//
// Why (round 4, tools/micro/lds_fill.hip on MI355X).  A workgroup that owns whole rows must stream the whole weight panel (FFN2 in
// the split modes: 4 MB per workgroup).  Through LDS that stream costs twice: the LDS-DMA fill (the 64 x 512 tile of tamf_gemm.h
// reached 74 GB/s per CU, a 64 KB weight stage per 16 MFMAs per wave) and the fragment reads of the same bytes (in bf16 the LDS
// array, 256 B/clk, is as busy as the matrix pipe).  But W fragments are PRIVATE to a wave when the waves split the columns, so
// LDS buys nothing for them: each lane can fetch its 16-byte fragment pieces straight from L2.  A coalesced register stream
// reads 115 - 126 GB/s per CU from L2 (all 256 CUs at once, 30 TB/s), LDS-DMA 128 - 136 - against the 74 the staged K loop saw.
// Only the activation rows (shared by all waves: 8 KB per K tile at 64 rows) go through LDS.
//
// Weight layout ("fragment-major", packed once at weight load, rowblock_pack_kernel): [K tile kt][16-column tile jt][f][lane][16 B] -
// K-tile-major, so that what all waves of all workgroups read at about the same time (one K tile of all 512 columns) is ONE
// contiguous 64 KB: the first layout, [jt][kt], made every wave walk four streams 128 KB apart - 32 streams per workgroup that
// fall into the same L2 sets - and the 4 MB panel came from the Infinity Cache for every workgroup (FFN2 f16x3: 107 us, 832 MB at
// 7.8 TB/s).  For column tile jt and K tile kt the two 1-KiB pieces [f = 0, 1][lane][16 B] hold, for lane (lr, g), bytes [16 (4 f + g), +16) of the K tile's 128-byte operand row of
// output column col(jt, lr) - exactly the MFMA A-operand fragment of tamf_gemm.h (wf[ni][f]) - so a wave-instruction reads 1 KiB
// contiguous.  In the 16-bit modes col() applies clip_wperm<8> inside every 32 columns, so that a lane ends up with 8 consecutive
// output columns (16-byte operand stores); f32 keeps the natural order (4 consecutive columns = 16 bytes).
//
// Workgroup = 8 waves (2 per SIMD) x (MI x 16 rows) x 512 columns; wave w owns columns [64 w, 64 w + 64) = 4 column tiles and
// all MI row tiles: 4 MI accumulator tiles.  K loop, per 128-byte K tile: one counted wait + barrier (the activation stage has
// landed), 2 MI fragment reads, the LDS-DMA piece of the K tile DA ahead, the 8 fragment loads of the K tile two ahead (three
// register sets, loop unrolled by three), then 4 MI x (MFMAs per product) MFMAs in TERM-major order (the lo.hi terms of all tiles,
// then hi.lo, then hi.hi: per accumulator the same sequence as Op::mma, but consecutive MFMAs never depend on each other).
// Results are bit-identical to EpiLN::run / residual_ln_kernel: same products in the same K order per element, same
// association tree of the LayerNorm sums (ln_row_sum512, tamf_device.h).
#pragma once
#include "tamf_gemm.h"
#include "tamf_gemm_clip.h"

#ifndef TAMF_RB_ABL  // (measurement builds, tools/ab_build.sh: 1 = no activation requests in the K loop, 2 = no weight loads, 3 = neither, 4 = activation requests re-read the first K tiles, 5 = weight loads re-read the first four K tiles)
#define TAMF_RB_ABL 0
#endif

template <class Op>
struct RowblockArgs {
  const typename Op::elem_t* A;
  int lda;          // elements
  const char* Wp;   // fragment-major packed weights [KT][512 / 16][2][1024 B]
  int M, K;
};

template <class Op>
struct RowblockCfg {
  static constexpr int N = 512, NW = 8, JL = N / 16 / NW;  // 4 column tiles per wave
  static constexpr bool CH8 = Op::PREC != 0;                // 8 consecutive output columns per lane (16-bit planes) or 4 (f32)
  static constexpr int NSTG = 8, DA = 6;                    // LDS stages of the activation rows; a K tile is requested DA ahead
};

// output column of packed row i (0..15) of column tile jt
template <class Op>
TAMF_DEV int rowblock_col(int jt, int i) {
  if constexpr (RowblockCfg<Op>::CH8) return 32 * (jt >> 1) + clip_wperm<8>(16 * (jt & 1) + i);
  else return 16 * jt + i;
}

// standard operand matrix W [512][ldw] -> fragment-major packed copy (one thread per 16-byte piece)
template <class Op>
__global__ void rowblock_pack_kernel(const typename Op::elem_t* W, int ldw, int K, char* out) {
  const int KT = (K * Op::EB) / GEMM_BKB;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // ((kt * 32 + jt) * 2 + f) * 64 + lane
  if (idx >= (long)32 * KT * 128) return;
  const int lane = (int)(idx & 63), f = (int)((idx >> 6) & 1);
  const long t = idx >> 7;
  const int jt = (int)(t % 32), kt = (int)(t / 32);
  const int lr = lane & 15, g = lane >> 4;
  const char* src = (const char*)W + (long)rowblock_col<Op>(jt, lr) * ldw * Op::EB + (long)kt * GEMM_BKB + (4 * f + g) * 16;
  *(int4*)(out + idx * 16) = *(const int4*)src;
}

// One MFMA "term" of a product: per accumulator the terms run in the order of Op::mma
template <class Op, int T>
TAMF_DEV void rowblock_mma_term(f32x4& acc, const int4 (&w)[2], const int4 (&x)[2]) {
  if constexpr (Op::PREC == 0) {
    constexpr int f = T / 4, c = T % 4;
    const int wv = c == 0 ? w[f].x : c == 1 ? w[f].y : c == 2 ? w[f].z : w[f].w;
    const int xv = c == 0 ? x[f].x : c == 1 ? x[f].y : c == 2 ? x[f].z : x[f].w;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(wv), as_f(xv), acc, 0, 0, 0);
  } else if constexpr (!Op::SPLIT) {
    acc = Op::mfma1(w[T], x[T], acc);
  } else {
    if constexpr (T == 0) acc = Op::mfma1(w[1], x[0], acc);       // lo . hi
    else if constexpr (T == 1) acc = Op::mfma1(w[0], x[1], acc);  // hi . lo
    else acc = Op::mfma1(w[0], x[0], acc);                        // hi . hi
  }
}
template <class Op> struct RowblockTerms { static constexpr int value = Op::PREC == 0 ? 8 : (Op::SPLIT ? 3 : 2); };

template <class Op, int MI, int T>
TAMF_DEV void rowblock_mma_all(f32x4 (&acc)[MI][4], const int4 (&wq)[4][2], const int4 (&af)[MI][2]) {
  if constexpr (T < RowblockTerms<Op>::value) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) rowblock_mma_term<Op, T>(acc[mi][jl], wq[jl], af[mi]);
    rowblock_mma_all<Op, MI, T + 1>(acc, wq, af);
  }
}

TAMF_DEV float rowblock_swap16_sum(float v) {  // v + (value of the lane 16 further / nearer: lane groups g ^ 1)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned lo = a[0], hi = a[1];
  return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
TAMF_DEV float rowblock_swap32_sum(float v) {  // lane groups g ^ 2
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = a[0], hi = a[1];
  return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
// b[nq] of ln_row_sum512 for the 32 columns of which this lane holds the two quads (qe: column tile 2 nql, qo: 2 nql + 1), in the
// association order of tamf_device.h: a[g] = Q(g) + Q(g + 4), b = (a0 + a1) + (a2 + a3), Q(i) = the quad of columns 4 i .. 4 i + 3
template <class Op>
TAMF_DEV float rowblock_block_sum(float qe, float qo) {
  if constexpr (RowblockCfg<Op>::CH8) {
    // lane group g holds Q(2 g) and Q(2 g + 1); its partner g ^ 2 holds Q(2 g + 4 mod 8) ..: t0 = a[2 (g & 1)], t1 = a[2 (g & 1) + 1]
    const float t0 = rowblock_swap32_sum(qe), t1 = rowblock_swap32_sum(qo);
    return rowblock_swap16_sum(t0 + t1);
  } else {
    // lane group g holds Q(g) and Q(g + 4): a[g] in the lane; then the tree over the four groups (groups_reduce)
    return groups_reduce<RedSum>(qe + qo);
  }
}

// The two waves of a SIMD (w and w + 4) run half an interval apart, as the X / Y waves of tamf_gemm_clip.h: measured with every load
// removed (-DTAMF_RB_ABL=3) the lock-step form - all eight waves: barrier, fragment reads, MFMAs - took 1.24 us per K tile against
// 0.8 us of MFMA issue (f16x3, 64 rows): behind every barrier the matrix pipe waited for the LDS latency of both its waves.
//   "early" waves (0-3), interval k: fragments of K tile k, requests, MFMAs of K tile k
//   "late"  waves (4-7), interval k: requests, MFMAs of K tile k - 1 (fragments read at the end of interval k - 1; static
//                                     priority: they are ready at the barrier), then the fragments of K tile k
// `cur` = the LDS stage of K tile kt, `nxt` = the stage the K tile DA ahead is requested into (never the same: the __restrict__
// qualifiers keep hipcc from putting an s_waitcnt vmcnt(0) between the LDS-DMA and the fragment reads)
template <int MI>
TAMF_DEV void rowblock_read_a(const char* cur, int a_frag, int c0, int c1, int4 (&af)[MI][2]) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    af[mi][0] = *(const int4*)(cur + a_frag + mi * 16 * GEMM_BKB + c0);
    af[mi][1] = *(const int4*)(cur + a_frag + mi * 16 * GEMM_BKB + c1);
  }
}
TAMF_DEV void rowblock_request(char* nxt, const char* a_src, const char* w_src, int4 (&wq_nxt)[4][2], long w_tile_stride) {
  // (the fragment loads first, the piece behind them: the wait at the head of the next interval leaves the piece in flight)
  if (!(TAMF_RB_ABL & 2) || TAMF_RB_ABL == 4) {
#pragma unroll
    for (int jl = 0; jl < 4; ++jl) {
      wq_nxt[jl][0] = *(const int4*)(w_src + jl * w_tile_stride);
      wq_nxt[jl][1] = *(const int4*)(w_src + jl * w_tile_stride + 1024);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!(TAMF_RB_ABL & 1) || TAMF_RB_ABL == 4) glds16<0>(a_src, nxt);
}
template <class Op, int MI>
TAMF_DEV void rowblock_ktile_early(const char* __restrict__ cur, char* __restrict__ nxt, const char* a_src, const char* w_src,
                                   int a_frag, int c0, int c1, const int4 (&wq_cur)[4][2], int4 (&wq_nxt)[4][2], f32x4 (&acc)[MI][4], long w_tile_stride) {
  int4 af[MI][2];
  rowblock_read_a<MI>(cur, a_frag, c0, c1, af);
  rowblock_request(nxt, a_src, w_src, wq_nxt, w_tile_stride);
  __builtin_amdgcn_sched_barrier(0);  // (the requests go out at the head of the interval, not behind its MFMAs where hipcc would sink them)
  rowblock_mma_all<Op, MI, 0>(acc, wq_cur, af);
  __builtin_amdgcn_sched_barrier(0);  // (nothing sinks below the next K tile's wait + barrier)
}
template <class Op, int MI>
TAMF_DEV void rowblock_ktile_late(const char* __restrict__ cur, char* __restrict__ nxt, const char* a_src, const char* w_src,
                                  int a_frag, int c0, int c1, const int4 (&wq_prev)[4][2], int4 (&wq_nxt)[4][2], int4 (&af)[MI][2], f32x4 (&acc)[MI][4],
                                  long w_tile_stride) {
  rowblock_request(nxt, a_src, w_src, wq_nxt, w_tile_stride);
  __builtin_amdgcn_sched_barrier(0);
  rowblock_mma_all<Op, MI, 0>(acc, wq_prev, af);  // K tile kt - 1, fragments read one interval ago
  __builtin_amdgcn_sched_barrier(0);
  rowblock_read_a<MI>(cur, a_frag, c0, c1, af);    // K tile kt, for the next interval
  __builtin_amdgcn_sched_barrier(0);
}

template <class Op, int MI>
__global__ __launch_bounds__(512, 2) void rowblock_ln_kernel(const RowblockArgs<Op> ga, const EpiLN<Op> epi) {
  typedef RowblockCfg<Op> C;
  constexpr int BM = MI * 16, BKB = GEMM_BKB, N = C::N, NSTG = C::NSTG, DA = C::DA, STAGE = BM * BKB;
  constexpr int A_PIECES = BM / 8;  // 1-KiB pieces of one K tile of the activation rows
  static_assert(A_PIECES <= 8 && DA + 1 <= NSTG && DA >= 2, "row-block geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)(smem + NSTG * STAGE);  // [2][BM][16]: the per-(column tile, 32-column block) partial sums of the two passes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
  const int lr = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM, M = ga.M;
  const int KT = (ga.K * Op::EB) / BKB;  // (even: rowblock_applies)

  // activation rows: every wave requests ONE 1-KiB piece per K tile (8 rows x 128 B; source-side XOR swizzle as gemm_tile): piece
  // wave % A_PIECES - with fewer pieces than waves (48 / 32 rows) two waves write the same bytes to the same place, which keeps
  // every wave's request stream identical (the counted waits below need no per-wave case)
  const int piece = wave % A_PIECES;
  const char* a_base;
  {
    const int prow = lane >> 3, pch = lane & 7, row = piece * 8 + prow;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    a_base = (const char*)ga.A + (long)gr * ga.lda * Op::EB + ((pch ^ swz_chunk<BKB>(row)) << 4);
  }
  char* a_dst = smem + piece * 1024;
  // weights: column tiles jt = 4 wave + jl
  constexpr long w_tile_stride = 2048, W_KT = 32 * 2048;  // bytes from column tile to column tile, from K tile to K tile
  const char* w_base = ga.Wp + (long)(4 * wave) * w_tile_stride + lane * 16;
  // fragment addressing of the activation stage (as gemm_tile)
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = lr * BKB;

  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int jl = 0; jl < 4; ++jl) acc[mi][jl] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Requests, the same in every interval of every wave: [the 8 fragment loads of the K tile whose MFMAs run in the NEXT interval of
  // this wave (two register sets), then the LDS-DMA piece of the K tile DA ahead].  vmcnt retires in order: the wait at the head of
  // an interval, vmcnt(1), leaves only the previous interval's piece in flight - a piece has two intervals to land, the fragments
  // (L2 hits: all workgroups stream the same panel) one.  Three register sets and a lead of two intervals did not fit 256 registers
  // beside the late waves' loop-carried fragments (104 bytes of scratch, whose reloads the compiler waits for with vmcnt(0)).  Past
  // the end of K the requests are still made - for the LAST K tile again (its piece lands in a stage nobody reads any more, its
  // fragments in a register set nobody uses) - so that every interval is the same straight-line code: with conditional requests
  // hipcc cannot count and puts an s_waitcnt vmcnt(0) in front of every MFMA block.
  int4 wq[2][4][2];
  const int kt_last = KT - 1;
  auto ka_of = [&](int k) { return TAMF_RB_ABL == 4 ? (k & 7) : (k < kt_last ? k : kt_last); };
  auto kw_of = [&](int k) { return TAMF_RB_ABL == 5 ? (k & 3) : (k < kt_last ? k : kt_last); };  // (5: the weight loads re-read the first four K tiles - L2 hits)
  constexpr int WAITN = (TAMF_RB_ABL & 1) && TAMF_RB_ABL != 4 ? 0 : 1;
  if (wave < 4) {
    // ---- early waves.  Prologue: pieces of K tiles 0 .. DA - 2, W(0), piece DA - 1; interval k: fragments of k, [W(k + 1), piece k + DA], MFMAs of k
#pragma unroll
    for (int i = 0; i < DA - 1; ++i) glds16<0>(a_base + (long)ka_of(i) * BKB, a_dst + (i % NSTG) * STAGE);
    rowblock_request(a_dst + ((DA - 1) % NSTG) * STAGE, a_base + (long)ka_of(DA - 1) * BKB, w_base, wq[0], w_tile_stride);
    __builtin_amdgcn_sched_barrier(0);
#define TAMF_RB_STEP(U, kt_)                                                                                                    \
  {                                                                                                                             \
    const int k_ = (kt_);                                                                                                       \
    clip_wait_vm<WAITN>();                                                                                                      \
    clip_barrier_lds();                                                                                                         \
    rowblock_ktile_early<Op, MI>(smem + (k_ % NSTG) * STAGE, a_dst + ((k_ + DA) % NSTG) * STAGE, a_base + (long)ka_of(k_ + DA) * BKB, \
                                 w_base + (long)kw_of(k_ + 1) * W_KT, a_frag, c0, c1, wq[U], wq[1 - (U)], acc, w_tile_stride);   \
  }
    for (int kt = 0; kt < KT; kt += 2) {
      TAMF_RB_STEP(0, kt)
      TAMF_RB_STEP(1, kt + 1)
    }
#undef TAMF_RB_STEP
  } else {
    // ---- late waves: the MFMAs of K tile k - 1 run in interval k.  Prologue: pieces of K tiles 0 .. DA - 1; interval 0: [W(0), piece DA],
    // fragments of K tile 0; interval k >= 1: [W(k), piece k + DA], MFMAs of K tile k - 1, fragments of K tile k; after the last
    // barrier: MFMAs of K tile KT - 1
#pragma unroll
    for (int i = 0; i < DA; ++i) glds16<0>(a_base + (long)ka_of(i) * BKB, a_dst + (i % NSTG) * STAGE);
    int4 af[MI][2];
    clip_wait_vm<0>();
    clip_barrier_lds();
    rowblock_request(a_dst + (DA % NSTG) * STAGE, a_base + (long)ka_of(DA) * BKB, w_base, wq[0], w_tile_stride);
    __builtin_amdgcn_sched_barrier(0);
    rowblock_read_a<MI>(smem, a_frag, c0, c1, af);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(2);
    // interval k: MFMAs with wq[(k - 1) % 2], requests into wq[k % 2]
#define TAMF_RB_STEP(U, kt_)                                                                                                    \
  {                                                                                                                             \
    const int k_ = (kt_);                                                                                                       \
    clip_wait_vm<WAITN>();                                                                                                      \
    clip_barrier_lds();                                                                                                         \
    rowblock_ktile_late<Op, MI>(smem + (k_ % NSTG) * STAGE, a_dst + ((k_ + DA) % NSTG) * STAGE, a_base + (long)ka_of(k_ + DA) * BKB, \
                                w_base + (long)kw_of(k_) * W_KT, a_frag, c0, c1, wq[U], wq[1 - (U)], af, acc, w_tile_stride);    \
  }
    TAMF_RB_STEP(0, 1)
    for (int kt = 2; kt < KT; kt += 2) {
      TAMF_RB_STEP(1, kt)
      TAMF_RB_STEP(0, kt + 1)
    }
#undef TAMF_RB_STEP
    clip_wait_vm<0>();  // W(KT - 1) (and the last, unused piece)
    rowblock_mma_all<Op, MI, 0>(acc, wq[1], af);  // K tile KT - 1 (KT even: register set 1)
    __builtin_amdgcn_s_setprio(0);
  }
  clip_wait_vm<0>();  // (the dummy requests of the last intervals)

  // ---- epilogue: bias + residual, LayerNorm over the 512 columns, fp32 state + operand ----
  // lane (lr, g) holds, per row tile mi (row m0 + 16 mi + lr) and column tile jl, the 4 columns colb(jl) .. + 3
  constexpr int CH = C::CH8 ? 8 : 4;
  auto colb = [&](int jl) { return 64 * wave + (C::CH8 ? 32 * (jl >> 1) + 8 * g + 4 * (jl & 1) : 16 * jl + 4 * g); };
  float bi[4][4], gam[4][4], bet[4][4];
#pragma unroll
  for (int jl = 0; jl < 4; ++jl) {
    g_loadn<4>(epi.bias + colb(jl), bi[jl]);
    g_loadn<4>(epi.gamma + colb(jl), gam[jl]);
    g_loadn<4>(epi.beta + colb(jl), bet[jl]);
  }
  float v[MI][4][4];
  {
    float rs[MI][4][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int gr = m0 + mi * 16 + lr;
      gr = gr < M ? gr : M - 1;
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) g_loadn<4>(epi.resid + (long)gr * N + colb(jl), rs[mi][jl]);
    }
    const float ws = epi.ctl.wscale;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int jl = 0; jl < 4; ++jl)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[mi][jl][r] = fmaf(acc[mi][jl][r], ws, bi[jl][r]) + rs[mi][jl][r];
  }
  float mean[MI], rstd[MI];
  {
#pragma clang fp contract(off)
    // pass 1: row sums.  red[0][row][4 c + nq] = b[nq] of column tile c (wave w: c = w / 2, nq = 2 (w % 2) + nql)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nql = 0; nql < 2; ++nql) {
        const float* e = v[mi][2 * nql];
        const float* o = v[mi][2 * nql + 1];
        const float qe = ((e[0] + e[1]) + e[2]) + e[3], qo = ((o[0] + o[1]) + o[2]) + o[3];
        const float b = rowblock_block_sum<Op>(qe, qo);
        if (g == 0) red[(mi * 16 + lr) * 16 + 2 * wave + nql] = b;
      }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float4* p = (const float4*)(red + (mi * 16 + lr) * 16);
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 b4 = p[c];
        t[c] = ((b4.x + b4.y) + b4.z) + b4.w;
      }
      mean[mi] = (((t[0] + t[1]) + t[2]) + t[3]) * (1.0f / N);
    }
    // pass 2: sums of squared deviations (second half of `red`: no barrier needed against the reads above)
    float* red2 = red + BM * 16;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nql = 0; nql < 2; ++nql) {
        float dq[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dlt = v[mi][2 * nql + h][r] - mean[mi];
            dq[h][r] = dlt * dlt;
          }
        const float qe = ((dq[0][0] + dq[0][1]) + dq[0][2]) + dq[0][3], qo = ((dq[1][0] + dq[1][1]) + dq[1][2]) + dq[1][3];
        const float b = rowblock_block_sum<Op>(qe, qo);
        if (g == 0) red2[(mi * 16 + lr) * 16 + 2 * wave + nql] = b;
      }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float4* p = (const float4*)(red2 + (mi * 16 + lr) * 16);
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 b4 = p[c];
        t[c] = ((b4.x + b4.y) + b4.z) + b4.w;
      }
      const float var = (((t[0] + t[1]) + t[2]) + t[3]) * (1.0f / N);
      rstd[mi] = 1.0f / sqrtf(var + epi.eps);
    }
  }
  float am = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int gr = m0 + mi * 16 + lr;
#pragma unroll
    for (int jl = 0; jl < 4; ++jl)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[mi][jl][r] = fmaf((v[mi][jl][r] - mean[mi]) * rstd[mi], gam[jl][r], bet[jl][r]);  // (explicit: the same in every LayerNorm)
    if (gr < M) {
      if constexpr (C::CH8) {
#pragma unroll
        for (int nql = 0; nql < 2; ++nql) {
          float y[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            y[r] = v[mi][2 * nql][r];
            y[4 + r] = v[mi][2 * nql + 1][r];
          }
          const long o = (long)gr * N + colb(2 * nql);
          gst16f(epi.xout + o, y[0], y[1], y[2], y[3]);
          gst16f(epi.xout + o + 4, y[4], y[5], y[6], y[7]);
          if (epi.xop) Op::template store_rc<8>(epi.xop, o, y, am);
        }
      } else {
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
          const long o = (long)gr * N + colb(jl);
          gst16f(epi.xout + o, v[mi][jl][0], v[mi][jl][1], v[mi][jl][2], v[mi][jl][3]);
          if (epi.xop) Op::template store_rc<4>(epi.xop, o, v[mi][jl], am);
        }
      }
    }
  }
  (void)CH;
  Op::range_flag(am, epi.ctl.status);
}
