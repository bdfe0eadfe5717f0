// Clip-aligned GEMM of the encoder layers (FFN1: 256-column tiles; FFN2, and in f32 out-proj and the QKV projection: 128-column
// tiles):
//
//   C[b*Sp + r][n] = sum_k A[b*Sp + r][k] * W[n][k]          one M tile = the Sp token rows of ONE clip
//
// Why its own kernel.  The 128 x 128 tiles of tamf_gemm.h stage (128 + 128) * 128 bytes from L2 per 128-byte K tile;
// measured on MI355X (tools/kbench.py, ablation bits: loads alone take as long as the MFMAs alone), their K loop runs into
// the L2 -> LDS path of a CU (about 70 GB/s per CU, MI355X_MICROARCH.md "Indexed rows: gather into LDS") as much as into the
// matrix pipe.  And M = 64 clips * 208 rows = 13 * 2^10 rows never fills 256 CUs evenly with power-of-two row tiles (every
// launch ends in a 3.25th round).  A tile of one whole clip (208 rows = 13 MFMA row tiles) by 256 / 128 columns stages
// (208 + BN) * 128 bytes per K tile for 208 * BN outputs - 1.8x / 1.25x the flops per staged byte - and B = 64 clips give
// 512 / 256 tiles: exact rounds of the 256 CUs.
//
// Workgroup = 8 waves (one workgroup per CU, two waves per SIMD), wave grid 2 (M) x 4 (N): waves 0-3 ("X") own the first
// XSUB row tiles, waves 4-7 ("Y") the other 13 - XSUB (waves w and w + 4 share a SIMD, so every SIMD carries all 13), each
// over BN / 4 columns.  Staging, swizzle, fragment addressing and the MFMA operand traits are those of tamf_gemm.h (LDS-DMA
// pieces of 8 rows x 128 bytes, source-side XOR swizzle, one barrier per K tile).
//
// The two waves of a SIMD run half a K tile apart.  Right after a barrier every wave would wait for its first fragments
// (LDS latency) and then issue its share of the next K tile's LDS-DMA pieces (the CU's address unit takes 16 cycles per
// piece and a wave is blocked while its piece queues) - with all eight waves in those phases together the matrix pipe idles
// for a third of each K tile.  So the X waves work as above - fragments of K tile i, all DMA pieces of a later K tile, MFMAs
// of K tile i - while their SIMD partners, the Y waves, spend the head of the interval on the MFMAs of K tile i - 1, whose
// fragments they read into registers at the end of the previous interval, and its tail on reading the fragments of K tile i.
// The matrix pipe of every SIMD is fed by Y while X waits and issues, and by X while Y reads.  Y issues its MFMAs at static
// priority and X, which carries the DMA issue, owns fewer of the 13 row tiles (6 at 256 columns, 2 at 128: tamf_hip.hip).
//
// Round-2 additions, each from a measurement on MI355X (tools/clip_timeline*.py: per-wave shader-clock stamps, standalone and
// inside the hipGraph step; tools/kbench.py ablation bits; tools/ab_*.sh: two builds alternating on one box):
//  * Inside the step the 128-column K loop waited for its LDS-DMA (interval 2 172 cycles standalone, 2 804 in situ: the A panel
//    of FFN2 comes from the Infinity Cache / HBM, the loader waves sat 1 020 cycles at the barrier).  Its stage is 42 KB, so there
//    are THREE stages and a K tile is requested two intervals ahead; the loaders wait with a counted vmcnt (everything but the
//    newest requests) instead of vmcnt(0).  FFN2 in situ 85 -> 73 us.
//  * A workgroup's tiles are one stream of K tiles: no drain between tiles, the next tile's requests go out before the
//    epilogue of the finished one.
//  * The epilogue goes from the accumulators to HBM: W rows are staged in a permuted order (clip_wperm) so that a lane owns
//    16-byte pieces of its output row.  No LDS round trip of the fp32 tile, no barrier; X stores its rows at the head of the
//    next tile's first interval while Y multiplies the last K tile, Y behind those MFMAs while X is back in the K loop; the
//    loaders' boundary wait leaves their own stores in flight (a store is acknowledged when the L2 has taken it, and all CUs
//    store at the same moment).
//  * Epilogues with an activation run their row tiles in a LOOP (lane-private LDS slot): unrolled, GELU is 11 - 13 KB of
//    straight-line code per wave role that every launch fetched cold - +3 us per FFN1 launch inside the step, invisible in a
//    standalone benchmark loop.
//  * The hi / lo split of an operand store is compiled with fp contraction off: fused with the multiply that produced the
//    value it stored different lo words in different kernel variants (batch-invariance tests).
// What did NOT pay (measured, reverted): the pieces of a K tile spread between the loaders' MFMAs (FFN2 72 -> 85 us), Y's
// fragment reads fused under its MFMAs (69 -> 76 us, or spills), erf from an LDS table instead of v_exp / v_rcp (the gather's
// bank conflicts cost more than the transcendentals), 128-column tiles for FFN1 (86 against 83 us), QKV on clip tiles in the
// 16-bit modes (two launches - Q | K on 256-column tiles, V transposed on 128-column ones: 62.5 against 62 us; used in f32, where it
// wins), LayerNorm inside the FFN2 kernel.
#pragma once
#include "tamf_gemm.h"

#ifdef TAMF_TIMELINE  // debug build (tools/clip_timeline.py): shader-clock stamps of waves 0 (X) and 4 (Y) of every workgroup
__device__ unsigned long long g_clip_ts[512 * 2 * 8 * 4];  // [workgroup][X|Y][interval 4..11][4 stamps]
#ifndef TAMF_TIMELINE_NI  // (-DTAMF_TIMELINE_NI=2: only the 128-column launches stamp - for runs of the whole step)
#define TAMF_TIMELINE_NI NI
#endif
#define TAMF_CLIP_TS(slot)                                                                     \
  if (dbg_on && NI == TAMF_TIMELINE_NI) {                                                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                 \
    if (lane_dbg == 0) g_clip_ts[((blockIdx.x * 2 + mh_dbg) * 8 + (it_dbg - 4)) * 4 + (slot)] = t_; \
  }
#else
#define TAMF_CLIP_TS(slot)
#endif

template <class Op>
struct ClipGemmArgs {
  const typename Op::elem_t* A;
  int lda;  // elements
  const typename Op::elem_t* W;
  int ldw;
  int n_clips, Sp;  // A has n_clips * Sp rows; a tile covers rows [b*Sp, b*Sp + Sp)
  int N, K;
  int n_tiles;      // (row parts of all clips) * (N / BN)
  // > 0: every clip is cut into TWO row parts, rows [0, split_rows) and [split_rows, Sp), each a tile of its own (the kernel is
  // then instantiated with NSUB = split_rows / 16 row tiles; the second part's rows past the clip are clamped on the load side and
  // skipped on the store side like a clip's own padding).  Used when whole-clip tiles would fill at most half of the CUs
  // (32 clips per GPU): same products in the same K order per output element, i.e. the same bits as the whole-clip tiles.
  int split_rows;
  // > 0 (= N / BN, the column tiles of a clip): COLUMN-SPLIT ROUNDS - the launch is R = n_tiles / grid exact rounds and round r takes
  // column tiles [r ntn / R, (r + 1) ntn / R) of EVERY clip, instead of all column tiles of the r-th R-th of the clips.  An XCD's chunk of
  // a round is then (R x as many clips) x (ntn / R column tiles): the weight rows it streams per round shrink by R and can stay in its
  // 4-MB L2 (FFN1, split modes: 2 MB instead of the whole 4-MB matrix), the A panels it fetches double.  Same tiles, same bits.
  int colsplit;
  int abl;          // kernel-benchmark ablations (-DTAMF_BENCH builds only, TAMF_ABL) (tools/kbench.py): 1 = no loads after the first K tiles, 2 = no MFMAs, 4 = no epilogue,
                    // 8 = no activation, 16 = every row tile is stored into the rows of the first one (no new lines to write back)
};

template <int NSUB, int NI, int XSUB, int CH>
struct ClipCfg {
  static constexpr int MT = NSUB * 16;        // tile rows (>= Sp)
  static constexpr int BN = 64 * NI;          // tile columns: 4 waves x NI MFMA column tiles
  // row tiles of the X waves (rows [0, 16 XSUB)) and of the Y waves (the rest); MSUB0 = the larger count (accumulator array)
  static constexpr int MSUBX = XSUB, MSUBY = NSUB - XSUB;
  static constexpr int MSUB0 = MSUBX > MSUBY ? MSUBX : MSUBY;
  static constexpr int ROWS = MT + BN;        // rows of one staged K tile: A rows then W rows
  static constexpr int STAGE = ROWS * GEMM_BKB;
  static constexpr int NPIECE = ROWS / 8, A_PIECES = MT / 8;
  static constexpr int CHUNK = CH;             // consecutive output columns of a lane per store (clip_wperm)
  static constexpr int NCHUNK = 4 * NI / CH;  // such chunks per lane and row tile, CH * 4 columns apart
  static_assert(NI % 2 == 0, "column tiles per wave");
  // stages of the K-tile stream (the epilogue goes from registers to HBM and needs no LDS): three where they fit - the
  // 128-column tiles - i.e. the LDS-DMA of a K tile is issued TWO intervals before its first read
  static constexpr int NSTAGE = (3 * STAGE <= 160 * 1024) ? 3 : 2;
  // + a lane-private scratch row tile per wave (16 bytes x NI per lane) for the rolled form of an epilogue with an activation
  static constexpr int SCRATCH_OFF = NSTAGE * STAGE, SCRATCH_WAVE = NI * 1024;
  // + the row terms of the tile's rows for the epilogues that normalise (deferred LayerNorm, tamf_device.h): three slots, tile of round
  // r in slot r % 3 - the X waves stage tile r + 1 when they have stored tile r - 1, while the Y waves may still be storing tile r - 1
  // themselves (slot (r - 1) % 3) and tile r's slot is about to be read by both
  static constexpr int STATS_OFF = SCRATCH_OFF + 8 * SCRATCH_WAVE;
  static constexpr int BYTES = STATS_OFF + 3 * MT * 8;
  static_assert(BYTES <= 160 * 1024, "LDS budget");
  // LDS-DMA pieces per K tile of loader wave nq: PIECES_HI for nq < PIECES_REM, else PIECES_HI - 1
  static constexpr int PIECES_HI = (NPIECE + 3) / 4, PIECES_REM = NPIECE % 4 == 0 ? 4 : NPIECE % 4;
};

// Column order of a tile.  The MFMAs take W as their ROW operand, so lane (lr, g) of a wave ends up with row lr of a row tile
// and, per column tile ni, the outputs of the four staged W rows 16 ni + 4 g + {0..3} of the wave's column block: four
// consecutive columns.  That is what a 4-byte output wants - one 16-byte store per lane and column tile, the four lane groups
// of a row filling 64 contiguous bytes per instruction (CH = 4: no permutation).  A 16-bit output plane wants EIGHT consecutive
// columns per lane and store: W row wperm(s) is staged at row s - a permutation of bits 2..4 inside every 32 rows, applied
// on the SOURCE side of the LDS-DMA - so that column tiles 2c and 2c + 1 together hold columns 32 c + 8 g + {0..7} (CH = 8).
// Either way a lane stores whole 16-byte pieces of its output row straight from the accumulators, every store instruction
// writes 64-byte runs, and the fp32 tile never passes through LDS.
template <int CH>
TAMF_DEV int clip_wperm(int s) {
  static_assert(CH == 4 || CH == 8, "columns per lane and store");
  if constexpr (CH == 8) {
    return (s & ~28) | ((s >> 2) & 4) | ((s << 1) & 24);  // staged bits [3:2] (g) -> [4:3], [4] (ni & 1) -> [2]
  } else {
    return s;
  }
}
// Source addressing of the LDS-DMA pieces of one tile.  Piece q covers staged rows [8q, 8q + 8) (lane: row 8q + lane / 8,
// 16-byte chunk lane % 8, XOR-swizzled by the STAGED row on the source side); rows < MT are the clip's A rows (clamped to the
// clip), the others the W rows of the column tile in clip_wperm order.  Loader wave nq takes pieces nq, nq + 4, nq + 8, ...:
// their staged rows are 32 apart, so the swizzle term is the same for all of them; the staged W row of piece nq + 4 i is
// el + lane / 8 + 32 (i + hq) with el < 32, and clip_wperm, a bit permutation, splits into a per-lane term and a per-piece
// (scalar) term.
struct ClipSrc {
  unsigned a0, a_last, w0;  // byte offsets from A / W of the lane's row in piece nq (A: unclamped; clamped last row; W: permuted)
  int hq;                   // 32-row block of W piece nq + 4 i: i + hq
  int rows;                 // valid rows of the tile's row part
};
// row part `v` of the launch: first row (in rows of A) and number of valid rows
template <class Op>
TAMF_DEV void clip_part(const ClipGemmArgs<Op>& ga, int v, int& base, int& rows) {
  if (ga.split_rows > 0) {
    base = (v >> 1) * ga.Sp + (v & 1) * ga.split_rows;
    rows = (v & 1) ? ga.Sp - ga.split_rows : ga.split_rows;
  } else {
    base = v * ga.Sp;
    rows = ga.Sp;
  }
}
template <class Op, class C>
TAMF_DEV ClipSrc clip_src(const ClipGemmArgs<Op>& ga, int b, int n0, int nq, int prow, int pch) {
  const int r0 = nq * 8 + prow;
  const unsigned swz = (unsigned)((pch ^ swz_chunk<GEMM_BKB>(r0)) << 4);
  const unsigned ldaB = (unsigned)(ga.lda * Op::EB), ldwB = (unsigned)(ga.ldw * Op::EB);
  const int e = nq * 8 - C::MT;  // staged W row of (virtual) piece nq, lane row 0: negative
  ClipSrc s;
  int base;
  clip_part(ga, b, base, s.rows);
  s.a0 = (unsigned)(base + r0) * ldaB + swz;
  s.a_last = (unsigned)(base + s.rows - 1) * ldaB + swz;
  s.w0 = (unsigned)(n0 + clip_wperm<C::CHUNK>((e & 31) + prow)) * ldwB + swz;
  s.hq = e >> 5;
  return s;
}
// issue pieces nq + 4 i (i = 0 .. ) of K tile kt into the stage at `stage_base`.
// Addressing: everything that changes from piece to piece and from K tile to K tile is wave-uniform - it goes into the SCALAR base of
// the request (s_add / s_addc), the per-lane part (row inside the piece, swizzled chunk) is one 32-bit VGPR offset that is fixed for
// the tile (global_load_lds_dwordx4 v_off, s[base:base+1]).  Only the last four A pieces of a tile can hold rows past the clip
// (a tile has fewer than 32 padding rows: shape_ok) and keep the per-lane clamp.  Why it matters: a vector instruction of ANY wave
// cannot issue while another wave of the SIMD streams MFMAs (tools/micro/issue_overlap.hip: 256 VALU adds beside an f32 MFMA stream
// take as long as the stream; only SALU instructions pass), so the 5 - 6 address instructions per piece that the compiler used to spend
// were not hidden by the Y waves' MFMAs - they were time in which the X waves' own MFMAs had not started.
template <int AUX>
TAMF_DEV void glds16_sv(const char* sbase /* uniform */, unsigned voff, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((tamf_gbl_void*)(sbase + (size_t)voff), (tamf_lds_void*)lds_wave_base, 16, 0, AUX);
}
template <class Op, class C>
TAMF_DEV void clip_issue(const ClipGemmArgs<Op>& ga, const ClipSrc& s, int nq, int prow, int kt, char* stage_base, int i0 = 0, int i1 = 1 << 20) {
  constexpr int QS = 4;
  constexpr int NI_ = (C::NPIECE + QS - 1) / QS;  // requests of a loader wave per K tile; [i0, i1): the ones to issue now (constants after unrolling)
  const unsigned ldaB = (unsigned)(ga.lda * Op::EB), ldwB = (unsigned)(ga.ldw * Op::EB);
  const char* Ak = (const char*)ga.A + (size_t)kt * GEMM_BKB;
  const char* Wk = (const char*)ga.W + (size_t)kt * GEMM_BKB;
#pragma unroll
  for (int i = 0; i < NI_; ++i) {
    const int q = nq + QS * i;
    if (i < i0 || i >= i1) continue;
    if ((i + 1) * QS <= C::NPIECE || q < C::NPIECE) {
      if (QS * i + QS - 1 < C::A_PIECES - 4) {  // (compile time: pieces whose rows are inside every clip these tiles are used for)
        glds16_sv<0>(Ak + (size_t)((unsigned)(QS * 8 * i) * ldaB), s.a0, stage_base + q * 1024);
      } else if (q < C::A_PIECES) {
        const unsigned o = (q * 8 + prow < s.rows) ? s.a0 + (unsigned)(QS * 8 * i) * ldaB : s.a_last;
        glds16_sv<0>(Ak, o, stage_base + q * 1024);
      } else {
        glds16_sv<0>(Wk + (size_t)((unsigned)clip_wperm<C::CHUNK>(32 * (i + s.hq)) * ldwB), s.w0, stage_base + q * 1024);
      }
    }
  }
}

// Wave priorities (f32).  While a wave streams MFMAs no other wave of its SIMD issues a vector instruction, whatever the priorities
// (tools/micro/issue_overlap.hip) - but priorities decide who goes first when both are ready.  The Y waves multiply at static priority 2;
// in f32 the X waves take priority 3 for everything that is NOT their MFMA stream (fragment reads, requests, epilogue) and 0 for the
// MFMAs: their requests are out - and their epilogue stores on the way - before the long fp32 MFMA phases of the pair begin
// (6.50 -> 6.31 ms per step at B = 64; the 16-bit modes lose 0.2 - 0.8 % with it and keep priority 0).  -DTAMF_CLIP_XPRIO=0: off (A/B)
#ifndef TAMF_CLIP_XPRIO
#define TAMF_CLIP_XPRIO 1
#endif
#ifndef TAMF_CLIP_YFUSE  // Y waves read the next K tile's fragments inside their MFMA stream (clip_mma_read_y): bit 0 = f32, bit 1 = the 16-bit modes (A/B)
#define TAMF_CLIP_YFUSE 1  // (f32 6.30 -> 6.235 ms per step; the 16-bit modes 0.2 - 0.8 % slower with it: profiles/r04/yfuse_c36.txt)
#endif
#ifndef TAMF_CLIP_SPREAD  // f32: LDS-DMA requests between the X waves' own MFMAs (clip_ktile_x; 0 = one batch ahead of them, A/B)
#define TAMF_CLIP_SPREAD 2  // (requests behind every 8 MFMAs)
#endif
#if TAMF_CLIP_XPRIO
#define TAMF_CLIP_XPRIO_HI if constexpr (Op::PREC == 0) __builtin_amdgcn_s_setprio(3);
#define TAMF_CLIP_XPRIO_LO if constexpr (Op::PREC == 0) __builtin_amdgcn_s_setprio(0);
#else
#define TAMF_CLIP_XPRIO_HI
#define TAMF_CLIP_XPRIO_LO
#endif

// X waves, one K tile: the fragments of the first row tiles are requested, then ALL pieces of the next K tile go out into
// `nxt`, then the MFMAs run with the A fragments streamed two row tiles ahead.  `cur` and `nxt` are the two LDS stages and
// never overlap; the __restrict__ qualifiers of this (inlined) helper are what keeps hipcc from placing an s_waitcnt vmcnt(0)
// in front of the LDS reads that follow the LDS-DMA in program order (as in tamf_attn.h).
template <class Op, class C, int NI, bool TR>
TAMF_DEV void clip_ktile_x(const char* __restrict__ cur, char* __restrict__ nxt, bool load_next, bool compute,
                           const ClipGemmArgs<Op>& ga, const ClipSrc& src4, int nq, int prow, int kt_next, int a_frag, int w_frag,
                           int c0, int c1, f32x4 (&acc)[C::MSUB0][NI], bool dbg_on = false, int it_dbg = 0, int lane_dbg = 0) {
  constexpr int BKB = GEMM_BKB;
  constexpr int mh_dbg = 0;
  (void)dbg_on; (void)it_dbg; (void)lane_dbg; (void)mh_dbg;
  TAMF_CLIP_XPRIO_HI
  int4 wf[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    wf[ni][0] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c0);
    wf[ni][1] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c1);
  }
  int4 af[C::MSUBX][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    af[mi][0] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c0);
    af[mi][1] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c1);
  }
  // f32: the requests go out BETWEEN this wave's own MFMAs, behind its first row tiles.  Beside the Y wave's fp32 MFMA stream no
  // request issues at all (tools/micro/issue_overlap.hip), so as one batch they ran after it - ~900 cycles per interval in which the
  // address unit took its 16 cycles per piece and no SIMD of the CU multiplied; inside the wave's own stream a request costs an issue
  // slot and the address unit works in the shadow of the 32-cycle MFMAs.  (Early row tiles: with two stages the data is due at the
  // next barrier.)  The 16-bit modes keep the batch: their Y phase is as long as the address unit needs, and it passes requests.
  constexpr bool SPREAD = TAMF_CLIP_SPREAD && Op::PREC == 0;
  constexpr int NREQ = (C::NPIECE + 3) / 4, GROUPS = C::MSUBX * NI;  // a group = the 8 MFMAs of one (row tile, column tile) product
  constexpr int PER = (NREQ + GROUPS - 1) / GROUPS > TAMF_CLIP_SPREAD ? (NREQ + GROUPS - 1) / GROUPS : TAMF_CLIP_SPREAD;
  if (load_next && !SPREAD) clip_issue<Op, C>(ga, src4, nq, prow, kt_next, nxt);
  TAMF_CLIP_XPRIO_LO
  TAMF_CLIP_TS(1)
#pragma unroll
  for (int mi = 0; mi < C::MSUBX; ++mi) {
    if (mi + 2 < C::MSUBX) {
      af[mi + 2][0] = *(const int4*)(cur + a_frag + (mi + 2) * 16 * BKB + c0);
      af[mi + 2][1] = *(const int4*)(cur + a_frag + (mi + 2) * 16 * BKB + c1);
    }
    if (compute) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (TR) Op::mma_t(acc[mi][ni], af[mi], wf[ni]);  // D rows = m (4g + reg), cols = n (lr)
        else Op::mma(acc[mi][ni], wf[ni], af[mi]);                 // D rows = n (4g + reg), cols = m (lr)
        if constexpr (SPREAD) {
          if ((mi * NI + ni) * PER < NREQ) {
            __builtin_amdgcn_sched_barrier(0);
            if (load_next) clip_issue<Op, C>(ga, src4, nq, prow, kt_next, nxt, (mi * NI + ni) * PER, (mi * NI + ni + 1) * PER);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
}
// Y waves: all fragments of a K tile into registers / the MFMAs on fragments read one interval earlier
template <class C, int NI>
TAMF_DEV void clip_read_y(const char* cur, int a_frag, int w_frag, int c0, int c1, int4 (&wf)[NI][2], int4 (&af)[C::MSUBY][2]) {
  constexpr int BKB = GEMM_BKB;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    wf[ni][0] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c0);
    wf[ni][1] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c1);
  }
#pragma unroll
  for (int mi = 0; mi < C::MSUBY; ++mi) {
    af[mi][0] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c0);
    af[mi][1] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c1);
  }
}
// Y waves, fused form: the MFMAs of K tile j - 1 row tile by row tile, and behind the MFMAs of a row tile the A fragments of K tile j
// for that row tile, read into the registers the MFMAs have just consumed (no second fragment set); the W fragments follow behind
// the last row tile.  An LDS read inside the wave's own MFMA stream costs a few cycles (MI355X_MICROARCH.md, LDS: "issued between
// MFMAs"), while behind the stream the reads of the four Y waves compete with the X waves' reads for the LDS array and for issue slots.
template <class Op, class C, int NI, bool TR>
TAMF_DEV void clip_mma_read_y(const char* nxt, bool more, int a_frag, int w_frag, int c0, int c1, int4 (&wf)[NI][2],
                              int4 (&af)[C::MSUBY][2], f32x4 (&acc)[C::MSUB0][NI]) {
  constexpr int BKB = GEMM_BKB;
#pragma unroll
  for (int mi = 0; mi < C::MSUBY; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if constexpr (TR) Op::mma_t(acc[mi][ni], af[mi], wf[ni]);
      else Op::mma(acc[mi][ni], wf[ni], af[mi]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      af[mi][0] = *(const int4*)(nxt + a_frag + mi * 16 * BKB + c0);
      af[mi][1] = *(const int4*)(nxt + a_frag + mi * 16 * BKB + c1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (more) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      wf[ni][0] = *(const int4*)(nxt + w_frag + ni * 16 * BKB + c0);
      wf[ni][1] = *(const int4*)(nxt + w_frag + ni * 16 * BKB + c1);
    }
  }
}
template <class Op, class C, int NI, bool TR>
TAMF_DEV void clip_mma_y(const int4 (&wf)[NI][2], const int4 (&af)[C::MSUBY][2], f32x4 (&acc)[C::MSUB0][NI]) {
#pragma unroll
  for (int mi = 0; mi < C::MSUBY; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if constexpr (TR) Op::mma_t(acc[mi][ni], af[mi], wf[ni]);
      else Op::mma(acc[mi][ni], wf[ni], af[mi]);
    }
}

// tile of this workgroup in round `r` of the persistent grid (-1: none): inside a round the XCDs own contiguous chunks of
// the tile list (xcd_remap), i.e. a few whole clips x all their column tiles - a clip's A panel is fetched into one L2
template <class Op>
TAMF_DEV int clip_tile_of(const ClipGemmArgs<Op>& ga, int round) {
  const int n_tiles = ga.n_tiles;
  const int G = gridDim.x, base = round * G;
  if (base >= n_tiles) return -1;
  const int cnt = n_tiles - base < G ? n_tiles - base : G;
  if ((int)blockIdx.x >= cnt) return -1;
  const int i = xcd_remap(blockIdx.x, cnt);
  if (ga.colsplit > 0) {  // (n_tiles = R G exactly, ntn = R per: ClipLaunch::launch)
    const int per = ga.colsplit / (n_tiles / G);
    return (i / per) * ga.colsplit + round * per + i % per;
  }
  return base + i;
}

// Register epilogue of one wave: row tile mi of the wave -> row row0 + 16 mi of the clip; chunk c of the lane = columns
// gn + 4 CH c .. + CH (column tile c for CH = 4, column tiles 2c and 2c + 1 for CH = 8)
// (ci / rs: second column constants and the staged (mean, rstd) of the tile's rows - epilogues with Epi::ROWSTATS, deferred LayerNorm)
template <class C, int NI, int MS, int ACT, class Epi>
TAMF_DEV void clip_store_rows_act(const Epi& epi, const f32x4 (&acc)[C::MSUB0][NI], int row0, int Sp, int m0, int gn,
                                  const float (&bi)[C::NCHUNK][C::CHUNK], const float (&ci)[C::NCHUNK][C::CHUNK], const float2* rs,
                                  float& am) {
  constexpr int CH = C::CHUNK;
  if constexpr (Epi::ROWSTATS && Epi::PREFETCH) {
    // the epilogue reads a row piece of its own output buffer per row tile (EpiResid: the residual stream, in place).  Requested inside
    // the row loop, each piece would wait behind the previous row tile's stores (vmcnt retires in order): one store round trip per
    // row tile.  So row tile mi + 1's pieces are requested BEFORE row tile mi is stored - the wait for them then leaves those stores
    // in flight (the compiler counts them: straight-line code).
    static_assert(C::NCHUNK == 1, "one row piece per lane and row tile");
    float u[2][CH];
    {
      const int r = row0 < Sp ? row0 : Sp - 1;
      epi.template prefetch<CH>(m0 + r, gn, u[0]);
    }
#pragma unroll
    for (int mi = 0; mi < MS; ++mi) {
      const int r = row0 + mi * 16;
      if (mi + 1 < MS) {
        const int rn = r + 16 < Sp ? r + 16 : Sp - 1;
        epi.template prefetch<CH>(m0 + rn, gn, u[(mi + 1) & 1]);
      }
      if (r < Sp) {
        float v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = acc[mi][j / 4][j % 4];
        epi.template finish_pf<CH>(m0 + r, gn, v, u[mi & 1], bi[0], ci[0], rs[r], am);
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MS; ++mi) {
    const int r = row0 + mi * 16;
    if (r < Sp) {
#pragma unroll
      for (int c = 0; c < C::NCHUNK; ++c) {
        float v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = acc[mi][c * (CH / 4) + j / 4][j % 4];
        if constexpr (Epi::ROWSTATS) epi.template finish_ln<CH>(ACT, m0 + r, gn + 4 * CH * c, v, bi[c], ci[c], rs[r], am);
        else epi.template finish_act<CH>(ACT, m0 + r, gn + 4 * CH * c, v, bi[c], am);
      }
    }
  }
}
// The same with the row tiles in a LOOP, for the epilogues that carry an activation: unrolled, the GELU of 6 / 7 row tiles is
// 11 - 13 KB of straight-line code per wave role that every launch fetches once, cold (in the step the kernels alternate; measured
// in situ: FFN1 82.6 us unrolled against 79.5 us with the looped LDS-walking epilogue it replaced).  A loop cannot index the
// accumulator registers, so the row tile of the iteration takes a round trip through a lane-private LDS slot (4 NI floats per
// lane: NI ds_write_b128 + NI ds_read_b128, conflict-free, no barrier), selected by a scalar branch.
template <class C, int NI, int MS, int ACT, class Epi>
TAMF_DEV void clip_store_rows_rolled(const Epi& epi, const f32x4 (&acc)[C::MSUB0][NI], int row0, int Sp, int m0, int gn,
                                     const float (&bi)[C::NCHUNK][C::CHUNK], const float (&ci)[C::NCHUNK][C::CHUNK], const float2* rs,
                                     char* slot /* wave scratch + 16 * lane */, float& am, bool one_row = false) {
  constexpr int CH = C::CHUNK;
#pragma clang loop unroll(disable)
  for (int mi = 0; mi < MS; ++mi) {
#pragma unroll
    for (int k = 0; k < MS; ++k)
      if (mi == k) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4*)(slot + ni * 1024) = acc[k][ni];
      }
    f32x4 a[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) a[ni] = *(const f32x4*)(slot + ni * 1024);
    const int r = row0 + mi * 16;
    if (r < Sp) {
#pragma unroll
      for (int c = 0; c < C::NCHUNK; ++c) {
        float v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = a[c * (CH / 4) + j / 4][j % 4];
        if constexpr (Epi::ROWSTATS) epi.template finish_ln<CH>(ACT, one_row ? m0 : m0 + r, gn + 4 * CH * c, v, bi[c], ci[c], rs[r], am);
        else epi.template finish_act<CH>(ACT, one_row ? m0 : m0 + r, gn + 4 * CH * c, v, bi[c], am);
      }
    }
  }
}
template <class C, int NI, int MS, class Epi>
TAMF_DEV void clip_store_rows(const Epi& epi, const f32x4 (&acc)[C::MSUB0][NI], int row0, int Sp, int m0, int gn,
                              const float (&bi)[C::NCHUNK][C::CHUNK], const float (&ci)[C::NCHUNK][C::CHUNK], const float2* rs, int abl,
                              char* slot) {
  float am = 0.f;  // range accumulator of the operand stores (Op::store_rc), flagged once per tile and wave
  if (abl & 24) {  // (benchmark ablations: 8 = no activation, 16 = every row of the tile is stored into the clip's first row)
    if (epi.act == ACT_GELU && !(abl & 8)) clip_store_rows_rolled<C, NI, MS, ACT_GELU>(epi, acc, row0, Sp, m0, gn, bi, ci, rs, slot, am, (abl & 16) != 0);
    else clip_store_rows_rolled<C, NI, MS, ACT_NONE>(epi, acc, row0, Sp, m0, gn, bi, ci, rs, slot, am, (abl & 16) != 0);
    return;
  }
  if (epi.act == ACT_GELU) {
    clip_store_rows_rolled<C, NI, MS, ACT_GELU>(epi, acc, row0, Sp, m0, gn, bi, ci, rs, slot, am);
  } else if (epi.act == ACT_SILU) {
    clip_store_rows_rolled<C, NI, MS, ACT_SILU>(epi, acc, row0, Sp, m0, gn, bi, ci, rs, slot, am);
  } else {
    clip_store_rows_act<C, NI, MS, ACT_NONE>(epi, acc, row0, Sp, m0, gn, bi, ci, rs, am);
  }
  epi.flag(am);
}

// Register epilogue of the transposed form (EpiVt): lane (lr, g) holds, per row tile and column tile ni, feature
// n0 + wn0 + 16 ni + lr of the four tokens 16 s + 4 g + {0..3} of row tile s.  In the 16-bit modes the keys 4g .. 4g+3 of the
// two 16-key groups of a 32-key block are adjacent in a V^T row (vt_key_pos), so the row tiles 2u and 2u + 1 of a wave give one
// 16-byte piece per plane; f32 keeps the natural key order, one piece per row tile.  Tokens >= Sp are written as zeros
// (they are padding keys: the row tile past the clip, and the clamped rows of its last one).  FIRST = the wave's first row tile
// (even: XSUB is), MS its row tiles.
template <class Op, class C, int NI, int FIRST, int MS, class Epi>
TAMF_DEV void clip_store_vt(const Epi& epi, const f32x4 (&acc)[C::MSUB0][NI], int g, int Sp, int b, int eg0 /* feature of ni = 0 */,
                            const float (&bb)[NI], const float2* rs /* Epi::ROWSTATS: the staged row factors of the clip's tokens */) {
  static_assert(FIRST % 2 == 0, "row-tile pairs must not straddle the X / Y split");
  float am = 0.f;
  // acc ws + bias, or - deferred LayerNorm of the token's row - acc ra[token] + c2: the factors of the lane's four tokens of a row tile
  // come from LDS once per row tile (they are the same for every feature tile ni)
  auto factors = [&](int tok, float (&ra)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (Epi::ROWSTATS) ra[j] = rs[tok + j].x;
      else ra[j] = epi.ctl.wscale;
    }
  };
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(bb[ni]));  // (one wait, ahead of the stores)
  if constexpr (Op::PREC == 0) {
#pragma unroll
    for (int mi = 0; mi < MS; ++mi) {
      const int tok = (FIRST + mi) * 16 + 4 * g;
      float ra[4];
      factors(tok < Sp ? tok : 0, ra);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tok < Sp ? fmaf(acc[mi][ni][j], ra[j], bb[ni]) : 0.f;
        epi.template store_keys<4>(b, eg0 + 16 * ni, tok, v, am);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < (MS + 1) / 2; ++u) {
      const int s0 = FIRST + 2 * u;  // row tiles s0 and s0 + 1: keys 16 s0 + 4g + j and 16 (s0 + 1) + 4g + j
      const int tok0 = s0 * 16 + 4 * g, tok1 = tok0 + 16;
      float ra0[4], ra1[4];
      factors(tok0 < Sp ? tok0 : 0, ra0);
      factors(tok1 < Sp && 2 * u + 1 < MS ? tok1 : 0, ra1);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = tok0 < Sp ? fmaf(acc[2 * u][ni][j], ra0[j], bb[ni]) : 0.f;
          if (2 * u + 1 < MS) v[4 + j] = tok1 < Sp ? fmaf(acc[2 * u + 1][ni][j], ra1[j], bb[ni]) : 0.f;
          else v[4 + j] = 0.f;
        }
        epi.template store_keys<8>(b, eg0 + 16 * ni, (s0 >> 1) * 32 + 8 * g, v, am);
      }
    }
  }
  epi.flag(am);
}

// a (free) register use that makes the compiler wait for the column constants HERE, once, and not at their first use inside
// the row loop - where the wait would be repeated per row tile and then also cover the previous row tile's stores
template <int NC, int CH>
TAMF_DEV void clip_settle(const float (&bi)[NC][CH]) {
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < CH; ++j) asm volatile("" ::"v"(bi[c][j]));
}

// barrier of the Y waves: they issue no loads, and their epilogue stores may drain behind it (no vmcnt wait)
TAMF_DEV void clip_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]); as a builtin, so that the
// compiler's own wait-count bookkeeping sees it
// The counts are compile-time formulas of the instructions issued behind the request that must have landed (PH LDS-DMA pieces,
// SX epilogue stores): tools/check_clip_stores.py (tests/test_isa_clip_waits.py) disassembles every instantiation and checks that
// hipcc emitted exactly the stores the formula counts.  -DTAMF_CLIP_SAFE_WAIT turns every counted wait into vmcnt(0) (a debug
// build to compare bits against: tools/ab_build.sh WORKTREE S with TAMF_HIPCC_FLAGS=-DTAMF_CLIP_SAFE_WAIT).
template <int N>
TAMF_DEV void clip_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
#ifdef TAMF_CLIP_SAFE_WAIT
  __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));
#else
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
#endif
}

// The K tiles this workgroup still has to request, across its tiles: round ri (tile clip_tile_of(ri)), K tile kti
template <class Op, class C>
struct ClipStream {
  ClipSrc s;
  int ri, kti;
  bool live;
  TAMF_DEV void open(const ClipGemmArgs<Op>& ga, int ntn, int nq, int prow, int pch) {
    ri = 0; kti = 0;
    load_src(ga, ntn, nq, prow, pch);
  }
  TAMF_DEV void load_src(const ClipGemmArgs<Op>& ga, int ntn, int nq, int prow, int pch) {
    const int t = clip_tile_of(ga, ri);
    live = t >= 0;
    if (live) s = clip_src<Op, C>(ga, t / ntn, (t % ntn) * C::BN, nq, prow, pch);
  }
  TAMF_DEV void advance(const ClipGemmArgs<Op>& ga, int KT, int ntn, int nq, int prow, int pch) {
    if (++kti == KT) {
      kti = 0;
      ++ri;
      load_src(ga, ntn, nq, prow, pch);
    }
  }
};

// Deferred LayerNorm: the row terms of the workgroup's first TWO tiles (slots 0 and 1), staged by ALL eight waves ahead of the first
// barrier, while the first K tiles are in flight - most launches give a workgroup one or two tiles, and staging a tile at the boundary
// in front of it waits behind the epilogue's stores
template <class Op, class C, class Epi>
TAMF_DEV void clip_stage_first(const ClipGemmArgs<Op>& ga, const Epi& epi, int ntn, float2* rstat, int tid) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int tr = clip_tile_of(ga, r);
    if (tr >= 0) {
      int base, rows;
      clip_part(ga, tr / ntn, base, rows);
      ln_stage<512, Epi::STAGE_AFF, (C::MT + 127) / 128>(epi.ln, epi.ctl.wscale, base, rows, base + rows, rstat + r * C::MT, tid);
    }
  }
}

template <class Op, int NSUB, int NI, int XSUB, class Epi>
__global__ __launch_bounds__(512, 2) void clip_gemm_kernel(const ClipGemmArgs<Op> ga, const Epi epi) {
  typedef ClipCfg<NSUB, NI, XSUB, Epi::LANE_CHUNK> C;
  constexpr int BKB = GEMM_BKB, NS = C::NSTAGE, LA = NS - 1;
  constexpr bool TR = Epi::TRANSPOSED;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;  // (& 7: lets the compiler fold the piece bounds)
  const int lr = lane & 15, g = lane >> 4;
  const int mh = wave >> 2, nq = wave & 3;
  static_assert(C::MSUBX >= 2 && C::MSUBY >= 1, "row tiles per wave half");
  const int wm0 = mh * C::MSUBX * 16, wn0 = nq * (NI * 16);
  const int KT = (ga.K * Op::EB) / BKB;
  const int ntn = ga.N / C::BN;
  const int prow = lane >> 3, pch = lane & 7;

  // fragment addressing (as gemm_tile): lane (lr, g) reads chunks g and 4 + g of tile row lr (+16 per MFMA tile)
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = (wm0 + lr) * BKB;
  const int w_frag = (C::MT + wn0 + lr) * BKB;
  const int lane_col = wn0 + C::CHUNK * g;  // first of the lane's output columns inside the tile (clip_wperm)
  char* slot = smem + C::SCRATCH_OFF + wave * C::SCRATCH_WAVE + lane * 16;
  float2* const rstat = (float2*)(smem + C::STATS_OFF);  // [3][MT] row terms of the tile of round r: slot r % 3

  // The workgroup is persistent over its tiles (rounds of the grid) and treats their K tiles as ONE stream: interval j
  // belongs to K tile j % KT of round j / KT and lives in stage j % NS.  X multiplies K tile j in interval j and requests K tile
  // j + LA (possibly the next tile's) into the stage that K tile j - 1 has just left; Y multiplies K tile j - 1 from registers
  // and reads K tile j.  A tile's rows are stored by X at the head of the next tile's first interval - while Y multiplies
  // the last K tile - and by Y behind those MFMAs, while X is back in the K loop: no interval without MFMAs, no drain.
  const int G = gridDim.x;
  const int my_tiles = ga.n_tiles / G + ((int)blockIdx.x < ga.n_tiles % G ? 1 : 0);
  if (my_tiles == 0) return;
  const int J = my_tiles * KT;

  // Two separate loops (not one loop with a branch inside): Y's fragment registers are loop-carried and would otherwise be
  // live - and spilled - across X's code.  Both execute the same J + 1 barriers.
  if (mh == 0) {
    ClipStream<Op, C> is;
    is.open(ga, ntn, nq, prow, pch);
#pragma unroll
    for (int p = 0; p < LA; ++p) {  // (J >= KT >= 2 >= LA)
      clip_issue<Op, C>(ga, is.s, nq, prow, is.kti, smem + p * C::STAGE);
      is.advance(ga, KT, ntn, nq, prow, pch);
    }
    f32x4 acc[C::MSUB0][NI];
#pragma unroll
    for (int mi = 0; mi < C::MSUBX; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (Epi::ROWSTATS) clip_stage_first<Op, C, Epi>(ga, epi, ntn, rstat, tid);
    // K tile 0 has landed: everything but the LA - 1 requests behind it
    if constexpr (LA == 2) {
      if (nq < C::PIECES_REM) clip_wait_vm<C::PIECES_HI>(); else clip_wait_vm<C::PIECES_HI - 1>();
    } else {
      clip_wait_vm<0>();
    }
    clip_barrier_lds();
    int round = 0, kt = 0, sc = 0, sn = LA % NS;
    int t = clip_tile_of(ga, 0);
    bool pre = false;  // this interval's requests went out ahead of the previous tile's epilogue
    for (int j = 0; j < J; ++j) {
      const bool ld = is.live && !(TAMF_ABL(ga.abl) & 1) && !pre, batch = ld || pre;
#ifdef TAMF_TIMELINE
      const bool dbg_on = wave == 0 && j >= 4 && j < 12 && blockIdx.x < 512;
      const int it_dbg = j, lane_dbg = lane, mh_dbg = 0;
      TAMF_CLIP_TS(0)
      clip_ktile_x<Op, C, NI, TR>(smem + sc * C::STAGE, smem + sn * C::STAGE, ld, !(TAMF_ABL(ga.abl) & 2), ga, is.s, nq, prow, is.kti, a_frag, w_frag,
                              c0, c1, acc, dbg_on, it_dbg, lane_dbg);
      TAMF_CLIP_TS(2)
#else
      clip_ktile_x<Op, C, NI, TR>(smem + sc * C::STAGE, smem + sn * C::STAGE, ld, !(TAMF_ABL(ga.abl) & 2), ga, is.s, nq, prow, is.kti, a_frag, w_frag,
                              c0, c1, acc);
#endif
      if (is.live && !pre) is.advance(ga, KT, ntn, nq, prow, pch);
      // K tile j + 1 has landed: everything but what was issued behind it (vmcnt retires in order).  In a tile's first interval
      // that is this interval's requests AND the SX stores of the previous tile's rows, which went out behind them: the wait must
      // not cover those - a store is acknowledged when the L2 has taken it, and the 256 CUs store their tiles at the same moment.
      constexpr int SX = TR ? ((C::MSUBX + (Op::PREC == 0 ? 0 : 1)) / (Op::PREC == 0 ? 1 : 2)) * NI * Epi::CHUNK_STORES
                            : C::MSUBX * C::NCHUNK * Epi::CHUNK_STORES;
      constexpr int PH = C::PIECES_HI;
      static_assert(PH + SX < 64, "vmcnt range");
      const bool behind = pre && !(TAMF_ABL(ga.abl) & 4);  // the stores are there
      pre = false;
      if (LA == 2 && batch) {
        if (behind) {
          if (nq < C::PIECES_REM) clip_wait_vm<PH + SX>(); else clip_wait_vm<PH - 1 + SX>();
        } else {
          if (nq < C::PIECES_REM) clip_wait_vm<PH>(); else clip_wait_vm<PH - 1>();
        }
      } else if (LA == 1 && behind) {
        clip_wait_vm<SX>();
      } else {
        clip_wait_vm<0>();
      }
      clip_barrier_lds();
      TAMF_CLIP_TS(3)
      sc = sc + 1 == NS ? 0 : sc + 1;
      sn = sn + 1 == NS ? 0 : sn + 1;
      if (++kt == KT) {  // the tile is complete: its rows go out while Y multiplies its last K tile
        TAMF_CLIP_XPRIO_HI
        const int b = t / ntn, n0 = (t % ntn) * C::BN;
        // column constants, then the next interval's requests (the epilogue must not delay them; and vmcnt retires in order: the
        // constants are waited for with the requests still in flight), then the rows
        float bi[C::NCHUNK][C::CHUNK];
        float ci[C::NCHUNK][C::CHUNK];
        float bb[NI];
        if constexpr (TR) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bb[ni] = epi.bias[n0 + wn0 + lr + 16 * ni];
        } else {
#pragma unroll
          for (int c = 0; c < C::NCHUNK; ++c) {
            if constexpr (Epi::ROWSTATS) epi.template lane_cols_ln<C::CHUNK>(n0 + lane_col + 4 * C::CHUNK * c, bi[c], ci[c]);
            else if (TAMF_ABL(ga.abl) & 32) { for (int j = 0; j < C::CHUNK; ++j) bi[c][j] = 0.f; }  // (ablation: no column constants)
            else epi.template lane_cols<C::CHUNK>(n0 + lane_col + 4 * C::CHUNK * c, bi[c]);
          }
        }
        if (is.live && !(TAMF_ABL(ga.abl) & 1)) {
          clip_issue<Op, C>(ga, is.s, nq, prow, is.kti, smem + sn * C::STAGE);
          is.advance(ga, KT, ntn, nq, prow, pch);
          pre = true;
        }
        if constexpr (TR) {
          if (!(TAMF_ABL(ga.abl) & 4)) clip_store_vt<Op, C, NI, 0, C::MSUBX>(epi, acc, g, ga.Sp, b, n0 + wn0 + lr, bb, rstat + (round % 3) * C::MT);
        } else {
          clip_settle(bi);
          if constexpr (Epi::ROWSTATS) clip_settle(ci);
          int base, rows;
          clip_part(ga, b, base, rows);
          if (!(TAMF_ABL(ga.abl) & 4))
            clip_store_rows<C, NI, C::MSUBX>(epi, acc, wm0 + lr, rows, base, n0 + lane_col, bi, ci, rstat + (round % 3) * C::MT, TAMF_ABL(ga.abl), slot);
        }
#pragma unroll
        for (int mi = 0; mi < C::MSUBX; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        kt = 0;
        t = clip_tile_of(ga, ++round);
        if constexpr (Epi::ROWSTATS) {
          // tiles 0 and 1 were staged up front; entering tile `round` >= 1, tile round + 1 (if any) goes into slot (round + 1) % 3 = the
          // slot of tile round - 2, which nobody reads any more (Y stored tile round - 2 before a barrier X passed a whole tile ago;
          // Y may still be storing tile round - 1 from ITS slot)
          if (round >= 1) {
            const int tn = clip_tile_of(ga, round + 1);
            if (tn >= 0) {
              int base, rows;
              clip_part(ga, tn / ntn, base, rows);
              ln_stage<256, Epi::STAGE_AFF, (C::MT + 63) / 64>(epi.ln, epi.ctl.wscale, base, rows, base + rows, rstat + ((round + 1) % 3) * C::MT, tid);
            }
          }
        }
      }
    }
  } else {
    // Y's MFMAs go first on the SIMD (static priority, no per-interval flips): they are ready at the top of the interval,
    // while X spends its head on the DMA pieces anyway; served from the leftovers of the older X wave (equal priority:
    // the older wave wins arbitration) Y finished LAST - 2 620 of 3 300 cycles - and its fragment reads and the barrier
    // followed with the matrix pipe idle.  With priority Y is done after ~1 200 cycles and reads while X multiplies.
    __builtin_amdgcn_s_setprio(2);
    f32x4 acc[C::MSUB0][NI];
#pragma unroll
    for (int mi = 0; mi < C::MSUBY; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (Epi::ROWSTATS) clip_stage_first<Op, C, Epi>(ga, epi, ntn, rstat, tid);
    clip_barrier_lds();  // K tile 0 has landed (X waited for it)
    int4 ywf[NI][2], yaf[C::MSUBY][2];
    clip_read_y<C, NI>(smem, a_frag, w_frag, c0, c1, ywf, yaf);
    clip_barrier_lds();  // (the fragment reads are complete: lgkmcnt(0) before every barrier)
    int round = 0, kt = 0, sc = 1 % NS;
    int t = clip_tile_of(ga, 0);
    for (int j = 1; j <= J; ++j) {
#ifdef TAMF_TIMELINE
      const bool dbg_on = wave == 4 && j >= 4 && j < 12 && blockIdx.x < 512;
      const int it_dbg = j, lane_dbg = lane, mh_dbg = 1;
#endif
      TAMF_CLIP_TS(0)
      // (at 256 columns the fused form needs > 256 registers with 7 row tiles in Y: f32 gives its Y waves 6 there, tamf_hip.hip ClipXsub)
      constexpr bool YFUSE = (NI == 2 || C::MSUBY <= 6) && (TAMF_CLIP_YFUSE & (Op::PREC == 0 ? 1 : 2)) != 0;
      if constexpr (YFUSE) clip_mma_read_y<Op, C, NI, TR>(smem + sc * C::STAGE, j < J, a_frag, w_frag, c0, c1, ywf, yaf, acc);  // K tile j - 1, fragments of K tile j
      else if (!(TAMF_ABL(ga.abl) & 2)) clip_mma_y<Op, C, NI, TR>(ywf, yaf, acc);  // K tile j - 1
#ifdef TAMF_TIMELINE
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[C::MSUBY - 1][NI - 1][3]));  // the stamp waits for the MFMA results
#endif
      TAMF_CLIP_TS(1)
      if (++kt == KT) {
        const int b = t / ntn, n0 = (t % ntn) * C::BN;
        if constexpr (TR) {
          float bb[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bb[ni] = epi.bias[n0 + wn0 + lr + 16 * ni];
          if (!(TAMF_ABL(ga.abl) & 4)) clip_store_vt<Op, C, NI, C::MSUBX, C::MSUBY>(epi, acc, g, ga.Sp, b, n0 + wn0 + lr, bb, rstat + (round % 3) * C::MT);
        } else {
          float bi[C::NCHUNK][C::CHUNK];
          float ci[C::NCHUNK][C::CHUNK];
#pragma unroll
          for (int c = 0; c < C::NCHUNK; ++c) {
            if constexpr (Epi::ROWSTATS) epi.template lane_cols_ln<C::CHUNK>(n0 + lane_col + 4 * C::CHUNK * c, bi[c], ci[c]);
            else if (TAMF_ABL(ga.abl) & 32) { for (int j = 0; j < C::CHUNK; ++j) bi[c][j] = 0.f; }  // (ablation: no column constants)
            else epi.template lane_cols<C::CHUNK>(n0 + lane_col + 4 * C::CHUNK * c, bi[c]);
          }
          clip_settle(bi);
          if constexpr (Epi::ROWSTATS) clip_settle(ci);
          int base, rows;
          clip_part(ga, b, base, rows);
          if (!(TAMF_ABL(ga.abl) & 4))
            clip_store_rows<C, NI, C::MSUBY>(epi, acc, wm0 + lr, rows, base, n0 + lane_col, bi, ci, rstat + (round % 3) * C::MT, TAMF_ABL(ga.abl), slot);
        }
#pragma unroll
        for (int mi = 0; mi < C::MSUBY; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        kt = 0;
        t = clip_tile_of(ga, ++round);
      }
      if (j == J) break;
      if constexpr (!YFUSE) clip_read_y<C, NI>(smem + sc * C::STAGE, a_frag, w_frag, c0, c1, ywf, yaf);
#ifdef TAMF_TIMELINE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      TAMF_CLIP_TS(2)
      clip_barrier_lds();
      TAMF_CLIP_TS(3)
      sc = sc + 1 == NS ? 0 : sc + 1;
    }
    __builtin_amdgcn_s_setprio(0);
  }
}
