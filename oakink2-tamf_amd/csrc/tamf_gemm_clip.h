// Clip-aligned GEMM for the token-row GEMMs of the encoder layers (QKV, FFN1, FFN2):
//
//   C[b*Sp + r][n] = sum_k A[b*Sp + r][k] * W[n][k]          one M tile = the Sp token rows of ONE clip
//
// Why its own kernel.  The 128 x 128 tiles of tamf_gemm.h stage (128 + 128) * 128 bytes from L2 per 128-byte K tile;
// measured on MI355X (tools/kbench.py, ablation bits), their K loop is bound by the L2 -> LDS path of a CU (about
// 70 GB/s per CU, MI355X_MICROARCH.md "Indexed rows: gather into LDS"), not by the matrix pipe: the loads alone take as
// long as the MFMAs alone.  And M = 64 clips * 208 rows = 13 * 2^10 rows never fills 256 CUs evenly with power-of-two
// row tiles (every launch ends in a 3.25th round).  A tile of one whole clip (208 rows = 13 MFMA row tiles) by 256 / 192 /
// 128 columns stages (208 + BN) * 128 bytes per K tile for 208 * BN outputs - 1.8x / 1.6x / 1.25x the flops per staged
// byte - and B = 64 clips give 512 / 512 / 256 tiles: exact rounds of the 256 CUs.
//
// Workgroup = 8 waves (one per CU, two waves per SIMD), wave grid 2 (M) x 4 (N): waves 0-3 own row tiles 0-6, waves 4-7
// row tiles 7-12 (waves w and w + 4 share a SIMD, so every SIMD carries 7 + 6 row tiles), each over BN / 4 columns.
// Staging, swizzle, fragment addressing and the MFMA operand traits are those of tamf_gemm.h (LDS-DMA pieces of 8 rows
// x 128 bytes, source-side XOR swizzle, all fragments of a K tile requested up front, next tile's pieces issued after the
// fragment reads).  Workgroups are persistent over their tiles; the first K tile of the next tile is requested before the
// epilogue of the current one.  The accumulators are parked in LDS in slabs of 64 rows and handed to the same row-wise
// epilogue functors as the other GEMMs.
#pragma once
#include "tamf_gemm.h"

template <class Op>
struct ClipGemmArgs {
  const typename Op::elem_t* A;
  int lda;  // elements
  const typename Op::elem_t* W;
  int ldw;
  int n_clips, Sp;  // A has n_clips * Sp rows; a tile covers rows [b*Sp, b*Sp + Sp)
  int N, K;
  int n_tiles;      // n_clips * (N / BN)
};

template <int NSUB, int NI>
struct ClipCfg {
  static constexpr int MT = NSUB * 16;        // tile rows (>= Sp)
  static constexpr int BN = 64 * NI;          // tile columns: 4 waves x NI MFMA column tiles
  static constexpr int MSUB0 = (NSUB + 1) / 2, MSUB1 = NSUB - MSUB0;
  static constexpr int ROWS = MT + BN;        // rows of one staged K tile: A rows then W rows
  static constexpr int STAGE = ROWS * GEMM_BKB;
  static constexpr int NPIECE = ROWS / 8, A_PIECES = MT / 8;
  static constexpr int NPW = (NPIECE + 7) / 8;  // pieces per wave
  static constexpr int SLAB = 64;             // rows per epilogue slab
  static constexpr int LDC = BN + 4;
  static constexpr int C_OFF = STAGE;         // the C slab overlays stage 1: stage 0 stays free for the next tile's first K tile
  static constexpr int CBYTES = SLAB * LDC * 4;
  static constexpr int BYTES = (2 * STAGE > C_OFF + CBYTES) ? 2 * STAGE : C_OFF + CBYTES;
  static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// source byte offsets of this lane's pieces of tile (clip b, column tile at n0): piece q = wave + 8 i covers staged rows
// [8q, 8q + 8); rows < MT come from the clip's A rows (clamped to the clip), the others from the W rows of the column tile
template <class Op, class C>
TAMF_DEV void clip_tile_offsets(unsigned (&off)[C::NPW], const ClipGemmArgs<Op>& ga, int b, int n0, int wave, int prow, int pch) {
#pragma unroll
  for (int i = 0; i < C::NPW; ++i) {
    const int row = (wave + 8 * i) * 8 + prow;
    const int swz = (pch ^ swz_chunk<GEMM_BKB>(row)) << 4;
    if (row < C::MT) {
      const int r = row < ga.Sp ? row : ga.Sp - 1;
      off[i] = (unsigned)(((long)b * ga.Sp + r) * ga.lda * Op::EB + swz);
    } else {
      const int wr = row - C::MT;
      off[i] = (unsigned)((long)(n0 + (wr < C::BN ? wr : C::BN - 1)) * ga.ldw * Op::EB + swz);
    }
  }
}
template <class C>
TAMF_DEV void clip_issue(const unsigned (&off)[C::NPW], const char* Ab, const char* Wb, int kt, char* stage_base, int wave) {
#pragma unroll
  for (int i = 0; i < C::NPW; ++i) {
    const int q = wave + 8 * i;
    if (C::NPIECE % 8 == 0 || q < C::NPIECE) {
      const char* src = (q < C::A_PIECES ? Ab : Wb) + off[i] + (long)kt * GEMM_BKB;
      glds16<0>(src, stage_base + q * 1024);
    }
  }
}

// One K tile of one wave: request the next K tile into `nxt`, then multiply the tile in `cur`.  `cur` and `nxt` are the two
// LDS stages and never overlap; the __restrict__ qualifiers of this (inlined) helper are what keeps hipcc from placing an
// s_waitcnt vmcnt(0) in front of the LDS reads that follow the LDS-DMA in program order (same device as in tamf_attn.h), so
// the A fragments can be streamed two row tiles ahead of their MFMAs instead of being held all at once (7 x 8 registers).
template <class Op, class C, int NI>
TAMF_DEV void clip_ktile(const char* __restrict__ cur, char* __restrict__ nxt, bool more, const unsigned (&off)[C::NPW],
                         const char* Ab, const char* Wb, int kt_next, int wave, int msub, int a_frag, int w_frag, int c0, int c1,
                         f32x4 (&acc)[C::MSUB0][NI]) {
  constexpr int BKB = GEMM_BKB;
  int4 wf[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    wf[ni][0] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c0);
    wf[ni][1] = *(const int4*)(cur + w_frag + ni * 16 * BKB + c1);
  }
  int4 af[C::MSUB0][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    af[mi][0] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c0);
    af[mi][1] = *(const int4*)(cur + a_frag + mi * 16 * BKB + c1);
  }
  if (more) clip_issue<C>(off, Ab, Wb, kt_next, nxt, wave);
#pragma unroll
  for (int mi = 0; mi < C::MSUB0; ++mi) {
    if (mi + 2 < C::MSUB0 && (mi + 2 < C::MSUB1 || mi + 2 < msub)) {
      af[mi + 2][0] = *(const int4*)(cur + a_frag + (mi + 2) * 16 * BKB + c0);
      af[mi + 2][1] = *(const int4*)(cur + a_frag + (mi + 2) * 16 * BKB + c1);
    }
    if (mi < C::MSUB1 || mi < msub) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) Op::mma(acc[mi][ni], wf[ni], af[mi]);  // D rows = n (4g + reg), cols = m (lr)
    }
  }
}

TAMF_DEV int clip_tile_of(int n_tiles, int round) {
  const int G = gridDim.x, base = round * G;
  if (base >= n_tiles) return -1;
  const int cnt = n_tiles - base < G ? n_tiles - base : G;
  return (int)blockIdx.x < cnt ? base + xcd_remap(blockIdx.x, cnt) : -1;
}

template <class Op, int NSUB, int NI, int SUBN, class Epi>
__global__ __launch_bounds__(512, 2) void clip_gemm_kernel(const ClipGemmArgs<Op> ga, const Epi epi) {
  typedef ClipCfg<NSUB, NI> C;
  constexpr int BKB = GEMM_BKB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;  // (& 7: lets the compiler fold the piece bounds)
  const int lr = lane & 15, g = lane >> 4;
  const int mh = wave >> 2, nq = wave & 3;
  const int msub = mh ? C::MSUB1 : C::MSUB0;
  const int wm0 = mh * C::MSUB0 * 16, wn0 = nq * (NI * 16);
  const int KT = (ga.K * Op::EB) / BKB;
  const int ntn = ga.N / C::BN;
  const char* Ab = (const char*)ga.A;
  const char* Wb = (const char*)ga.W;
  const int prow = lane >> 3, pch = lane & 7;

  // fragment addressing (as gemm_tile): lane (lr, g) reads chunks g and 4 + g of tile row lr (+16 per MFMA tile)
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = (wm0 + lr) * BKB;
  const int w_frag = (C::MT + wn0 + lr) * BKB;

  unsigned off[C::NPW];

  // tile of this workgroup in round `r` of the persistent grid (-1: none): inside a round the XCDs own contiguous chunks
  // of the tile list (xcd_remap), i.e. a few whole clips x all their column tiles - a clip's A panel is fetched into one L2
  int round = 0;
  int t = clip_tile_of(ga.n_tiles, round);
  if (t < 0) return;
  clip_tile_offsets<Op, C>(off, ga, t / ntn, (t % ntn) * C::BN, wave, prow, pch);
  clip_issue<C>(off, Ab, Wb, 0, smem, wave);
  while (true) {
    const int b = t / ntn, n0 = (t % ntn) * C::BN;
    const int m0 = b * ga.Sp;
    f32x4 acc[C::MSUB0][NI];
#pragma unroll
    for (int mi = 0; mi < C::MSUB0; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();  // K tile 0 has landed (vmcnt(0) + barrier); the previous tile's C slab has been consumed

    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      clip_ktile<Op, C, NI>(smem + cur * C::STAGE, smem + (cur ^ 1) * C::STAGE, kt + 1 < KT, off, Ab, Wb, kt + 1, wave, msub,
                            a_frag, w_frag, c0, c1, acc);
      __syncthreads();
    }

    // the next tile's first K tile goes into stage 0 (last read at kt = KT - 2, KT is even) while this tile's epilogue runs
    const int tn = clip_tile_of(ga.n_tiles, ++round);
    const bool more = tn >= 0;
    if (more) {
      clip_tile_offsets<Op, C>(off, ga, tn / ntn, (tn % ntn) * C::BN, wave, prow, pch);
      clip_issue<C>(off, Ab, Wb, 0, smem, wave);
    }
    // epilogue in slabs of 64 rows: a wave parks the row tiles it holds that fall into the slab, then all 8 waves walk it
    float* Ct = (float*)(smem + C::C_OFF);
    constexpr int NSLAB = (C::MT + C::SLAB - 1) / C::SLAB;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
#pragma unroll
      for (int mi = 0; mi < C::MSUB0; ++mi) {
        const int s = mh * C::MSUB0 + mi;  // row tile of the clip
        if ((mi < C::MSUB1 || mi < msub) && (s >> 2) == sl) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const f32x4 v = acc[mi][ni];
            *(float4*)(Ct + ((s & 3) * 16 + lr) * C::LDC + wn0 + ni * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      __syncthreads();
      // column sub-blocks of SUBN for epilogues that decide per block (QKV: Q | K | V boundaries are multiples of 64)
#pragma unroll
      for (int cs = 0; cs < C::BN; cs += SUBN)
        epi.template run<C::SLAB, SUBN, 512>(Ct + cs, C::LDC, m0 + sl * C::SLAB, n0 + cs, m0 + ga.Sp, tid);
      __syncthreads();
    }
    if (!more) break;
    t = tn;
  }
}
