// LDS-staged MFMA GEMM for gfx950 with row-wise fused epilogues.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: [M][lda], W: [N][ldw] operand rows (K contiguous, Op::EB bytes/element)
//
// A workgroup owns a BM x BN tile: 128 x 128 with 4 waves (2 x 2; two workgroups per CU) or 64 x d with 8 waves
// (2 x 4; the LayerNorm-fused GEMMs need whole rows).  K is walked in tiles of 128 bytes per row - 64 bf16, 32 f32 or
// 32 split-bf16 elements, whose hi and lo halves are interleaved at 64-byte granularity so that one tile row is one
// full 128-byte line in every mode.  Tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1-KiB "piece" =
// 8 rows per wave-instruction, LDS destination = wave-uniform base + 16*lane), double-buffered, one __syncthreads()
// per K tile; the pieces of tile k+1 are issued before the MFMAs of tile k.  LDS rows are XOR-swizzled
// (swz_chunk<128>, tools/lds_bank_sim.py) so the ds_read_b128 fragment reads are bank-conflict free; because the
// DMA writes linearly the swizzle is applied to each lane's SOURCE address.  The workgroup id is remapped so that
// all N-tiles of one M-tile run on the same XCD (the A panel is fetched into one L2 instead of eight).  After the K
// loop the fp32 accumulators are parked in an LDS C tile (aliasing the staging buffers) and the epilogue walks it
// row-wise with 16-byte global accesses: bias / activation / residual + LayerNorm / V-transpose / DDPM update are all
// fused here, so every GEMM output is written to HBM exactly once, already in the operand format of its consumer.
#pragma once
#include "tamf_device.h"

template <class Op>
struct GemmArgs {
  const typename Op::elem_t* A;
  int lda;  // elements
  const typename Op::elem_t* W;
  int ldw;
  int M, N, K;
  // bits 0-7: K-loop rotation stride per workgroup (0 = off); bits 8-11: L2 touch-prefetch distance in K tiles;
  // bit 12: ablation "no loads after tile 0"; bit 13: ablation "no compute"; bit 16: 4 x 2 XCD arrangement of the tile grid
  int krot;
  // remainder splitting (set by the launcher): workgroups [0, n_full) take whole BM x BN tiles; the tiles left over
  // after the last full round of the chip are cut into SPLIT column slices, one workgroup each (see gemm_kernel)
  int n_full;
  // > 0: persistent launch - the grid is one round of workgroups and workgroup b walks tiles b, b + grid, b + 2 grid, ...
  // (n_tiles in total), which gives every CU the same number of tiles +-1 where the hardware's greedy dispatch does not
  int n_tiles;
};

constexpr int GEMM_BKB = 128;  // bytes per tile row per K tile, every mode

template <int BM, int BN>
struct GemmSmem {
  static constexpr int LDC = BN + 4;
  static constexpr int STAGE = (BM + BN) * GEMM_BKB;
  static constexpr int CBYTES = BM * LDC * 4;
  static constexpr int BYTES = (2 * STAGE > CBYTES) ? 2 * STAGE : CBYTES;
  // + (mean, rstd) of the tile's rows for the epilogues that normalise (deferred LayerNorm, tamf_device.h): staged ahead of the K loop
  static constexpr int STATS_OFF = BYTES, TOTAL = BYTES + BM * 8;
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  Each provides run<BM,BN,NT>(Ct, LDC, m0, n0, M, tid): Ct is the fp32 C tile in LDS.
// ------------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };

TAMF_DEV void ct_load8(const float* Ct, int LDC, int row, int col, float (&v)[8]) {
  const float4 a = *(const float4*)(Ct + row * LDC + col);
  const float4 b = *(const float4*)(Ct + row * LDC + col + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
TAMF_DEV void g_load8(const float* p, float (&v)[8]) {
  const float4 a = *(const float4*)p;
  const float4 b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <int N>
TAMF_DEV void g_loadn(const float* p, float (&v)[N]) {
  static_assert(N % 4 == 0, "16-byte loads");
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    const float4 a = *(const float4*)(p + j);
    v[j] = a.x; v[j + 1] = a.y; v[j + 2] = a.z; v[j + 3] = a.w;
  }
}
// a (free) register use that makes the compiler wait for loaded values HERE, once, ahead of a loop that stores: vmcnt retires in
// order, so a wait for a load that the compiler places at its first use INSIDE such a loop also waits for the previous
// iteration's stores - an epilogue then runs one store round trip per row (seen in the ISA as s_waitcnt vmcnt(0) between stores)
template <int N>
TAMF_DEV void settle(const float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" ::"v"(v[j]));
}
// Deferred LayerNorm in front of a GEMM (tamf_device.h).  LN(u) = rstd ((u - mean 1) o gamma) + beta, and the centring u - mean 1 = C u
// with C = I - 1 1^T / d is linear, so it is folded into the weight together with the gain: W'' = W diag(gamma) C, i.e.
// W''[n][k] = gamma[k] W[n][k] - (sum_j gamma[j] W[n][j]) / d (rows of zero sum, tamf_finalize_weights).  Then
//   LN(u) . W^T = rstd[m] (u . W''^T)[m][n] + c2[n],   c2 = W beta + b
// and the epilogue is ONE fma with a per-row factor, out = fma(acc, ra, c2[n]), ra = rstd ws - the cost of the plain fma(acc, ws, bias).
// With rstd = 1 - no LayerNorm in front - it IS fma(acc, ws, bias).  The kernel stages (ra, -) per row of the tile (ln_stage<.., true>).
TAMF_DEV float aff(float acc, float2 st, float c2) { return fmaf(acc, st.x, c2); }

TAMF_DEV void g_store8(float* p, const float (&v)[8]) {
  gst16f(p, v[0], v[1], v[2], v[3]);
  gst16f(p + 4, v[4], v[5], v[6], v[7]);
}

// out = act(C + bias[n] + rowadd[m][n]) stored as an operand (FFN1+GELU, input_merge.0+SiLU, hoisted GEMMs)
// LN = true: the A operand holds UN-normalised rows u and the LayerNorm in front of this GEMM is applied here (deferred LayerNorm):
// the weight is W diag(gamma) C (gain and centring folded in), bias = c2 = W beta + b, ln = partial statistics of the rows (staged by
// the kernel: `rs`).
template <class OutOp, bool LN = false>
struct EpiBiasAct {
  const float* bias;    // [N] or null
  const float* rowadd;  // [M][ld_rowadd] or null
  int ld_rowadd;
  typename OutOp::elem_t* out;
  int ldo;
  int act;
  EpiCtl ctl;
  LnStats ln{};
  static constexpr bool ROWSTATS = LN;
  static constexpr bool STAGE_AFF = true;  // the kernel stages the row factor ra = rstd ws
  static constexpr bool PREFETCH = false;
  // Column constants of a thread (its 8 columns are the same for all its rows): loaded once per tile, ahead of the row
  // loops - a global load inside the row loop serialises the loop on L2 latency, and a load issued after stores waits for
  // them (vmcnt is in order), which is why the slab-wise epilogue of tamf_gemm_clip.h fetches these before its first slab
  struct Cols {
    float bi[8];
    // a (free) register use that makes the compiler wait for the loads HERE, once: left to the first use inside a row loop,
    // its s_waitcnt vmcnt(0) is repeated every iteration and then waits for the previous iteration's stores
    TAMF_DEV void settle() const {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(bi[j]));
    }
  };
  template <int BN, int NT>
  TAMF_DEV Cols cols(int n0, int tid) const {
    constexpr int VPR = BN / 8;
    Cols c;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.bi[j] = 0.f;
    if (bias) g_load8(bias + n0 + (tid % VPR) * 8, c.bi);
    return c;
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid, const float2* rs = nullptr) const {
    run_c<BM, BN, NT>(Ct, LDC, m0, n0, M, tid, cols<BN, NT>(n0, tid), rs);
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run_c(const float* Ct, int LDC, int m0, int n0, int M, int tid, const Cols& cc, const float2* rs = nullptr) const {
    constexpr int VPR = BN / 8, RSTEP = NT / VPR;
    static_assert(NT % VPR == 0, "column group must be fixed per thread");
    const int col = (tid % VPR) * 8, gn = n0 + col;
    float am = 0.f;
    if constexpr (LN) {
      cc.settle();
      for (int row = tid / VPR; row < BM; row += RSTEP) {
        const int gr = m0 + row;
        if (gr >= M) break;
        float v[8];
        ct_load8(Ct, LDC, row, col, v);
        finish_ln<8>(act, gr, gn, v, cc.bi, cc.bi, rs[row], am);
      }
    } else if (rowadd) {
      // the row terms of all of this thread's rows are requested (and waited for) before the first store: a load issued behind
      // stores waits for them (vmcnt retires in order) - one store round trip per row otherwise (input_merge.0)
      constexpr int NR = BM / RSTEP;
      static_assert(BM % RSTEP == 0, "rows per thread");
      float ra[NR][8];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        int gr = m0 + tid / VPR + i * RSTEP;
        gr = gr < M ? gr : M - 1;
        g_load8(rowadd + (long)gr * ld_rowadd + gn, ra[i]);
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) settle(ra[i]);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int row = tid / VPR + i * RSTEP, gr = m0 + row;
        if (gr < M) {
          float v[8];
          ct_load8(Ct, LDC, row, col, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], ctl.wscale, cc.bi[j]) + ra[i][j];
          act_store<8>(act, gr, gn, v, am);
        }
      }
    } else {
      for (int row = tid / VPR; row < BM; row += RSTEP) {
        const int gr = m0 + row;
        if (gr >= M) break;
        float v[8];
        ct_load8(Ct, LDC, row, col, v);
        finish_act<8>(act, gr, gn, v, cc.bi, am);
      }
    }
    OutOp::range_flag(am, ctl.status);
  }
  TAMF_DEV void flag(float am) const { OutOp::range_flag(am, ctl.status); }
  // bias, row term, activation and operand store of N (4 or 8) consecutive columns gn .. of row gr: shared by the LDS-walking
  // form above and the register form of tamf_gemm_clip.h, so both produce the same bits.  `a` = this->act (the register form
  // passes it as a literal per branch, so that its unrolled row tiles carry one activation).  `am`: range accumulator of the
  // caller (Op::store_rc), flagged once after its loops
  template <int N>
  TAMF_DEV void finish_act(int a, int gr, int gn, float (&v)[N], const float (&bi)[N], float& am) const {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = fmaf(v[j], ctl.wscale, bi[j]);
    if (rowadd) {
      float b[N];
      g_loadn<N>(rowadd + (long)gr * ld_rowadd + gn, b);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] += b[j];
    }
    act_store<N>(a, gr, gn, v, am);
  }
  // the same with the deferred LayerNorm of the row (st.x = its factor ra): bi = c2 (the second constant set of the register epilogue is
  // EpiResid's; unused here)
  template <int N>
  TAMF_DEV void finish_ln(int a, int gr, int gn, float (&v)[N], const float (&bi)[N], const float (&)[N], float2 st, float& am) const {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = aff(v[j], st, bi[j]);
    act_store<N>(a, gr, gn, v, am);
  }
  template <int N>
  TAMF_DEV void lane_cols_ln(int gn, float (&bi)[N], float (&ci)[N]) const {
    g_loadn<N>(bias + gn, bi);
#pragma unroll
    for (int j = 0; j < N; ++j) ci[j] = 0.f;
  }
  // activation + operand store (the sum is complete)
  template <int N>
  TAMF_DEV void act_store(int a, int gr, int gn, float (&v)[N], float& am) const {
    if (a == ACT_SILU) {
      if constexpr (OutOp::PREC == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = silu_exact(v[j]);
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = silu_fast(v[j]);
      }
    } else if (a == ACT_GELU) {
      if constexpr (OutOp::PREC == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = gelu_erf(v[j]);
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = gelu_erf_fast(v[j]);
      }
    }
#ifdef TAMF_H_NT
    if (a == ACT_GELU) { OutOp::template store_rc<N, true>(out, (long)gr * ldo + gn, v, am); return; }  // (the FFN hidden activations: A/B build)
#endif
    OutOp::template store_rc<N>(out, (long)gr * ldo + gn, v, am);
  }
  // register form: a lane stores 16 bytes per instruction - 4 columns of a 4-byte output, 8 of a 16-bit plane
  static constexpr int LANE_CHUNK = OutOp::PREC == 0 ? 4 : 8;
  static constexpr bool TRANSPOSED = false;
  static constexpr int CHUNK_STORES = OutOp::SPLIT ? 2 : 1;  // global store instructions of one finish_act<LANE_CHUNK>
  template <int N>
  TAMF_DEV void lane_cols(int gn, float (&bi)[N]) const {
#pragma unroll
    for (int j = 0; j < N; ++j) bi[j] = 0.f;
    if (bias) g_loadn<N>(bias + gn, bi);
  }
};

// The same projection as two clip-tile launches (tamf_gemm_clip.h): EpiQK = its Q and K columns (register form only), EpiVt =
// its V columns with the MFMA operands exchanged, so that a lane ends up with runs of consecutive KEYS of one feature - what
// a V^T row stores.  Same operations per element as EpiQKV::run, i.e. the same bits.
template <class Op, bool LN = false>
struct EpiQK {
  const float* bias;  // [2d] (the Q and K parts of in_proj_bias; LN: of c2)
  typename Op::elem_t* qk;
  int d;
  float qscale;
  int act;  // (ACT_NONE; the register epilogue dispatches on it)
  EpiCtl ctl;
  LnStats ln{};
  static constexpr bool ROWSTATS = LN;
  static constexpr bool STAGE_AFF = true;
  static constexpr bool PREFETCH = false;
  static constexpr int LANE_CHUNK = Op::PREC == 0 ? 4 : 8;
  static constexpr bool TRANSPOSED = false;
  static constexpr int CHUNK_STORES = Op::SPLIT ? 2 : 1;
  template <int N>
  TAMF_DEV void finish_act(int, int gr, int gn, float (&v)[N], const float (&bi)[N], float& am) const {
    const float sc = (gn < d) ? qscale : 1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = fmaf(v[j], ctl.wscale, bi[j]) * sc;
    Op::template store_rc<N>(qk, (long)gr * (2 * d) + gn, v, am);
  }
  // (deferred LayerNorm of the row: st.x = its factor rstd ws - EpiQKV<Op, true>::proj, the same operations)
  template <int N>
  TAMF_DEV void finish_ln(int, int gr, int gn, float (&v)[N], const float (&bi)[N], const float (&)[N], float2 st, float& am) const {
    const float sc = (gn < d) ? qscale : 1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = aff(v[j], st, bi[j]) * sc;
    Op::template store_rc<N>(qk, (long)gr * (2 * d) + gn, v, am);
  }
  TAMF_DEV void flag(float am) const { Op::range_flag(am, ctl.status); }
  template <int N>
  TAMF_DEV void lane_cols(int gn, float (&bi)[N]) const { g_loadn<N>(bias + gn, bi); }
  template <int N>
  TAMF_DEV void lane_cols_ln(int gn, float (&bi)[N], float (&ci)[N]) const {
    g_loadn<N>(bias + gn, bi);
#pragma unroll
    for (int j = 0; j < N; ++j) ci[j] = 0.f;
  }
};
template <class Op, bool LN = false>
struct EpiVt {
  const float* bias;  // [d] (the V part of in_proj_bias; LN: of c2)
  typename Op::elem_t* vt;
  int H, hd, Skp;
  int act;
  EpiCtl ctl;
  LnStats ln{};
  static constexpr bool ROWSTATS = LN;  // (transposed: the staged row factors are indexed by the lane's KEYS, clip_store_vt)
  static constexpr bool STAGE_AFF = true;
  static constexpr bool PREFETCH = false;
  static constexpr int LANE_CHUNK = 4;  // (W rows staged in their natural order)
  static constexpr bool TRANSPOSED = true;
  static constexpr int CHUNK_STORES = Op::SPLIT ? 2 : 1;  // (of one store_keys)
  // feature eg (column of the V block) of clip b: N stored key positions from pos0 (N = 8: one 16-byte piece per 16-bit plane,
  // the keys 4g .. 4g+3 of two consecutive 16-key groups, vt_key_pos; N = 4: four consecutive keys, f32)
  template <int N>
  TAMF_DEV void store_keys(int b, int eg, int pos0, const float (&v)[N], float& am) const {
    const int h = eg / hd, e = eg % hd;
    Op::template store_rc<N>(vt, ((long)(b * H + h) * hd + e) * Skp + pos0, v, am);
  }
  TAMF_DEV void flag(float am) const { Op::range_flag(am, ctl.status); }
};

// in_proj: columns [0,d) = Q (scaled by qscale), [d,2d) = K -> row-major [M][2d]; [2d,3d) = V -> transposed
// per (clip, head): Vt[((b*H + h)*hd + e)][s], keys contiguous (what the P.V MFMA wants as its K axis).
// LN = true: deferred LayerNorm of the input rows (see EpiBiasAct): the weight is in_proj diag(gamma) C, bias = c2.
template <class Op, bool LN = false>
struct EpiQKV {
  const float* bias;  // [3d]
  typename Op::elem_t* qk;
  typename Op::elem_t* vt;
  int d, H, hd, Sp, Skp;
  float qscale;
  EpiCtl ctl;
  LnStats ln{};
  static constexpr bool ROWSTATS = LN;
  static constexpr bool STAGE_AFF = true;
  // C element of tile row `row` -> projected value: acc ws + bias, or the row's deferred LayerNorm applied on the way
  TAMF_DEV float proj(float acc, int row, float bb, float ws, const float2* rs) const {
    if constexpr (LN) return aff(acc, rs[row], bb);
    else return fmaf(acc, ws, bb);
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid, const float2* rs = nullptr) const {
    const float ws = ctl.wscale;
    if (n0 < 2 * d) {
      constexpr int VPR = BN / 8, RSTEP = NT / VPR;
      static_assert(NT % VPR == 0, "column group must be fixed per thread");
      const float sc = (n0 < d) ? qscale : 1.0f;
      const int col = (tid % VPR) * 8, gn = n0 + col;
      float b[8];
      g_load8(bias + gn, b);
      settle(b);
      float am = 0.f;
      // rows in batches of 4: the C-tile reads of a batch are requested together (one LDS latency per batch instead of one per row:
      // left to itself the loop is read - wait - convert - store, row by row)
      constexpr int NR = BM / RSTEP, RB = NR % 4 == 0 ? 4 : (NR % 2 == 0 ? 2 : 1);
      for (int r0 = 0; r0 < NR; r0 += RB) {
        float v[RB][8];
#pragma unroll
        for (int i = 0; i < RB; ++i) ct_load8(Ct, LDC, tid / VPR + (r0 + i) * RSTEP, col, v[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int gr = m0 + tid / VPR + (r0 + i) * RSTEP;
          if (gr < M) {
            if constexpr (LN) {
              const float2 ra = rs[tid / VPR + (r0 + i) * RSTEP];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[i][j] = aff(v[i][j], ra, b[j]) * sc;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[i][j] = fmaf(v[i][j], ws, b[j]) * sc;
            }
            Op::template store_rc<8>(qk, (long)gr * (2 * d) + gn, v[i], am);
          }
        }
      }
      Op::range_flag(am, ctl.status);
    } else if constexpr (Op::PREC == 0) {
      // f32: V^T rows keep the natural key order; one thread = 8 consecutive keys of one feature
      for (int it = tid; it < (BM / 8) * BN; it += NT) {
        const int col = it % BN, rg = it / BN;
        const int gr0 = m0 + rg * 8;
        if (gr0 >= M) continue;
        const int eg = n0 - 2 * d + col;
        const int h = eg / hd, e = eg % hd;
        const int b = gr0 / Sp, s0 = gr0 % Sp;
        const float bb = bias[n0 + col];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = proj(Ct[(rg * 8 + j) * LDC + col], rg * 8 + j, bb, ws, rs);
        Op::template store<8>(vt, ((long)(b * H + h) * hd + e) * Skp + s0, v);
      }
    } else {
      // 16-bit modes: V^T is key-permuted (vt_key_pos): inside a 32-key block the keys 4q .. 4q+3 of the two 16-key groups are
      // adjacent, i.e. 16-byte piece q of the block's 64-byte plane row holds keys {4q .. 4q+3, 16+4q .. 16+4q+3}.
      static_assert(BM % 32 == 0 && BN % 16 == 0 && NT % 64 == 0, "V^T epilogue tiling");
      const int lane = tid & 63, wv = tid >> 6;
      float am = 0.f;
      if (Sp % 16 == 0 && !Op::SPLIT) {
        // (bf16; measured on the QKV launch at B = 64 with tools/gemm_timeline.py: tile epilogues median 2.5 -> 1.8 us, p90 6.5 -> 4.0 us,
        //  launch 34.3 -> 32.0 us.  In the split modes the same scheme - a lane per hi or lo piece, the split computed twice - made the
        //  V tiles slower, p90 9.2 -> 12.2 us: they keep the 8-byte runs below.)
        // Clips are whole 16-key groups (T = 196: Sp = 208), so the tile's eight 16-row groups are 16-key groups of (at most two)
        // clips.  One lane = one 16-byte piece of one feature: lane = piece (q, and hi / lo plane in the split modes) + PPL * feature,
        // so 8 (4) neighbouring lanes write one whole 128-byte (64-byte) line and a wave-instruction 8 (16) complete lines.  The
        // two groups of a key block are consecutive row groups of the tile; a group whose partner lies in another tile (odd clips
        // start in the middle of a 32-row block) is stored as an 8-byte run.  The walk over the row groups is wave-uniform.
        // (the earlier form stored 8-byte runs only: the V tiles of the QKV launch took 9 us in their epilogue, Q / K tiles 3)
        constexpr int PPL = Op::SPLIT ? 8 : 4;  // 16-byte pieces per (feature, 32-key block): [hi 4 | lo 4] or 4
        constexpr int FPW = 64 / PPL;           // features per wave-iteration
        const int piece = lane % PPL, f_lo = lane / PPL, q = piece & 3;
        const int plane_off = (Op::SPLIT && piece >= 4) ? 64 : 0;
        const int ng = (M - m0 < BM ? M - m0 : BM) / 16;  // row groups of this tile that exist (M is a multiple of Sp)
        constexpr int NGRP = BN / FPW, NFG = (NGRP + NT / 64 - 1) / (NT / 64);  // feature groups of the tile / per wave
        float bbs[NFG];
#pragma unroll
        for (int i = 0; i < NFG; ++i) {
          const int fg = wv + i * (NT / 64);
          bbs[i] = bias[n0 + (fg < NGRP ? fg : 0) * FPW + f_lo];
        }
        settle(bbs);
#pragma unroll
        for (int i = 0; i < NFG; ++i) {
          const int fg = wv + i * (NT / 64);
          if (fg >= NGRP) break;
          const int col = fg * FPW + f_lo;
          const int eg = n0 - 2 * d + col;
          const int h = eg / hd, e = eg % hd;
          const float bb = bbs[i];
          for (int r16 = 0; r16 < ng;) {
            const int gr = m0 + r16 * 16;
            const int b = gr / Sp, k16 = (gr % Sp) / 16;  // clip and 16-key group
            const bool pair = !(k16 & 1) && r16 + 1 < ng && (k16 + 1) * 16 < Sp;
            const long row_idx = ((long)(b * H + h) * hd + e) * Skp + (k16 >> 1) * 32;  // first position of the key block
            char* pp = (char*)vt + Op::byte_off(row_idx) + plane_off + q * 16;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = proj(Ct[(r16 * 16 + 4 * q + j) * LDC + col], r16 * 16 + 4 * q + j, bb, ws, rs);
            if (pair) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[4 + j] = proj(Ct[(r16 * 16 + 16 + 4 * q + j) * LDC + col], r16 * 16 + 16 + 4 * q + j, bb, ws, rs);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[4 + j] = 0.f;
            }
            uint32_t wh[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              Op::split2(v[2 * j], v[2 * j + 1], wh[j], wl[j]);
              am = fmaxf(fmaxf(am, fabsf(v[2 * j])), fabsf(v[2 * j + 1]));
            }
            if (Op::SPLIT && piece >= 4) {
#pragma unroll
              for (int j = 0; j < 4; ++j) wh[j] = wl[j];
            }
            if (pair) {
              gst16(pp, wh[0], wh[1], wh[2], wh[3]);
              r16 += 2;
            } else {  // the group alone: keys 4q .. 4q+3 of group k16 & 1 are bytes [8 (k16 & 1), +8) of the piece
              gst8(pp + 8 * (k16 & 1), wh[0], wh[1]);
              r16 += 1;
            }
          }
        }
      } else {
        // any other Sp (a multiple of 8): one thread = one run of 4 keys of one feature; a wave = 8 features x the 8 runs of one
        // 32-row block, lanes ordered so that the 8 runs of a feature are adjacent (8 segments of 64 bytes per store instruction)
        constexpr int NB32 = BM / 32, WITER = NB32 * (BN / 8);  // wave-iterations of the tile
        const int e_lo = lane >> 3, u_lo = (lane >> 2) & 1, gq = lane & 3;
        for (int wi = wv; wi < WITER; wi += NT / 64) {
          const int col = (wi / NB32) * 8 + e_lo, row0 = (wi % NB32) * 32 + u_lo * 16 + gq * 4;
          const int gr0 = m0 + row0;
          if (gr0 >= M) continue;
          const int eg = n0 - 2 * d + col;
          const int h = eg / hd, e = eg % hd;
          const int b = gr0 / Sp, s0 = gr0 % Sp;
          const float bb = bias[n0 + col];
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = proj(Ct[(row0 + j) * LDC + col], row0 + j, bb, ws, rs);
          Op::template store_rc<4>(vt, ((long)(b * H + h) * hd + e) * Skp + vt_key_pos<Op>(s0), v, am);
        }
      }
      Op::range_flag(am, ctl.status);
    }
  }
};

// input_merge.2 (+ bias, nan_to_num, + positional row) scattered into the token rows of the sequence;
// also builds the timestep-embedding table.  GEMM row r = (b, tau) with b = r / Tdiv, tau = r % Tdiv goes
// to output row b*Sp + P + tau and gets pe[tau * pe_stride + n] added.
template <class Op>
struct EpiSeqRows {
  const float* bias;
  const float* pe;
  int pe_stride;
  float* xout;  // [rows][d] fp32 or null
  typename Op::elem_t* xop;  // operand planes or null
  int d, Tdiv, Sp, P;
  // the rows the encoder input needs besides the frame tokens, written by the tile that holds a clip's first frame (one launch
  // less per step than a separate kernel): prefix row 0 = timestep-embedding table row of the clip's current t (has_t), prefix
  // rows 1.. = the step-invariant tokens, pad rows [S, Sp) = 0.  pstatic == null: none (the timestep-table build).
  const float* pstatic;  // [B][P - has_t][d]
  const float* temb;     // [n_t][d]
  const int* tcur;
  int has_t, S;
  int t_off;             // steps since the counter was last written (position of this step inside its captured graph)
  EpiCtl ctl;
  static constexpr bool ROWSTATS = false;
  static constexpr bool PREFETCH = false;
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid) const {
    constexpr int VPR = BN / 8, RSTEP = NT / VPR;
    static_assert(NT % VPR == 0, "column group must be fixed per thread");
    const int col = (tid % VPR) * 8, gn = n0 + col;
    float bi[8];
    g_load8(bias + gn, bi);
    float am = 0.f;
    // every global load of the epilogue is issued (and waited for: settle) before its first store - vmcnt retires in order, so a
    // load issued behind stores waits for them
    constexpr int NR = BM / RSTEP;
    static_assert(BM % RSTEP == 0, "rows per thread");
    float pv[NR][8];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int gr = m0 + tid / VPR + i * RSTEP;
      gr = gr < M ? gr : M - 1;
      g_load8(pe + (long)(gr % Tdiv) * pe_stride + gn, pv[i]);
    }
    settle(bi);
#pragma unroll
    for (int i = 0; i < NR; ++i) settle(pv[i]);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = tid / VPR + i * RSTEP, gr = m0 + row;
      if (gr < M) {
        const int b = gr / Tdiv, tau = gr % Tdiv;
        const long orow = (long)b * Sp + P + tau;
        float v[8];
        ct_load8(Ct, LDC, row, col, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = nan_to_num(fmaf(v[j], ctl.wscale, bi[j])) + pv[i][j];
        if (xout) g_store8(xout + orow * d + gn, v);
        if (xop) Op::template store_rc<8>(xop, orow * d + gn, v, am);
      }
    }
    if (pstatic) {
      const int rows_per_clip = P + (Sp - S);
      const int m1 = m0 + BM < M ? m0 + BM : M;
      for (int b = (m0 + Tdiv - 1) / Tdiv; b * Tdiv < m1; ++b) {  // clips whose first frame row lies in this tile
        for (int it = tid; it < rows_per_clip * VPR; it += NT) {
          const int j = it / VPR, c = n0 + (it % VPR) * 8;
          float v[8];
          int s;
          if (j < P) {
            s = j;
            const float* src = (has_t && j == 0) ? temb + (long)(tcur[b] - t_off) * d : pstatic + ((long)b * (P - has_t) + (j - has_t)) * d;
            g_load8(src + c, v);
          } else {
            s = S + (j - P);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
          }
          const long o = ((long)b * Sp + s) * d + c;
          g_store8(xout + o, v);
          if (xop) Op::template store_rc<8>(xop, o, v, am);  // (null in f32: the operand matrix IS the fp32 state there)
        }
      }
    }
    Op::range_flag(am, ctl.status);
  }
};

// output_process.poseFinal (+ bias, nan_to_num) fused with either the x0 write-out in the reference layout
// (B, F, 1, T) or the DDPM reverse update of the frame-major sampler state.
enum { HEAD_X0 = 0, HEAD_DDPM = 1, HEAD_RESIDUAL = 2 };
template <class Op>
struct EpiHead {
  const float* bias;  // [XN] padded
  int mode;
  int F, T, Sp, P, XK;  // XK: row stride of the state (F padded)
  float* x0_out;        // HEAD_X0: (B, F, 1, T); HEAD_RESIDUAL: (B, T, F)
  const float* x_in;    // HEAD_RESIDUAL: (B, T, F)
  float* xs;            // HEAD_DDPM: state [B*T][XK]
  typename Op::elem_t* xs_op;
  const int* tcur;      // device: current timestep index of every clip (uniform inside the loop)
  const float* c1;
  const float* c2;
  const float* sigma;
  int n_steps;
  const LoopParams* __restrict__ lp;  // HEAD_DDPM only
  // Position of this step inside its captured graph: the device-side step counter `tcur` is written once per graph launch (by the
  // one tiny kernel at the graph's end), step g of the graph works at t = tcur - g.  No per-step counter kernel and no atomics:
  // a ticket scheme - the last workgroup of this launch to finish decrements the counter - cost the head launch 30 -> 37 us (208
  // same-address atomics from 8 XCDs).
  int t_off;
  EpiCtl ctl;
  // The last LayerNorm of the encoder, deferred into this GEMM (tamf_device.h): the weight is W_f diag(gamma) C, bias = c2; ln.part is
  // null in f32, whose rows arrive normalised - the row factor is then ws and the projection is fma(acc, ws, bias) bit for bit
  LnStats ln{};
  static constexpr bool ROWSTATS = true;
  static constexpr bool STAGE_AFF = true;
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid, const float2* rs) const { run_rows<BM, BN, NT>(Ct, LDC, m0, n0, M, tid, rs); }
  template <int BM, int BN, int NT>
  TAMF_DEV void run_rows(const float* Ct, int LDC, int m0, int n0, int M, int tid, const float2* rs) const {
    constexpr int VPR = BN / 8, RSTEP = NT / VPR;
    static_assert(NT % VPR == 0, "column group must be fixed per thread");
    const int col = (tid % VPR) * 8, gn = n0 + col;
    float bi[8];
    g_load8(bias + gn, bi);
    float am = 0.f;
    for (int row = tid / VPR; row < BM; row += RSTEP) {
      const int gr = m0 + row;
      if (gr >= M) break;
      const int b = gr / Sp, s = gr % Sp;
      if (s < P || s >= P + T) continue;
      const int tau = s - P;
      float v[8];
      ct_load8(Ct, LDC, row, col, v);
      {  // the head's projection itself: every mode below consumes v[j] = acc ws + bias (of the normalised row)
        const float2 ra = rs[row];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = aff(v[j], ra, bi[j]);
      }
      if (mode == HEAD_X0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = gn + j;
          if (c < F) x0_out[((long)b * F + c) * T + tau] = nan_to_num(v[j]);
        }
      } else if (mode == HEAD_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = gn + j;
          const long o = ((long)b * T + tau) * F + c;
          if (c < F) x0_out[o] = nan_to_num(x_in[o] + v[j]);
        }
      } else {
        const int ti = tcur[0] - t_off;
        const float k1 = c1[ti], k2 = c2[ti], sg = sigma[ti];
        const float* noise = lp->noise;
        float* dump = lp->dump;
        const long noise_draw_stride = lp->noise_draw_stride;
        const unsigned long long seed = lp->seed;
        const long long clip_base = lp->clip_base;
        const unsigned draw = (unsigned)(n_steps - ti);
        const long srow = ((long)b * T + tau) * XK;
        float xt[8], xn[8], ez[8];
        g_load8(xs + srow + gn, xt);
        if (ti != 0) {
          if (noise) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              ez[j] = (gn + j < F) ? noise[(long)draw * noise_draw_stride + ((long)b * F + gn + j) * T + tau] : 0.f;
          } else {  // 8 consecutive features of one frame = 2 Philox blocks
            float z0[4], z1[4];
            const unsigned blk = ((unsigned)tau * 128u + (unsigned)gn) >> 2;
            philox_normal4(seed, clip_base + b, draw, blk, z0);
            philox_normal4(seed, clip_base + b, draw, blk + 1, z1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ez[j] = z0[j];
              ez[4 + j] = z1[j];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = gn + j;
          if (c < F) {
            const float x0 = nan_to_num(v[j]);
            // mean = coef1*x0 + coef2*x_t ; sample = mean + [t!=0]*sigma*eps  (gaussian_diffusion.py:221-224,459)
            float r = __fadd_rn(__fmul_rn(k1, x0), __fmul_rn(k2, xt[j]));
            if (ti != 0) r = __fadd_rn(r, __fmul_rn(sg, ez[j]));
            xn[j] = r;
            if (dump) dump[(long)(draw - 1) * noise_draw_stride + ((long)b * F + c) * T + tau] = r;
          } else {
            xn[j] = 0.f;
          }
        }
        g_store8(xs + srow + gn, xn);
        if (xs_op) Op::template store_rc<8>(xs_op, srow + gn, xn, am);
      }
    }
    Op::range_flag(am, ctl.status);
  }
};

// plain fp32 store (test hook)
struct EpiStoreF32 {
  const float* bias;
  float* out;
  int ldo;
  int act;
  EpiCtl ctl;
  static constexpr bool ROWSTATS = false;
  static constexpr bool PREFETCH = false;
  struct Cols {
    float bi[8];
    // a (free) register use that makes the compiler wait for the loads HERE, once: left to the first use inside a row loop,
    // its s_waitcnt vmcnt(0) is repeated every iteration and then waits for the previous iteration's stores
    TAMF_DEV void settle() const {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(bi[j]));
    }
  };
  template <int BN, int NT>
  TAMF_DEV Cols cols(int n0, int tid) const {
    Cols c;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.bi[j] = 0.f;
    if (bias) g_load8(bias + n0 + (tid % (BN / 8)) * 8, c.bi);
    return c;
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid) const {
    run_c<BM, BN, NT>(Ct, LDC, m0, n0, M, tid, cols<BN, NT>(n0, tid));
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run_c(const float* Ct, int LDC, int m0, int n0, int M, int tid, const Cols& cc) const {
    constexpr int VPR = BN / 8, RSTEP = NT / VPR;
    static_assert(NT % VPR == 0, "column group must be fixed per thread");
    const int col = (tid % VPR) * 8, gn = n0 + col;
    for (int row = tid / VPR; row < BM; row += RSTEP) {
      const int gr = m0 + row;
      if (gr >= M) break;
      float v[8];
      float am = 0.f;
      ct_load8(Ct, LDC, row, col, v);
      finish_act<8>(act, gr, gn, v, cc.bi, am);
    }
  }
  TAMF_DEV void flag(float) const {}  // (fp32 output: nothing to check)
  template <int N>
  TAMF_DEV void finish_act(int a, int gr, int gn, float (&v)[N], const float (&bi)[N], float&) const {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = fmaf(v[j], ctl.wscale, bi[j]);
    if (a == ACT_SILU) {
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = silu_exact(v[j]);
    } else if (a == ACT_GELU) {
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = gelu_erf(v[j]);
    }
    float* p = out + (long)gr * ldo + gn;
#pragma unroll
    for (int j = 0; j < N; j += 4) gst16f(p + j, v[j], v[j + 1], v[j + 2], v[j + 3]);
  }
  static constexpr int LANE_CHUNK = 4;
  static constexpr bool TRANSPOSED = false;
  static constexpr int CHUNK_STORES = 1;
  template <int N>
  TAMF_DEV void lane_cols(int gn, float (&bi)[N]) const {
#pragma unroll
    for (int j = 0; j < N; ++j) bi[j] = 0.f;
    if (bias) g_loadn<N>(bias + gn, bi);
  }
};

// Residual add of a post-LN sublayer with the LayerNorm of its INPUT deferred (tamf_device.h, "Deferred LayerNorm"):
//   u_next[m][n] = ((u[m][n] - mean[m]) rstd[m] gamma[n] + bb[n]) + C[m][n] ws            bb = beta + bias of this GEMM
// read from and written back to the fp32 residual stream in place (every element by one lane), stored as the operand of the next
// GEMM, and summarised per 32-column block as (S_b, Q_b) for the LayerNorm that the consumers of u_next will apply.
// No LayerNorm in front (layer 0: the rows are the encoder's input): ln.part = null, gamma = ones, bb = bias.
template <class Op>
struct EpiResid {
  const float* bb;           // [d]
  const float* gamma;        // [d]
  float* x;                  // [M][d]
  typename Op::elem_t* xop;  // [M][d] operand planes
  int d;
  float2* part_out;          // [M][d / 32]
  int act;                   // ACT_NONE (the register epilogue dispatches on it)
  EpiCtl ctl;
  LnStats ln{};
  static constexpr bool ROWSTATS = true;
  static constexpr bool STAGE_AFF = false;  // the kernel stages (mean, rstd): the residual is normalised itself, not a product of it
  static constexpr bool PREFETCH = true;  // the register epilogue requests row tile mi + 1's residual piece ahead of row tile mi's stores
  static constexpr int LANE_CHUNK = 8;
  static constexpr bool TRANSPOSED = false;
  // fp32 row piece (2 x 16 bytes), operand piece(s) (f32: the residual stream IS the operand, xop = null), block statistics
  static constexpr int CHUNK_STORES = 2 + (Op::PREC == 0 ? 0 : Op::SPLIT ? 2 : 1) + 1;
  TAMF_DEV void flag(float am) const { Op::range_flag(am, ctl.status); }
  template <int N>
  TAMF_DEV void lane_cols_ln(int gn, float (&bi)[N], float (&ci)[N]) const {
    g_loadn<N>(bb + gn, bi);
    g_loadn<N>(gamma + gn, ci);
  }
  // one row piece of 8 columns from gn (a multiple of 8): v = the accumulators, u = the piece of the residual stream as it is.
  // GROUPS: the 4 lanes that hold the 32-column block are the 4 lane groups of the wave (register epilogue) or a quad (LDS walk).
  template <bool GROUPS>
  TAMF_DEV void finish_piece(int gr, int gn, float (&v)[8], const float (&u)[8], const float (&bi)[8], const float (&ci)[8], float2 st,
                             float& am) const {
    // (on pairs: v_pk_add / v_pk_mul / v_pk_fma_f32 - the same IEEE operations per element as the scalar form)
    tamf_f32x2 w[4];
    {
#pragma clang fp contract(off)
      const tamf_f32x2 mean2 = {st.x, st.x}, rstd2 = {st.y, st.y}, ws2 = {ctl.wscale, ctl.wscale};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const tamf_f32x2 uu = {u[2 * j], u[2 * j + 1]}, g2 = {ci[2 * j], ci[2 * j + 1]}, b2 = {bi[2 * j], bi[2 * j + 1]};
        const tamf_f32x2 a2 = {v[2 * j], v[2 * j + 1]};
        const tamf_f32x2 y = __builtin_elementwise_fma((uu - mean2) * rstd2, g2, b2);  // (the LayerNorm formula of every kernel here)
        w[j] = __builtin_elementwise_fma(a2, ws2, y);
        v[2 * j] = w[j].x;
        v[2 * j + 1] = w[j].y;
      }
    }
    const long o = (long)gr * d + gn;
    g_store8(x + o, v);
    if constexpr (Op::PREC != 0) Op::template store_rc<8>(xop, o, v, am);
    const float2 p = ln_block_partial<GROUPS>(w);
    if ((gn & 31) == 0) part_out[(long)gr * (d >> 5) + (gn >> 5)] = p;
  }
  template <int N>
  TAMF_DEV void finish_ln(int, int gr, int gn, float (&v)[N], const float (&bi)[N], const float (&ci)[N], float2 st, float& am) const {
    static_assert(N == 8, "row pieces of 8 columns");
    float u[8];
    g_load8(x + (long)gr * d + gn, u);
    finish_piece<true>(gr, gn, v, u, bi, ci, st, am);
  }
  template <int N>
  TAMF_DEV void prefetch(int gr, int gn, float (&u)[N]) const {
    static_assert(N == 8, "row pieces of 8 columns");
    g_load8(x + (long)gr * d + gn, u);
  }
  template <int N>
  TAMF_DEV void finish_pf(int gr, int gn, float (&v)[N], const float (&u)[N], const float (&bi)[N], const float (&ci)[N], float2 st, float& am) const {
    finish_piece<true>(gr, gn, v, u, bi, ci, st, am);
  }
  template <int BM, int BN, int NT>
  TAMF_DEV void run(const float* Ct, int LDC, int m0, int n0, int M, int tid, const float2* rs) const {
    constexpr int VPR = BN / 8, RSTEP = NT / VPR, NR = BM / RSTEP;
    static_assert(NT % VPR == 0 && BM % RSTEP == 0 && VPR % 4 == 0, "a quad of threads = one 32-column block of one row");
    const int col = (tid % VPR) * 8, gn = n0 + col;
    float bi[8], ci[8];
    g_load8(bb + gn, bi);
    g_load8(gamma + gn, ci);
    // every global load ahead of the first store (vmcnt retires in order)
    float u[NR][8];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int gr = m0 + tid / VPR + i * RSTEP;
      gr = gr < M ? gr : M - 1;
      g_load8(x + (long)gr * d + gn, u[i]);
    }
    settle(bi);
    settle(ci);
#pragma unroll
    for (int i = 0; i < NR; ++i) settle(u[i]);
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = tid / VPR + i * RSTEP, gr = m0 + row;
      if (gr < M) {  // (uniform over the quad: its four threads share the row)
        float v[8];
        ct_load8(Ct, LDC, row, col, v);
        finish_piece<false>(gr, gn, v, u[i], bi, ci, rs[row], am);
      }
    }
    Op::range_flag(am, ctl.status);
  }
};

// ------------------------------------------------------------------------------------------------
// The kernel
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void tamf_lds_void;
typedef const __attribute__((address_space(1))) void tamf_gbl_void;

// cache-policy bits of the LDS-DMA loads (compile-time experiment knobs; 0 = default policy)
#ifndef TAMF_GLDS_AUX_A
#define TAMF_GLDS_AUX_A 0
#endif
#ifndef TAMF_GLDS_AUX_W
#define TAMF_GLDS_AUX_W 0
#endif
template <int AUX>
TAMF_DEV void glds16(const char* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((tamf_gbl_void*)gsrc, (tamf_lds_void*)lds_wave_base, 16, 0, AUX);
}

// logical block id with XCD-contiguous chunks (bijective for any grid size; blocks b and b+8 share an XCD)
TAMF_DEV int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int i = bid >> 3;  // position inside this XCD's chunk, in dispatch order
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// waves per SIMD the kernel is built for (LDS admits 2 workgroups per CU for the 128 x 128 tiles, 1 for 64 x 512):
// telling the compiler keeps its occupancy heuristics from squeezing the register budget
template <int BM, int BN, int NWV>
struct GemmOcc {
  static constexpr int WG_PER_CU = (GemmSmem<BM, BN>::TOTAL > 80 * 1024) ? 1 : 2;
  static constexpr int WAVES_PER_SIMD = (WG_PER_CU * NWV / 4) > 0 ? (WG_PER_CU * NWV / 4) : 1;
};

#ifdef TAMF_TIMELINE
__device__ unsigned long long g_gemm_ts[8192 * 5];
#endif
// One BM x BN output tile at (m0, n0); `lb` is the tile's logical index (K rotation only), `bid` the hardware block id.
template <class Op, int BM, int BN, int WGM, int WGN, class Epi>
TAMF_DEV void gemm_tile(const GemmArgs<Op>& ga, const Epi& epi, const int m0, const int n0, const int lb, const int bid,
                        char* smem) {
  TAMF_TS(ts0);
  constexpr int BKB = GEMM_BKB;
  constexpr int NT = WGM * WGN * 64;
  constexpr int NWV = WGM * WGN;
  constexpr int CPR = BKB / 16;    // 16-byte chunks per tile row (8)
  constexpr int RPI = 1024 / BKB;  // tile rows per 1-KiB piece (8)
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int A_PIECES = BM / RPI, W_PIECES = BN / RPI;
  constexpr int A_PW = (A_PIECES + NWV - 1) / NWV, W_PW = (W_PIECES + NWV - 1) / NWV;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile");
  typedef GemmSmem<BM, BN> SM;
  constexpr int A_BYTES = BM * BKB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int M = ga.M;
  const int KT = (ga.K * Op::EB) / BKB;
  const char* Ab = (const char*)ga.A;
  const char* Wb = (const char*)ga.W;

  // per-piece source byte offsets of this lane (k-tile term added at issue time); piece q of the A (W) tile is
  // handled by wave q % NWV and covers tile rows [8q, 8q+8)
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
    w_off[i] = (unsigned)((long)(n0 + (row < BN ? row : BN - 1)) * ga.ldw * Op::EB + ((pch ^ swz_chunk<BKB>(row)) << 4));
  }

#define TAMF_ISSUE_ALL(kt_, s_)                                                                      \
  {                                                                                                  \
    _Pragma("unroll") for (int ii = 0; ii < A_PW; ++ii) {                                            \
      const int q_ = wave + ii * NWV;                                                                \
      if (A_PIECES % NWV == 0 || q_ < A_PIECES)                                                      \
        glds16<TAMF_GLDS_AUX_A>(Ab + a_off[ii] + (long)(kt_) * BKB, smem + (s_) * SM::STAGE + q_ * 1024);             \
    }                                                                                                \
    _Pragma("unroll") for (int ii = 0; ii < W_PW; ++ii) {                                            \
      const int q_ = wave + ii * NWV;                                                                \
      if (W_PIECES % NWV == 0 || q_ < W_PIECES)                                                      \
        glds16<TAMF_GLDS_AUX_W>(Wb + w_off[ii] + (long)(kt_) * BKB, smem + (s_) * SM::STAGE + A_BYTES + q_ * 1024);   \
    }                                                                                                \
  }

  // L2 prefetch by touch: waves 0..3 each issue ONE sparse dword load (L1-bypassing, sc1) per K tile that pulls the
  // lines of tile kt+pf into the XCD's L2, so the LDS-DMA of that tile later sees L2-hit instead of MALL/HBM latency;
  // no LDS is needed for this extra prefetch depth.  Rows touched by this workgroup: its own A rows (waves 0,1) and a
  // 1/8 slice of the W rows (waves 2,3) - the other workgroups sharing the XCD cover the other slices.
  const int pf = (ga.krot >> 8) & 0xF;
  const char* tbase = nullptr;
  {
    const int peer = (bid >> 3) & 7;
    if (wave < 2) {
      const int r = wave * 64 + lane;
      if (r < BM) {
        int gr = m0 + r;
        gr = gr < M ? gr : M - 1;
        tbase = Ab + (long)gr * ga.lda * Op::EB;
      }
    } else if (wave < 4) {
      constexpr int WSL = BN / 8;
      const int j = (wave - 2) * 64 + lane;
      if (j < WSL) tbase = Wb + (long)(n0 + peer * WSL + j) * ga.ldw * Op::EB;
    }
  }
  unsigned tsink = 0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane (lr, g) reads chunks g and 4+g of tile row lr (+16 per MFMA tile); the row swizzle
  // depends on lr only because tile rows are multiples of 16 apart
  const int sw = swz_chunk<BKB>(lr);
  const int c0 = ((g ^ sw) << 4), c1 = (((4 + g) ^ sw) << 4);
  const int a_frag = (wm0 + lr) * BKB;
  const int w_frag = A_BYTES + (wn0 + lr) * BKB;

  const int krs = ga.krot & 0xFF;
  const bool abl_noload = (TAMF_ABL(ga.krot) & 0x1000) != 0, abl_nocomp = (TAMF_ABL(ga.krot) & 0x2000) != 0;
  const int rot = krs ? (int)(((unsigned)lb * (unsigned)krs) % (unsigned)KT) : 0;
  TAMF_ISSUE_ALL(rot, 0)
  // deferred LayerNorm: (mean, rstd) of the tile's rows, staged behind the staging buffers while the first K tile is in flight
  float2* const rstat = (float2*)(smem + SM::STATS_OFF);
  if constexpr (Epi::ROWSTATS) ln_stage<NT, Epi::STAGE_AFF, (BM + NT / 4 - 1) / (NT / 4)>(epi.ln, epi.ctl.wscale, m0, BM, M, rstat, tid);
  __syncthreads();
  TAMF_TS(ts1);

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < KT) && !abl_noload;
    int ktn = kt + 1 + rot;
    ktn = ktn >= KT ? ktn - KT : ktn;
    const char* cb = smem + cur * SM::STAGE;
    unsigned tv = 0;
    if (pf && tbase && kt + pf < KT) {
      int ktp = kt + pf + rot;
      ktp = ktp >= KT ? ktp - KT : ktp;
      tv = __hip_atomic_load((const unsigned*)(tbase + (long)ktp * BKB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
      // all fragments of the tile are requested up front: LDS latency is paid once and the MFMAs stream behind
      // counted lgkmcnt waits
      int4 af[MI][2], wf[NI][2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        af[mi][0] = *(const int4*)(cb + a_frag + mi * 16 * BKB + c0);
        af[mi][1] = *(const int4*)(cb + a_frag + mi * 16 * BKB + c1);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        wf[ni][0] = *(const int4*)(cb + w_frag + ni * 16 * BKB + c0);
        wf[ni][1] = *(const int4*)(cb + w_frag + ni * 16 * BKB + c1);
      }
      // the next tile's pieces are issued AFTER this tile's fragment reads: hipcc places a vmcnt(0) in front of any
      // LDS read that follows an LDS-DMA in program order, which would serialise the DMA latency with the MFMAs;
      // issued here the pieces fly under the MFMAs and are waited for at the barrier below
      if (more) TAMF_ISSUE_ALL(ktn, cur ^ 1)
      if (!abl_nocomp) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) Op::mma(acc[mi][ni], wf[ni], af[mi]);  // D rows = n (4g+reg), cols = m (lr)
      }
      // the MFMAs are pure register operations, so the scheduler is free to sink them below the barrier's s_waitcnt vmcnt(0) -
      // which then waits for the next tile's LDS-DMA BEFORE the math that was meant to cover it (hipcc 7.2 does exactly that once
      // the block above is straight-line code: QKV 66 -> 80 us).  Nothing crosses this point.
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    tsink ^= tv;  // consumed after the barrier's vmcnt(0): keeps the touch load alive without an extra wait
  }
#undef TAMF_ISSUE_ALL

  TAMF_TS(ts2);
  // park the accumulators in the LDS C tile: lane (g, lr) holds C[m = lr][n = 4g .. 4g+3] of each 16x16 tile
  float* Ct = (float*)smem;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const f32x4 v = acc[mi][ni];
      *(float4*)(Ct + (wm0 + mi * 16 + lr) * SM::LDC + wn0 + ni * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
  __syncthreads();
  if (tsink == 0x9E3779B9u && ga.M < 0) Ct[0] = 1.0f;  // never true; keeps the touch loads from being optimised away
  if constexpr (Epi::ROWSTATS) epi.template run<BM, BN, NT>(Ct, SM::LDC, m0, n0, M, tid, rstat);
  else epi.template run<BM, BN, NT>(Ct, SM::LDC, m0, n0, M, tid);
#ifdef TAMF_TIMELINE
  if (tid == 0 && bid < 8192) {
    unsigned long long* o = g_gemm_ts + bid * 5;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = wall_clock64(); o[4] = tamf_hw_cu_id();
  }
#endif
}

// Workgroup -> tile mapping.  A launch has n_full whole tiles (a multiple of the chip's resident workgroup count, so
// they run as full rounds) and R left-over tiles; with SPLIT > 1 each left-over tile is processed as SPLIT column
// slices of BN / SPLIT by SPLIT workgroups, so that the last, partial round is a round of short workgroups on every
// CU instead of a round of full-length workgroups on a fraction of them.  Slices are dispatched last (highest block ids).
template <class Op, int BM, int BN, int WGM, int WGN, class Epi, int SPLIT>
__global__ __launch_bounds__(WGM* WGN * 64, (GemmOcc<BM, BN, WGM * WGN>::WAVES_PER_SIMD)) void gemm_kernel(
    const GemmArgs<Op> ga, const Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntn = ga.N / BN;
  const int bid = blockIdx.x;
  if (SPLIT == 1 && ga.n_tiles > 0) {
    const int G = gridDim.x;
    const int ntm = ga.n_tiles / ntn;
#ifndef TAMF_PERSIST_XCD42  // (A/B builds: 1 = on.  Measured in round 4, profiles/r04/qkv_xcd42_c11.txt: FETCH 119 -> 95.5 MB per QKV launch, but the
#define TAMF_PERSIST_XCD42 0  //  launch 61.8 - 63.2 -> 63.2 - 63.5 us (f16x3), 29.6 -> 30.8 us (bf16), whole step +0.4 %: off)
#endif
    if (TAMF_PERSIST_XCD42 && (ga.krot & 0x10000) && (G & 7) == 0 && (ntm & 3) == 0 && (ntn & 1) == 0 && ntm * ntn == ga.n_tiles) {
      // 4 x 2 arrangement of the XCDs over the tile grid, persistent form (round 4 experiment; QKV: 104 x 12 tiles on 512 workgroups):
      // XCD x owns the row panels p = rq (mod 4), rq = x / 2, and the column half x % 2, and keeps them over all its rounds - its L2
      // then holds HALF of W (1.55 MB of the 4 MB L2; in the round-robin order every L2 pulls all 3.1 MB of W again in each of the
      // 2.44 rounds, its output stores having evicted them: that, not a re-fetched A panel, is the 108.7 MB the round-3 counters
      // showed) and an A panel is fetched by the two XCDs of its row quarter.  Same tiles, same K order per tile: the same bits.
      const int x = bid & 7, slot = bid >> 3, S = G >> 3;  // S workgroup slots per XCD
      const int rq = x >> 1, ch = x & 1, C = ntn >> 1, nloc = (ntm >> 2) * C;  // tiles of this XCD
      for (int j = slot; j < nloc; j += S) {
        const int mb = 4 * (j / C) + rq, nbk = ch * C + j % C;
        gemm_tile<Op, BM, BN, WGM, WGN, Epi>(ga, epi, mb * BM, nbk * BN, mb * ntn + nbk, bid, smem);
        __syncthreads();
      }
      return;
    }
    for (int base = 0; base < ga.n_tiles; base += G) {
      const int cnt = ga.n_tiles - base < G ? ga.n_tiles - base : G;
      if (bid < cnt) {
        const int lb = base + xcd_remap(bid, cnt);
        gemm_tile<Op, BM, BN, WGM, WGN, Epi>(ga, epi, (lb / ntn) * BM, (lb % ntn) * BN, lb, bid, smem);
        __syncthreads();  // the C tile (aliasing the stages) has been consumed before the next tile's first DMA
      }
    }
    return;
  }
  if (SPLIT == 1 || bid < ga.n_full) {
    const int nb = SPLIT == 1 ? (int)gridDim.x : ga.n_full;
    const int nrb = nb / ntn;  // whole row blocks covered by these workgroups
    if ((ga.krot & 0x10000) && nb == nrb * ntn && (nrb & 3) == 0 && (ntn & 1) == 0) {
      // 4 x 2 arrangement of the XCDs over the tile grid (bit 16): an XCD owns a quarter of the row blocks and half of the
      // column tiles, walked row-major - its concurrent tiles share half of W (instead of all of it) and their A panels
      const int x = bid & 7, i = bid >> 3, R = nrb >> 2, C = ntn >> 1;
      const int mb = (x >> 1) * R + i / C, nbk = (x & 1) * C + i % C;
      gemm_tile<Op, BM, BN, WGM, WGN, Epi>(ga, epi, mb * BM, nbk * BN, mb * ntn + nbk, bid, smem);
      return;
    }
    const int lb = xcd_remap(bid, nb);
    gemm_tile<Op, BM, BN, WGM, WGN, Epi>(ga, epi, (lb / ntn) * BM, (lb % ntn) * BN, lb, bid, smem);
  } else {
    constexpr int BNS = BN / SPLIT;
    const int j = xcd_remap(bid - ga.n_full, (int)gridDim.x - ga.n_full);
    const int big = ga.n_full + j / SPLIT, sub = j % SPLIT;
    gemm_tile<Op, BM, BNS, WGM, WGN, Epi>(ga, epi, (big / ntn) * BM, (big % ntn) * BN + sub * BNS, big, bid, smem);
  }
}
