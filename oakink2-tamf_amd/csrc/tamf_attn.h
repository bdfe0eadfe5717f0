// Multi-head self-attention over one clip's S = T + prefix tokens (full, unmasked softmax(QK^T)V per
// (clip, head); interaction_segment_mdm.py:63-70,171 -> nn.MultiheadAttention).
//
// Workgroup = NW <= 16 waves = NW tiles of 16 queries of one (clip, head); a T = 196 clip (13 tiles) is one workgroup,
// so its K/V are streamed once (34 us vs 36 us with two workgroups).  Keys/values are streamed through LDS in
// blocks of 32 keys (LDS-DMA, double-buffered: block kb+1 is in flight while block kb is multiplied, see AttnBlock)
// with an online softmax, so LDS and registers are bounded for every arithmetic mode.
// Both products run "swapped" so that the softmax axis (keys) lies along the MFMA row index and each lane owns
// one query column:
//     S^T[key][query] = K . Q^T      A operand = K rows from LDS (swizzled, ds_read_b128), B operand = Q (registers)
//     O^T[e][query]  += V^T . P^T    A operand = V^T rows from LDS (keys contiguous),     B operand = P (registers,
//                                    straight from the S^T accumulator layout - no cross-lane movement)
// Q arrives pre-scaled by log2(e)/sqrt(hd) (QKV epilogue), so probabilities are exp2(s - max).
// Operand rows are consumed in 128-byte groups exactly as in the GEMM (tamf_device.h "Operand traits").
#pragma once
#include "tamf_device.h"

template <class Op>
struct AttnArgs {
  const typename Op::elem_t* qk;  // [B*Sp][2d]  (Q | K)
  const typename Op::elem_t* vt;  // [B*H*hd][Skp]
  typename Op::elem_t* out;       // [B*Sp][d]
  int S, Sp, Skp, d, H;
  int abl;  // kernel-benchmark ablations of attn_res_kernel (-DTAMF_BENCH builds only, TAMF_ABL; tools/attn_bench.py): 1 = no LDS-DMA,
            // 2 = no MFMAs, 4 = no fragment reads, 8 = no exp2 / hi-lo split, 16 = no output store
  // attn_res_kernel, set by launch_attn: the clip's LAST query tile is computed by four waves, a quarter of the key blocks each.
  // nwq = query tiles per workgroup (the other tiles; waves 0 .. nwq-1 own them), hw = the wave indices of the four key-split waves in
  // the LAST workgroup of the pair (4 bits each), chosen on the SIMDs that carry the fewest tiles there
  int ksplit, nwq, hw;
};

template <class Op, int HD>
struct AttnCfg {
  static constexpr int EB = Op::EB;
  static constexpr int KROWB = HD * EB;   // bytes per K row (one head)
  static constexpr int KG = KROWB / 128;  // 128-byte groups per K row
  static constexpr int VROWB = 32 * EB;   // bytes of one V^T row block (32 keys): 64 (bf16) / 128 (f32, bf16x3)
  // V^T rows are stored unpadded (LDS-DMA writes linearly) with their 16-byte chunks XOR-swizzled by the row:
  // 128-byte rows like a GEMM tile row ((e >> 1) & 7), 64-byte rows by swz_chunk<64> - conflict-free for the
  // ds_read_b128 fragment reads (tools/lds_bank_sim.py)
  static constexpr int VSTR = VROWB;
  static constexpr int K_BYTES = 32 * KROWB;
  static constexpr int V_BYTES = HD * VSTR;
  static constexpr int STAGE = K_BYTES + V_BYTES;
  static constexpr int SMEM = 2 * STAGE;
  static_assert(KROWB % 128 == 0, "head slice must be whole 128-byte groups");
};

// One block of 32 keys for one wave: issue block kb+1 into `nxt`, then S^T, online softmax and O^T += V^T P^T on
// the block in `cur`.  `cur` and `nxt` are the two LDS stages and never overlap; declaring them __restrict__ on this
// (inlined) helper is what lets the LDS-DMA stay in flight: without the alias scopes hipcc puts an s_waitcnt vmcnt(0)
// in front of the first LDS read that follows an LDS-DMA in program order, which serialises the next block's latency
// with this block's math.  The landed data is published by the caller's vmcnt(0) + barrier.
template <class Op, int HD>
struct AttnBlock {
  typedef AttnCfg<Op, HD> C;
  static constexpr int EB = Op::EB, KG = C::KG, NT16 = HD / 16;
  static constexpr int KCH = C::KROWB / 16;  // chunks per K row
  static constexpr int VCH = C::VROWB / 16;  // chunks per V^T row block
  // chunk swizzles (involutions): K rows of 128 bytes use the GEMM tile swizzle, longer K rows XOR the low 4 chunk
  // bits with the row; V^T rows see the comment at AttnCfg::VSTR
  static TAMF_DEV int kswz(int ch, int row) {
    if constexpr (C::KROWB >= 256) return (ch & ~15) | ((ch ^ row) & 15);
    else return ch ^ ((row >> 1) & 7);
  }
  static TAMF_DEV int vswz(int ch, int e) {
    if constexpr (C::VROWB == 128) return ch ^ ((e >> 1) & 7);
    else return ch ^ swz_chunk<64>(e);
  }
  // LDS-DMA pieces (1 KiB = 64 lanes x 16 B, linear in LDS; the swizzle goes on each lane's SOURCE chunk)
  static constexpr int K_RPP = 1024 / C::KROWB, K_PIECES = C::K_BYTES / 1024;  // K rows per piece
  static constexpr int V_RPP = 1024 / C::VROWB, V_PIECES = C::V_BYTES / 1024;

  static TAMF_DEV void issue(char* __restrict__ st, const char* kbase, const char* vbase, int kb, int Sp, int d, int Skp,
                             int wave, int nw, int lane) {
    for (int q = wave; q < K_PIECES; q += nw) {
      const int r = q * K_RPP + lane / KCH, pc = lane % KCH;
      int key = kb * 32 + r;
      key = key < Sp ? key : Sp - 1;  // rows past the clip are clamped; they are masked in run()
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (long)key * (2 * d) * EB + kswz(pc, r) * 16),
                                       (__attribute__((address_space(3))) void*)(st + q * 1024), 16, 0, 0);
    }
    for (int q = wave; q < V_PIECES; q += nw) {
      const int e = q * V_RPP + lane / VCH, pc = lane % VCH;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((long)e * Skp + kb * 32) * EB + vswz(pc, e) * 16),
                                       (__attribute__((address_space(3))) void*)(st + C::K_BYTES + q * 1024), 16, 0, 0);
    }
  }

  static TAMF_DEV void run(const char* __restrict__ cur, char* __restrict__ nxt, const char* kbase, const char* vbase,
                           int kb, int nkb, int S, int Sp, int d, int Skp, int wave, int nw, int lane, bool active,
                           const int4 (&qf)[KG][2], f32x4 (&o)[NT16], float& m_run, float& l_run) {
    const int lr = lane & 15, g = lane >> 4;
    const char* Ks = cur;
    const char* Vs = cur + C::K_BYTES;
    if (kb + 1 < nkb) issue(nxt, kbase, vbase, kb + 1, Sp, d, Skp, wave, nw, lane);
    if (!active) return;

    // ---- S^T tiles: keys 16t + 4g + reg, query lr
    f32x4 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = t * 16 + lr;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        int4 kf[2];
        kf[0] = *(const int4*)(Ks + row * C::KROWB + kswz(kg * 8 + g, row) * 16);
        kf[1] = *(const int4*)(Ks + row * C::KROWB + kswz(kg * 8 + 4 + g, row) * 16);
        Op::mma(st[t], kf, qf[kg]);
      }
    }
    // ---- mask + online softmax (per query = per lane column; the 4 lane groups hold disjoint keys)
    if (kb == nkb - 1) {  // only the last block can contain keys >= S
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb * 32 + t * 16 + 4 * g + r >= S) st[t][r] = -1e30f;
    }
    float bm = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])),
                     fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
    bm = groups_reduce<RedMax>(bm);
    // Rescale on demand: the reference point m_run of a query only has to keep exp2(s - m_run) in range, it need not be
    // the exact running maximum (softmax is invariant to it, the final 1 / l removes it).  The accumulators are rescaled
    // only when some query of this wave sees its block maximum exceed m_run by more than 2^8 (always in the first block,
    // rarely afterwards): the usual case costs no exp2 / 32 multiplies for alpha.  Wave-uniform decision.
    constexpr float RESCALE_LOG2 = 8.0f;
    if (__builtin_amdgcn_ballot_w64(bm > m_run + RESCALE_LOG2) != 0ull) {
      const float m_new = fmaxf(m_run, bm);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) o[nt] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[t][r] = __builtin_amdgcn_exp2f(st[t][r] - m_run);
        ps += st[t][r];
      }
    ps = groups_reduce<RedSum>(ps);
    l_run += ps;

    // ---- O^T += V^T . P^T
    if constexpr (Op::PREC == 0) {
      // f32: k-slot of lane group g in MFMA (t, r) is key 16t + 4g + r
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const int e = nt * 16 + lr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int4 vf = *(const int4*)(Vs + e * C::VSTR + vswz(t * 4 + g, e) * 16);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.x), st[t][0], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.y), st[t][1], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.z), st[t][2], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.w), st[t][3], o[nt], 0, 0, 0);
        }
      }
    } else {
      // bf16: B fragment element j of lane group g is P[key 4g + j] (j < 4) / P[key 16 + 4g + j - 4] (j >= 4);
      // V^T is stored with exactly that key order, so its A fragment is one 16-byte chunk per plane.
      uint32_t wh[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p0 = st[(2 * j) >> 2][(2 * j) & 3], p1 = st[(2 * j + 1) >> 2][(2 * j + 1) & 3];
        Op::split2(p0, p1, wh[j], wl[j]);
      }
      const int4 ph = make_int4((int)wh[0], (int)wh[1], (int)wh[2], (int)wh[3]);
      const int4 pl = make_int4((int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]);
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        // V^T is stored key-permuted (vt_key_pos): the fragment of lane group g is chunk g (hi) / 4 + g (lo) of row e
        const int e = nt * 16 + lr;
        const int4 vh = *(const int4*)(Vs + e * C::VSTR + vswz(g, e) * 16);
        if constexpr (Op::SPLIT) {
          const int4 vl = *(const int4*)(Vs + e * C::VSTR + vswz(4 + g, e) * 16);
          o[nt] = Op::mfma1(vl, ph, o[nt]);
          o[nt] = Op::mfma1(vh, pl, o[nt]);
        }
        o[nt] = Op::mfma1(vh, ph, o[nt]);
      }
    }
  }
};

#ifdef TAMF_TIMELINE  // debug build: per-workgroup wall-clock stamps (tools/attn_timeline.py)
__device__ unsigned long long g_attn_ts[8192 * 4];
#endif

template <class Op, int HD>
__global__ __launch_bounds__(1024) void attn_kernel(const AttnArgs<Op> aa) {
  TAMF_TS(ts0);
  typedef AttnCfg<Op, HD> C;
  typedef AttnBlock<Op, HD> BLK;
  constexpr int EB = Op::EB, KG = C::KG;
  constexpr int NT16 = HD / 16;  // output tiles along e
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, nw = nthr >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / aa.H, h = bh % aa.H;
  const int q0 = (blockIdx.x * nw + wave) * 16;
  const bool active = q0 < aa.Sp;
  const int S = aa.S, Sp = aa.Sp, d = aa.d;
  const long row_base = (long)b * Sp;

  // Q fragments (B operand): lane (g, lr) -> query q0+lr, fragments g and 4+g of each 128-byte group of its head slice
  int4 qf[KG][2];
  {
    int q = q0 + lr;
    q = q < Sp ? q : Sp - 1;
    const char* qb = (const char*)aa.qk + ((row_base + q) * (2 * d) + h * HD) * EB + g * 16;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      qf[kg][0] = *(const int4*)(qb + kg * 128);
      qf[kg][1] = *(const int4*)(qb + kg * 128 + 64);
    }
  }

  f32x4 o[NT16];
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  const int nkb = (S + 31) / 32;
  const char* kbase = (const char*)aa.qk + (row_base * (2 * d) + d + h * HD) * EB;
  const char* vbase = (const char*)aa.vt + ((long)bh * HD) * aa.Skp * EB;

  BLK::issue(smem, kbase, vbase, 0, Sp, d, aa.Skp, wave, nw, lane);
  __syncthreads();
  TAMF_TS(ts1);
  for (int kb = 0; kb < nkb; ++kb) {
    BLK::run(smem + (kb & 1) * C::STAGE, smem + ((kb + 1) & 1) * C::STAGE, kbase, vbase, kb, nkb, S, Sp, d, aa.Skp, wave, nw,
             lane, active, qf, o, m_run, l_run);
    __syncthreads();  // block kb+1 has landed (vmcnt(0) + barrier) and everyone is done reading block kb
  }
#ifdef TAMF_TIMELINE
  if (tid == 0) {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 8192) {
      g_attn_ts[wg * 4 + 0] = ts0;
      g_attn_ts[wg * 4 + 1] = ts1;
      g_attn_ts[wg * 4 + 2] = wall_clock64();
      g_attn_ts[wg * 4 + 3] = tamf_hw_cu_id();
    }
  }
#endif

  if (!active) return;
  const int q = q0 + lr;
  if (q >= Sp) return;
  const float inv = 1.0f / l_run;
  // lane (g, lr): query q, e = 16 nt + 4g + reg -> 4 consecutive elements per tile
  // (outputs are convex combinations of V rows, which the QKV epilogue range-checks: plain stores)
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) {
    float v[4] = {o[nt][0] * inv, o[nt][1] * inv, o[nt][2] * inv, o[nt][3] * inv};
    Op::template store<4>(aa.out, (row_base + q) * d + h * HD + nt * 16 + 4 * g, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Resident-K attention for sequences of up to 224 keys (every shape the MF-MDM launchers produce:
// T <= 196 -> S <= 201).
//
// Why.  The streaming kernel above keeps ONE 32-key block (16 / 32 KB) in flight per CU and synchronises its 13 waves 7 times;
// with K / V^T coming out of the Infinity Cache (the QKV GEMM has just written them) a block takes 2 - 4 us to arrive and only
// 0.3 (bf16) - 1.3 us (split modes) to multiply: rocprofv3 PMC, profiles/r02/pmc_attention_*: matrix pipe busy 14 - 29 %,
// SQ_WAIT_ANY 43 %, no bank conflicts, 3.4 TB/s.  It is bound by exposed load latency, not by MFMA, LDS or bandwidth.
//
// Here all of K of the (clip, head) is requested in ONE burst (53 KB bf16, 106 KB split modes: >100 LDS-DMA pieces in flight),
// and as many 32-key blocks of V^T as fit next to it (all 7 in bf16 and at hd = 64; 3 at hd = 128 in the split modes) in a
// second burst that flies under the S^T = K Q^T products.  Every wave computes the scores of its 16 queries against ALL keys
// (14 accumulator tiles), so the softmax is the exact two-pass one of the reference (max over all keys, then exp2 / sum) - no
// running maximum, no rescale - and the waves meet at three barriers in total (K landed; V^T landed and K released; and, when
// V^T did not fit beside K, its remaining blocks, which are fetched into K's space under the first P V products).  Between
// the barriers the waves drift apart, so one wave's MFMAs cover another's exp2 / hi-lo splits / LDS reads.
// Same operand layouts, swizzles, MFMA operand order and output as the streaming kernel; per query the result depends only on
// that query's row of Q and the clip's K / V (batch- and split-invariant).
// ------------------------------------------------------------------------------------------------------------------------
// logical index with XCD-contiguous chunks (the bijection of tamf_gemm.h's xcd_remap: blocks b and b + 8 share an XCD)
TAMF_DEV int attn_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

constexpr int attn_res_maxw(int nkb) { return nkb <= 4 ? 8 : 16; }  // waves per workgroup of attn_res_kernel<.., nkb> (AttnRes::MAXW)
template <class Op, int HD, int NKB_>
struct AttnRes {
  typedef AttnCfg<Op, HD> C;
  typedef AttnBlock<Op, HD> BLK;
  static constexpr int EB = Op::EB, KG = C::KG, NT16 = HD / 16;
  static constexpr int NKB = NKB_, NKT = 2 * NKB;        // key blocks / key tiles held in registers (S <= 32 NKB)
  // waves per workgroup the kernel is built for: a clip of up to NKT key tiles has at most NKT query tiles (one wave each); the
  // short-clip form (NKB = 4: S <= 128, every clip of up to 123 frames) therefore never runs more than 8 waves, i.e. 2 per SIMD and
  // 256 registers per lane - with __launch_bounds__(1024) it was held to 128 and spilled 880 - 1 048 bytes per lane (round 4 verdict)
  static constexpr int MAXW = attn_res_maxw(NKB_);
  // shortest sequence this instantiation is launched for (launch_attn takes the smallest NKB that fits: 4 up to 128 keys, 6 up to
  // 192, 7 up to 224); key tiles entirely below it are never masked
  static constexpr int S_MIN = NKB_ <= 4 ? 1 : NKB_ <= 6 ? 129 : 193;
  static constexpr int LDS_MAX = 160 * 1024;
  static constexpr bool TWO = Op::SPLIT || Op::PREC == 0;  // two 16-byte V^T fragments per lane and feature tile (hi | lo, or f32's two key groups)

  // LDS map: [K: Sp rows x KROWB][V^T blocks 0 .. nv1) x V_BYTES]; blocks nv1 .. nkb) later overwrite K from offset 0
  static int k_bytes(int Sp) { return Sp * C::KROWB; }
  static int nv1(int Sp, int nkb) {
    const int fit = (LDS_MAX - k_bytes(Sp)) / C::V_BYTES;
    return fit < nkb ? fit : nkb;
  }
  static bool fits(int S, int Sp) {
    const int nkb = (S + 31) / 32;
    if (nkb > NKB || S < S_MIN || Sp > NKT * 16 || (k_bytes(Sp) % 1024) != 0) return false;
    const int n1 = nv1(Sp, nkb);
    return n1 >= 1 && (nkb - n1) * C::V_BYTES <= k_bytes(Sp);
  }
  // (at least NKT key tiles of K rows: the straight-line score pass reads the rows of all NKT tiles - the ones past the clip hold
  //  V^T bytes or nothing, and are masked - and must stay inside the workgroup's allocation)
  static int smem(int S, int Sp) {
    const int need = k_bytes(Sp) + nv1(Sp, (S + 31) / 32) * C::V_BYTES, rows = NKT * 16 * C::KROWB;
    return need > rows ? need : rows;
  }

  // key split of the last query tile (attn_res_kernel): bytes of the partial area (3 foreign partials per 128-byte row group + m / l),
  // and how much the allocation grows for nwq tile-owning waves - or -1 when it does not fit
  static constexpr int TPC_ = EB == 4 ? 2 : 4, NCH_ = (NT16 + TPC_ - 1) / TPC_, TPCE_ = NT16 < TPC_ ? NT16 : TPC_;
  static constexpr int KSPLIT_BYTES = NCH_ * 3 * TPCE_ * 1024 + 512;
  static int ksplit_extra(int S, int Sp, int nwq) {
    const int nkb = (S + 31) / 32, n1 = nv1(Sp, nkb), base = smem(S, Sp);
    int extra;
    if (n1 < nkb) {  // inside the first V^T area, behind the nwq output slots of 2 KiB; whatever sticks out is allocated on top
      const int end = k_bytes(Sp) + nwq * 2048 + KSPLIT_BYTES;
      extra = end > base ? end - base : 0;
    } else {
      extra = KSPLIT_BYTES;
    }
    return base + extra <= LDS_MAX ? extra : -1;
  }

  // K rows [r0, r1) (multiples of K_RPP): piece q covers rows [q K_RPP, (q + 1) K_RPP)
  static TAMF_DEV void issue_k(char* Ks, const char* kbase, int r0, int r1, int d, int wave, int nw, int lane) {
    const int q0 = r0 / BLK::K_RPP, np = r1 / BLK::K_RPP;
    for (int q = q0 + wave; q < np; q += nw) {
      const int r = q * BLK::K_RPP + lane / BLK::KCH, pc = lane % BLK::KCH;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (long)r * (2 * d) * EB + BLK::kswz(pc, r) * 16),
                                       (__attribute__((address_space(3))) void*)(Ks + q * 1024), 16, 0, 0);
    }
  }
  // V^T blocks [kb0, kb1) into dst (block kb at dst + (kb - kb0) * V_BYTES)
  static TAMF_DEV void issue_v(char* dst, const char* vbase, int kb0, int kb1, int Skp, int wave, int nw, int lane) {
    const int np = (kb1 - kb0) * BLK::V_PIECES;
    for (int q = wave; q < np; q += nw) {
      const int kb = kb0 + q / BLK::V_PIECES, qq = q % BLK::V_PIECES;
      const int e = qq * BLK::V_RPP + lane / BLK::VCH, pc = lane % BLK::VCH;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((long)e * Skp + kb * 32) * EB + BLK::vswz(pc, e) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    }
  }

  // K fragment i = (key tile i / KG, 128-byte group i % KG) of lane (lr, g): row 16 kt + lr, chunks 8 kg + g and 8 kg + 4 + g.  Both
  // swizzles depend on the row only through lr (rows of a lane are 16 apart), so the lane's byte offsets inside a key tile are
  // computed once (koff) and the tile term is a compile-time constant (an immediate offset of the ds_read)
  struct KOff { int o[KG][2]; };
  static TAMF_DEV KOff koff(int lr, int g) {
    KOff k;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      k.o[kg][0] = lr * C::KROWB + BLK::kswz(kg * 8 + g, lr) * 16;
      k.o[kg][1] = lr * C::KROWB + BLK::kswz(kg * 8 + 4 + g, lr) * 16;
    }
    return k;
  }
  static TAMF_DEV void kfrag(const char* Ks, int i, const KOff& ko, int4 (&kf)[2]) {
    const int kt = i / KG, kg = i % KG;
    kf[0] = *(const int4*)(Ks + kt * (16 * C::KROWB) + ko.o[kg][0]);
    kf[1] = *(const int4*)(Ks + kt * (16 * C::KROWB) + ko.o[kg][1]);
  }
  // phase A: V^T blocks [0, n1) are requested into Vs, then S^T = K Q^T for ALL NKT key tiles, straight-line: tiles past the
  // clip read rows of the V^T area (any bits) and are overwritten by the key mask.  The K fragments are requested two
  // (tile, group) steps ahead of the MFMAs that consume them (a wave alone would otherwise expose one LDS latency per step).
  // Ks / Vs never overlap: the __restrict__ qualifiers keep hipcc from waiting for the V^T requests before the first K
  // fragment read (as AttnBlock::run).
  // The FIRST call (T0 = 0) also requests what the later phases need - the second half of K into K2 and the V^T blocks [0, n1) into
  // Vs - before its MFMAs; K1 (read here), K2 and Vs never overlap, which the __restrict__ qualifiers tell the compiler.
  template <int T0, int T1>
  static TAMF_DEV void scores(const char* __restrict__ Ks, char* __restrict__ K2, char* __restrict__ Vs, const char* kbase,
                              const char* vbase, int k2r0, int k2r1, int d, int n1, int Skp, int wave, int nw, int lane,
                              const int4 (&qf)[KG][2], f32x4 (&st)[NKT], int abl, unsigned mask = ~0u) {
    const int lr = lane & 15, g = lane >> 4;
    if (T0 == 0 && !(abl & 1)) {
      if (k2r1 > k2r0) issue_k(K2 - (long)k2r0 * C::KROWB, kbase, k2r0, k2r1, d, wave, nw, lane);
      issue_v(Vs, vbase, 0, n1, Skp, wave, nw, lane);
    }
    constexpr int F0 = T0 * KG, NF = T1 * KG, LA = 2;
    const KOff ko = koff(lr, g);
    int4 kf[LA + 1][2];
#pragma unroll
    for (int i = 0; i <= LA; ++i) kf[i][0] = kf[i][1] = qf[0][0];  // (defined contents for the no-read ablation)
    // `mask`: bit kb = this wave multiplies key block kb (wave-uniform; all ones but for the key-split waves of the last query tile);
    // a skipped tile keeps the -1e30 the caller put there
    auto on = [&](int i) { return ((mask >> (i / KG / 2)) & 1u) != 0; };
#pragma unroll
    for (int i = F0; i < F0 + LA; ++i)
      if (!(abl & 4) && on(i)) kfrag(Ks, i, ko, kf[i % (LA + 1)]);
#pragma unroll
    for (int i = F0; i < NF; ++i) {
      if (i + LA < NF && !(abl & 4) && on(i + LA)) kfrag(Ks, i + LA, ko, kf[(i + LA) % (LA + 1)]);
      __builtin_amdgcn_sched_barrier(0);  // (left alone, hipcc sinks each read to its use: read, lgkmcnt(0), three MFMAs, ...)
      if (on(i)) {
        if (i % KG == 0) st[i / KG] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(abl & 2)) Op::mma(st[i / KG], kf[i % (LA + 1)], qf[i % KG]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // V^T fragment of feature tile nt: row e = 16 nt + lr of the block, chunk g (hi) / 4 + g (lo); again the lane term is hoisted
  static TAMF_DEV void vfrag(const char* vb, int nt, int vo0, int vo1, int4 (&vf)[2]) {
    vf[0] = *(const int4*)(vb + nt * (16 * C::VSTR) + vo0);
    if constexpr (TWO) vf[1] = *(const int4*)(vb + nt * (16 * C::VSTR) + vo1);
  }
  // O^T += V^T P^T over the key blocks [kb0, kb1) whose V^T blocks lie at Vs (block kb at Vs + (kb - kb0) * V_BYTES); before
  // that the blocks [dkb0, dkb1) are requested into `dma_dst` (never overlapping Vs)
  static TAMF_DEV void pv(const char* __restrict__ Vs, char* __restrict__ dma_dst, const char* vbase, int kb0, int kb1, int dkb0,
                          int dkb1, int Skp, int wave, int nw, int lane, const uint32_t (&ph)[NKB][4], const uint32_t (&pl)[NKB][4],
                          f32x4 (&o)[NT16], int abl, unsigned mask = ~0u) {
    const int lr = lane & 15, g = lane >> 4;
    const int vo0 = lr * C::VSTR + BLK::vswz(g, lr) * 16, vo1 = lr * C::VSTR + BLK::vswz(4 + g, lr) * 16;
    if (dkb1 > dkb0 && !(abl & 1)) issue_v(dma_dst, vbase, dkb0, dkb1, Skp, wave, nw, lane);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb >= kb0 && kb < kb1 && ((mask >> kb) & 1u)) {  // (wave-uniform)
        const char* vb = Vs + (kb - kb0) * C::V_BYTES;
        const int4 p_h = make_int4((int)ph[kb][0], (int)ph[kb][1], (int)ph[kb][2], (int)ph[kb][3]);
        const int4 p_l = make_int4((int)pl[kb][0], (int)pl[kb][1], (int)pl[kb][2], (int)pl[kb][3]);
        constexpr int LA = 2;
        int4 vf[LA + 1][2];
#pragma unroll
        for (int nt = 0; nt <= LA; ++nt) vf[nt][0] = vf[nt][1] = p_h;  // (defined contents for the no-read ablation)
#pragma unroll
        for (int nt = 0; nt < LA; ++nt)
          if (!(abl & 4)) vfrag(vb, nt, vo0, vo1, vf[nt]);
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) {
          if (nt + LA < NT16 && !(abl & 4)) vfrag(vb, nt + LA, vo0, vo1, vf[(nt + LA) % (LA + 1)]);
          __builtin_amdgcn_sched_barrier(0);
          const int4 vh = vf[nt % (LA + 1)][0];
          if (!(abl & 2)) {
            if constexpr (Op::PREC == 0) {
              // f32: k-slot of lane group g in MFMA (t, r) is key 16 t + 4 g + r; ph / pl carry the probabilities of t = 0 / 1 as floats
              // (the probabilities are taken from the scalar array elements: bit-casting the components of the int4 temporaries
              //  p_h / p_l read component 0 four times with hipcc 7.2 - cf. groups_reduce in tamf_device.h)
              const int4 v1 = vf[nt % (LA + 1)][1];
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vh.x), __builtin_bit_cast(float, ph[kb][0]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vh.y), __builtin_bit_cast(float, ph[kb][1]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vh.z), __builtin_bit_cast(float, ph[kb][2]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vh.w), __builtin_bit_cast(float, ph[kb][3]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(v1.x), __builtin_bit_cast(float, pl[kb][0]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(v1.y), __builtin_bit_cast(float, pl[kb][1]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(v1.z), __builtin_bit_cast(float, pl[kb][2]), o[nt], 0, 0, 0);
              o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(v1.w), __builtin_bit_cast(float, pl[kb][3]), o[nt], 0, 0, 0);
            } else {
              if constexpr (Op::SPLIT) {
                const int4 vl = vf[nt % (LA + 1)][1];
                o[nt] = Op::mfma1(vl, p_h, o[nt]);
                o[nt] = Op::mfma1(vh, p_l, o[nt]);
              }
              o[nt] = Op::mfma1(vh, p_h, o[nt]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
};

template <class Op, int HD, int NKB_>
__global__ __launch_bounds__(attn_res_maxw(NKB_) * 64) void attn_res_kernel(const AttnArgs<Op> aa) {
  typedef AttnCfg<Op, HD> C;
  typedef AttnRes<Op, HD, NKB_> R;
  constexpr int EB = Op::EB, KG = C::KG, NT16 = HD / 16, NKB = R::NKB, NKT = R::NKT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, nw = nthr >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  // (clip, head) pair of this workgroup: consecutive workgroups go to consecutive XCDs, so the pairs are dealt to the XCDs in
  // contiguous chunks (xcd_remap) - the clips whose Q | K / V^T rows the QKV GEMM's tiles on that XCD wrote last are then looked
  // up in the same L2 (aa.abl bit 5 in bench builds: plain order)
  int bh = blockIdx.y;
#ifndef TAMF_ATTN_PLAIN_ORDER  // (A/B builds)
  if (gridDim.x == 1 && !(TAMF_ABL(aa.abl) & 32)) bh = attn_xcd_remap(bh, gridDim.y);
#endif
  const int b = bh / aa.H, h = bh % aa.H;
  const int S = aa.S, Sp = aa.Sp, d = aa.d;
  // Key split of the clip's LAST query tile (aa.ksplit; launch_attn sets it when the clip has 4 n + 1 query tiles, e.g. 13 at T = 196).
  // The waves of a SIMD run their MFMA streams one after the other (per-wave phase stamps, f32: the score pass of waves 0 / 4 / 8 / 12
  // - SIMD 0 - ends after 9 900 / 17 300 / 24 800 / 32 100 ticks, the three waves of the other SIMDs after 24 600, and everybody waits at
  // the barrier for wave 12), so 13 tiles on 4 SIMDs cost 4 tile times, not 3.25.  The other tiles are dealt to the workgroups of
  // the pair as before; the last one belongs to four extra waves of the pair's LAST workgroup, placed (by launch_attn) on the SIMDs
  // that carry the fewest tiles there: wave hq takes the key blocks kb = hq mod 4 (scores, exact softmax over THOSE keys, P V) and
  // they merge through LDS: O = sum_i O_i 2^(m_i - m), l = sum_i l_i 2^(m_i - m), i = 0..3 in that order.  WHICH waves compute the
  // quarters depends on the batch (the query split); WHAT is computed depends on the clip's length only, so a clip's result stays
  // independent of the batch it is in.
  const int nwq = aa.ksplit ? aa.nwq : nw;   // waves that own a query tile
  const int nqt = (Sp + 15) / 16;
  const int nqt_own = aa.ksplit ? nqt - 1 : nqt;  // tiles dealt to tile-owning waves
  int tile = blockIdx.x * nwq + wave;
  int hq = -1;                               // >= 0: key-split wave hq of the last tile
  unsigned kmask = ~0u;                      // key blocks this wave multiplies
  bool idle = wave >= nwq || tile >= nqt_own;  // loads and barriers only
  if (aa.ksplit && blockIdx.x == gridDim.x - 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (wave == ((aa.hw >> (4 * j)) & 15)) hq = j;
    if (hq >= 0) { idle = false; tile = nqt - 1; kmask = 0x11111111u << hq; }
  }
  if (idle) { kmask = 0u; tile = nqt - 1; }
  const int q0 = tile * 16;
  const long row_base = (long)b * Sp;
  const int nkb = (S + 31) / 32;
  const int kbytes = Sp * C::KROWB;
  int n1 = (R::LDS_MAX - kbytes) / C::V_BYTES;
  n1 = n1 < nkb ? n1 : nkb;
  char* Ks = smem;
  char* Vs = smem + kbytes;

  const char* kbase = (const char*)aa.qk + (row_base * (2 * d) + d + h * HD) * EB;
  const char* vbase = (const char*)aa.vt + ((long)bh * HD) * aa.Skp * EB;

  // Q fragments first (oldest in the vmcnt order), then the K burst
  int4 qf[KG][2];
  {
    int q = q0 + lr;
    q = q < Sp ? q : Sp - 1;  // (a split of the queries over workgroups may leave the last wave without a tile: it recomputes the last row)
    const char* qb = (const char*)aa.qk + ((row_base + q) * (2 * d) + h * HD) * EB + g * 16;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      qf[kg][0] = *(const int4*)(qb + kg * 128);
      qf[kg][1] = *(const int4*)(qb + kg * 128 + 64);
    }
  }
  const int abl = TAMF_ABL(aa.abl);
  // K arrives in two halves: the key tiles [0, NKT / 2) first - every CU of the chip asks for its K at the same moment, and nothing
  // can be multiplied before the first bytes are there - the rest (and V^T) under the score products of the first half
  constexpr int TH = NKT / 2;
  const int kh = TH * 16 < Sp ? TH * 16 : Sp;  // rows of the first half
  if (!(abl & 1)) R::issue_k(Ks, kbase, 0, kh, d, wave, nw, lane);
  __syncthreads();  // the first half of K has landed (vmcnt(0) of every wave + barrier)

  // ---- pass 1: scores of this wave's 16 queries against all keys; V^T blocks [0, n1) fly underneath
  f32x4 st[NKT];
  if (kmask != ~0u) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) st[kt] = f32x4{-1e30f, -1e30f, -1e30f, -1e30f};
  }
  R::template scores<0, TH>(Ks, Ks + (long)kh * C::KROWB, Vs, kbase, vbase, kh, Sp, d, n1, aa.Skp, wave, nw, lane, qf, st, abl, kmask);
  __syncthreads();  // the second half of K (and the V^T blocks) have landed
  R::template scores<TH, NKT>(Ks, Vs, Vs, kbase, vbase, 0, 0, d, 0, aa.Skp, wave, nw, lane, qf, st, abl, kmask);

  // ---- exact softmax over the keys (per query = per lane column; the 4 lane groups hold disjoint keys)
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
    if (kt * 16 + 15 >= R::S_MIN) {  // (compile time: only these tiles can hold keys >= S for the clip lengths this instantiation serves)
      // branch-free: a (wave-uniform) branch around the rewrite made every st[] a phi of two versions, and the short-clip form - where
      // all eight tiles can be masked - ended up with 1 000 register moves and 352 - 1 048 bytes of spills per lane (round 4 verdict #8)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[kt][r] = (kt * 16 + 4 * g + r >= S) ? -1e30f : st[kt][r];
    }
  float m = -1e30f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) m = fmaxf(m, fmaxf(fmaxf(st[kt][0], st[kt][1]), fmaxf(st[kt][2], st[kt][3])));
  m = groups_reduce<RedMax>(m);
  float l = 0.f;
  uint32_t ph[NKB][4], pl[NKB][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (abl & 8) ? st[2 * kb + t][r] : __builtin_amdgcn_exp2f(st[2 * kb + t][r] - m);
        st[2 * kb + t][r] = p;
        l += p;
      }
    // B fragment element j of lane group g is P[key 4g + j] (j < 4) / P[key 16 + 4g + j - 4] (j >= 4) of the block
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (Op::PREC == 0) {  // f32: the probabilities themselves, key tile 2 kb in ph, 2 kb + 1 in pl
        // (copied to scalars first: __builtin_bit_cast straight from a vector element reads element 0 with hipcc 7.2, tamf_device.h)
        const float p0 = st[2 * kb][j], p1 = st[2 * kb + 1][j];
        ph[kb][j] = __builtin_bit_cast(uint32_t, p0);
        pl[kb][j] = __builtin_bit_cast(uint32_t, p1);
      } else {
        if (abl & 8) { const float pa = st[2 * kb + (j >> 1)][(2 * j) & 3]; ph[kb][j] = pl[kb][j] = __builtin_bit_cast(uint32_t, pa); }
        else Op::split2(st[2 * kb + (j >> 1)][(2 * j) & 3], st[2 * kb + (j >> 1)][((2 * j) & 3) + 1], ph[kb][j], pl[kb][j]);
      }
    }
  }
  l = groups_reduce<RedSum>(l);

  f32x4 o[NT16];
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();  // V^T blocks [0, n1) have landed; nobody reads K any more
  // ---- pass 2: O^T = V^T P^T; the blocks that did not fit beside K are fetched into K's space meanwhile
  R::pv(Vs, Ks, vbase, 0, n1, n1, nkb, aa.Skp, wave, nw, lane, ph, pl, o, abl, kmask);
  if (n1 < nkb) {
    __syncthreads();
    R::pv(Ks, Vs, vbase, n1, nkb, 0, 0, aa.Skp, wave, nw, lane, ph, pl, o, abl, kmask);
  }
  if (idle) {  // (stores nothing; the barrier is the one the key-split waves wait at before their merge)
    if (aa.ksplit) __syncthreads();
    return;
  }

  // ---- output: the wave's 16 x HD tile goes through a lane-private 2-KiB LDS slot per 128-byte row group, so that a store
  // instruction writes whole 128-byte lines (8 rows x 128 B per dwordx4 wave-instruction) instead of 8-byte pieces of 16 rows:
  // the row-per-lane form (16 global_store_dwordx2 per lane) cost 7.7 of the kernel's 32 us (tools/attn_bench.py, ablation 16).
  // The slot lies in the part of LDS nobody reads in the last phase: the first V^T area when the blocks were split, K otherwise.
  if ((abl & 16) && l != 12345.0f) return;
  constexpr int TPC = Op::EB == 4 ? 2 : 4;     // feature tiles per 128-byte row group (32 f32 or split elements / 64 bf16)
  constexpr int NCH = (NT16 + TPC - 1) / TPC, TPCE = NT16 < TPC ? NT16 : TPC;
  if (hq >= 0) {
    // partials of the key-split waves: [row group ch][foreign helper 0..2][tile of the group] x 1 KiB (lane-major f32x4), then m / l of
    // the four.  The area: behind the output slots of the tile-owning waves inside the first V^T area when V^T was fetched in two parts
    // (nobody reads that area in the last phase; the helpers store directly and need no slot), else behind everything; launch_attn
    // sizes the allocation (AttnRes::ksplit_extra)
    const int base_bytes = kbytes + n1 * C::V_BYTES > NKT * 16 * C::KROWB ? kbytes + n1 * C::V_BYTES : NKT * 16 * C::KROWB;  // (= AttnRes::smem)
    char* part = n1 < nkb ? Vs + nwq * 2048 : smem + base_bytes;
    float* ml = (float*)(part + NCH * 3 * TPCE * 1024);
    if (g == 0) { ml[hq * 32 + lr] = m; ml[hq * 32 + 16 + lr] = l; }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      if (ch != hq) {
        const int src = hq < ch ? hq : hq - 1;
#pragma unroll
        for (int tl = 0; tl < TPCE; ++tl) *(f32x4*)(part + ((ch * 3 + src) * TPCE + tl) * 1024 + lane * 16) = o[ch * TPC + tl];
      }
    __syncthreads();  // the partials are in LDS (every wave of the workgroup comes here once: the others at the end of their stores)
    if (hq >= NCH || q0 + lr >= Sp) return;
    float mi[4], li[4], mg = -1e30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { mi[i] = ml[i * 32 + lr]; li[i] = ml[i * 32 + 16 + lr]; mg = fmaxf(mg, mi[i]); }
    float w[4], lg = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = __builtin_amdgcn_exp2f(mi[i] - mg); lg += li[i] * w[i]; }
    const float invg = 1.0f / lg;
    char* grow = (char*)aa.out + ((row_base + q0 + lr) * d + h * HD) * EB + hq * 128;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {  // (static indices into o[]: a run-time index would put the accumulators into scratch)
      if (ch != hq) continue;
#pragma unroll
      for (int tl = 0; tl < TPCE; ++tl) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (fixed order: helper 0, 1, 2, 3)
          const f32x4 mine = o[ch * TPC + tl];
          const f32x4 theirs = *(const f32x4*)(part + ((ch * 3 + (i < ch ? i : (i > 0 ? i - 1 : 0))) * TPCE + tl) * 1024 + lane * 16);
          const f32x4 oi = i == ch ? mine : theirs;
          acc += oi * w[i];
        }
        const float v0 = acc[0] * invg, v1 = acc[1] * invg, v2 = acc[2] * invg, v3 = acc[3] * invg;
        if constexpr (Op::PREC == 0) {
          *(int4*)(grow + (4 * tl + g) * 16) = make_int4(as_i(v0), as_i(v1), as_i(v2), as_i(v3));
        } else {
          uint32_t h0, l0, h1, l1;
          Op::split2(v0, v1, h0, l0);
          Op::split2(v2, v3, h1, l1);
          char* gp = grow + (2 * tl + (g >> 1)) * 16 + 8 * (g & 1);  // (the layout of the slot rows below: hi plane, lo plane 64 bytes further)
          *(int2*)gp = make_int2((int)h0, (int)h1);
          if constexpr (Op::SPLIT) *(int2*)(gp + 64) = make_int2((int)l0, (int)l1);
        }
      }
    }
    return;
  }
  const float inv = 1.0f / l;
  // slot rows are 128 bytes (a wave's slot = 2 KiB <= the 16 K rows of its own queries, so the slots always fit the area they borrow),
  // their 16-byte chunks XOR-swizzled by the row: the 8-byte column writes of the 16 lanes of a group are two-way conflicts at worst
  constexpr int RS = 128;
  char* slot = (n1 < nkb ? Vs : Ks) + wave * (16 * RS);
  char* gout = (char*)aa.out + ((row_base + q0) * d + h * HD) * EB;
  const int srow = lane >> 3, spiece = lane & 7;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int tl = 0; tl < TPCE; ++tl) {
      const int nt = ch * TPC + tl;
      const float v0 = o[nt][0] * inv, v1 = o[nt][1] * inv, v2 = o[nt][2] * inv, v3 = o[nt][3] * inv;
      if (q0 + lr >= Sp) continue;  // (rows past the clip are not stored - and with 128-byte K rows their slot rows would lie outside K)
      if constexpr (Op::PREC == 0) {
        // (written with the integer vector type the read-back below uses: a float4 store and an int4 load of the same bytes do not
        //  alias under the type-based rules)
        const int c = 4 * tl + g;  // 16-byte chunk of elements 16 tl + 4 g ..
        *(int4*)(slot + lr * RS + ((c ^ (lr & 7)) << 4)) = make_int4(as_i(v0), as_i(v1), as_i(v2), as_i(v3));
      } else {
        uint32_t h0, l0, h1, l1;
        Op::split2(v0, v1, h0, l0);
        Op::split2(v2, v3, h1, l1);
        // element e = 16 tl + 4 g of the group: hi at byte 2 e = chunk 2 tl + g / 2, half g & 1; lo 4 chunks (64 bytes) further
        const int c = 2 * tl + (g >> 1);
        char* sp = slot + lr * RS + 8 * (g & 1);
        *(int2*)(sp + ((c ^ (lr & 7)) << 4)) = make_int2((int)h0, (int)h1);
        if constexpr (Op::SPLIT) *(int2*)(sp + (((c + 4) ^ (lr & 7)) << 4)) = make_int2((int)l0, (int)l1);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 8 + srow;
      const int4 w = *(const int4*)(slot + row * RS + ((spiece ^ (row & 7)) << 4));
      if (q0 + row < Sp && (NT16 >= TPC || spiece * 16 < NT16 * 16 * Op::EB))
        *(int4*)(gout + (long)row * d * EB + ch * 128 + spiece * 16) = w;
    }
  }
  if (aa.ksplit) __syncthreads();  // (the barrier the key-split waves wait at before their merge)
}
