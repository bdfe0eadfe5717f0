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
