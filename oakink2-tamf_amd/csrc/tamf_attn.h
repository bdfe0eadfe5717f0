// Multi-head self-attention over one clip's S = T + prefix tokens (full, unmasked softmax(QK^T)V per
// (clip, head); interaction_segment_mdm.py:63-70,171 -> nn.MultiheadAttention).
//
// Workgroup = NW waves = NW tiles of 16 queries of one (clip, head); keys/values are streamed through LDS in
// blocks of 32 keys with an online softmax, so LDS and registers are bounded for every arithmetic mode.
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
  // LDS stride of a V^T row: bf16 and bf16x3 are read with 8-byte accesses and padded (conflict-free per
  // tools/lds_bank_sim.py); f32 is read with ds_read_b128 and XOR-swizzled like a GEMM tile row
  static constexpr int VSTR = (Op::PREC == 1) ? 80 : (Op::PREC == 2 ? 144 : 128);
  static constexpr int K_BYTES = 32 * KROWB;
  static constexpr int V_BYTES = HD * VSTR;
  static constexpr int SMEM = K_BYTES + V_BYTES;
  static_assert(KROWB % 128 == 0, "head slice must be whole 128-byte groups");
};

template <class Op, int HD>
__global__ __launch_bounds__(512) void attn_kernel(const AttnArgs<Op> aa) {
  typedef AttnCfg<Op, HD> C;
  constexpr int EB = Op::EB, KG = C::KG;
  constexpr int NT16 = HD / 16;  // output tiles along e
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + C::K_BYTES;

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
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
  constexpr int KCH = C::KROWB / 16;  // chunks per K row
  constexpr int VCH = C::VROWB / 16;  // chunks per V^T row block
  // K-row swizzle: 128-byte rows use the GEMM tile swizzle, longer rows XOR the low 4 chunk bits with the row
  auto kswz = [](int ch, int row) -> int {
    if constexpr (C::KROWB >= 256) return (ch & ~15) | ((ch ^ row) & 15);
    else return ch ^ ((row >> 1) & 7);
  };

  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();  // everyone is done reading the previous block
    // ---- stage K block: 32 keys x KROWB bytes (rows past the clip are clamped; they are masked below)
    for (int q = tid; q < 32 * KCH; q += nthr) {
      const int r = q / KCH, ch = q % KCH;
      int key = kb * 32 + r;
      key = key < Sp ? key : Sp - 1;
      const int4 v = *(const int4*)((const char*)aa.qk + ((row_base + key) * (2 * d) + d + h * HD) * EB + ch * 16);
      *(int4*)(Ks + r * C::KROWB + kswz(ch, r) * 16) = v;
    }
    // ---- stage V^T block: HD rows x 32 keys
    for (int q = tid; q < HD * VCH; q += nthr) {
      const int e = q / VCH, ch = q % VCH;
      const int4 v = *(const int4*)((const char*)aa.vt + (((long)bh * HD + e) * aa.Skp + kb * 32) * EB + ch * 16);
      int sch = ch;
      if constexpr (Op::PREC == 0) sch = ch ^ ((e >> 1) & 7);
      *(int4*)(Vs + e * C::VSTR + sch * 16) = v;
    }
    __syncthreads();
    if (!active) continue;

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
    float bm = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb * 32 + t * 16 + 4 * g + r;
        if (key >= S) st[t][r] = -1e30f;
        bm = fmaxf(bm, st[t][r]);
      }
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float m_new = fmaxf(m_run, bm);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[t][r] = __builtin_amdgcn_exp2f(st[t][r] - m_new);
        ps += st[t][r];
      }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) o[nt] *= alpha;

    // ---- O^T += V^T . P^T
    if constexpr (Op::PREC == 0) {
      // f32: k-slot of lane group g in MFMA (t, r) is key 16t + 4g + r
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const int e = nt * 16 + lr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ch = (t * 4 + g) ^ ((e >> 1) & 7);
          const int4 vf = *(const int4*)(Vs + e * C::VSTR + ch * 16);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.x), st[t][0], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.y), st[t][1], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.z), st[t][2], o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(vf.w), st[t][3], o[nt], 0, 0, 0);
        }
      }
    } else {
      // bf16: B fragment element j of lane group g is P[key 4g + j] (j < 4) / P[key 16 + 4g + j - 4] (j >= 4);
      // the V^T A fragment is read with the same key permutation (two 8-byte reads per plane).
      uint32_t hi[8];
      float lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = st[j >> 2][j & 3];
        hi[j] = f2bf(pv);
        lo[j] = pv - bf2f(hi[j]);
      }
      const bf16x8 ph = __builtin_bit_cast(bf16x8, make_int4((int)(hi[0] | (hi[1] << 16)), (int)(hi[2] | (hi[3] << 16)),
                                                              (int)(hi[4] | (hi[5] << 16)), (int)(hi[6] | (hi[7] << 16))));
      bf16x8 pl = ph;
      if constexpr (Op::PREC == 2) {
        uint32_t l2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) l2[j] = f2bf(lo[j]);
        pl = __builtin_bit_cast(bf16x8, make_int4((int)(l2[0] | (l2[1] << 16)), (int)(l2[2] | (l2[3] << 16)),
                                                   (int)(l2[4] | (l2[5] << 16)), (int)(l2[6] | (l2[7] << 16))));
      }
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const char* vp = Vs + (nt * 16 + lr) * C::VSTR + 8 * g;
        const int2 a0 = *(const int2*)vp;
        const int2 a1 = *(const int2*)(vp + 32);
        const bf16x8 vh = __builtin_bit_cast(bf16x8, make_int4(a0.x, a0.y, a1.x, a1.y));
        if constexpr (Op::PREC == 2) {
          const int2 b0 = *(const int2*)(vp + 64);
          const int2 b1 = *(const int2*)(vp + 96);
          const bf16x8 vl = __builtin_bit_cast(bf16x8, make_int4(b0.x, b0.y, b1.x, b1.y));
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[nt], 0, 0, 0);
        }
        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[nt], 0, 0, 0);
      }
    }
  }

  if (!active) return;
  const int q = q0 + lr;
  if (q >= Sp) return;
  const float inv = 1.0f / l_run;
  // lane (g, lr): query q, e = 16 nt + 4g + reg -> 4 consecutive elements per tile
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) {
    float v[4] = {o[nt][0] * inv, o[nt][1] * inv, o[nt][2] * inv, o[nt][3] * inv};
    Op::template store<4>(aa.out, (row_base + q) * d + h * HD + nt * 16 + 4 * g, v);
  }
}
