// Device-side building blocks shared by the gfx950 kernels: MFMA operand traits for the three arithmetic
// modes, LDS swizzles, bf16 split helpers, wave reductions, Philox noise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define TAMF_DEV __device__ __forceinline__

// Kernel-benchmark ablations (no loads / no MFMAs / no epilogue / ..., tools/kbench.py) deliberately produce wrong results: they
// exist only in -DTAMF_BENCH builds (tools/ab_build.sh with TAMF_HIPCC_FLAGS=-DTAMF_BENCH); the product library compiles them
// out, so no tuning word can switch arithmetic off.
#ifdef TAMF_BENCH
#define TAMF_ABL(x) (x)
#else
#define TAMF_ABL(x) 0
#endif

// -DTAMF_TIMELINE: debug build whose GEMM and attention kernels stamp per-workgroup phase times (100 MHz wall clock)
// into device buffers read back by tamf_debug_timeline (tools/gemm_timeline.py, tools/attn_timeline.py)
#ifdef TAMF_TIMELINE
#define TAMF_TS(var) const unsigned long long var = wall_clock64()
__device__ __forceinline__ unsigned long long tamf_hw_cu_id() {  // (XCC id << 32) | HW_ID (cu, sh, se in bits 8..15)
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  return ((unsigned long long)(xcc & 15) << 32) | hw;
}
#else
#define TAMF_TS(var)
#endif

// Wide global stores of the streaming outputs (activation operands, the residual stream): plain stores.  Write-through
// (sc1) and nt stores were measured on the whole loop and lose 10 % (2.63 -> 2.89 / 2.86 ms per step, f16x3): the consumers
// of X_op / QK_op / A_op find a good part of them in the writer's L2, which a write-through store gives up.
typedef uint32_t tamf_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t tamf_u32x2 __attribute__((ext_vector_type(2)));
TAMF_DEV void gst16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const tamf_u32x4 v = {a, b, c, d};
  *(tamf_u32x4*)p = v;
}
TAMF_DEV void gst8(void* p, uint32_t a, uint32_t b) {
  const tamf_u32x2 v = {a, b};
  *(tamf_u32x2*)p = v;
}
#ifdef TAMF_H_NT  // (A/B build, round 6: the FFN hidden activations - 109 MB per launch in the split modes, read once by FFN2 - stored
// non-temporally, so that they do not push the weight panel and the clips' A panels out of the XCD's L2: DESIGN.md section 6, "re-fetch")
TAMF_DEV void gst16_nt(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const tamf_u32x4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, (tamf_u32x4*)p);
}
TAMF_DEV void gst8_nt(void* p, uint32_t a, uint32_t b) {
  const tamf_u32x2 v = {a, b};
  __builtin_nontemporal_store(v, (tamf_u32x2*)p);
}
#endif
TAMF_DEV void gst16f(float* p, float a, float b, float c, float d) {
  gst16(p, __builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d));
}

TAMF_DEV uint32_t f2bf(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)x); }
TAMF_DEV float bf2f(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
// two floats -> packed bf16 pair (round to nearest even; one v_cvt_pk_bf16_f32), element 0 in the low half
typedef float tamf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tamf_bf16x2 __attribute__((ext_vector_type(2)));
TAMF_DEV uint32_t pack_bf16(float a, float b) {
  const tamf_f32x2 f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, tamf_bf16x2));
}
// bf16x3 split of a pair: hi = bf16(v), lo = bf16(v - hi); both packed, element 0 in the low half (6 VALU per pair)
// (fp contract off, here and in the other splits: the subtraction must not fuse with a multiply that produced `a` in the
// caller - whether it does depends on the code around the call, and two kernel variants would then store different lo
// words for the same value: the batch-invariance tests compare them bit for bit)
TAMF_DEV void split_bf16x3(float a, float b, uint32_t& hi, uint32_t& lo) {
#pragma clang fp contract(off)
  hi = pack_bf16(a, b);
  lo = pack_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u));
}
// the same for IEEE half: hi = f16(v), lo = f16(v - hi) (v - hi is exact in fp32); |v| must stay below 65504
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tamf_f16x2 __attribute__((ext_vector_type(2)));
TAMF_DEV uint32_t pack_f16(float a, float b) {
  const tamf_f32x2 f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, tamf_f16x2));
}
TAMF_DEV void split_f16x3(float a, float b, uint32_t& hi, uint32_t& lo) {
#pragma clang fp contract(off)
  hi = pack_f16(a, b);
  const tamf_f16x2 h = __builtin_bit_cast(tamf_f16x2, hi);
  lo = pack_f16(a - (float)h[0], b - (float)h[1]);
}
TAMF_DEV float as_f(int v) { return __builtin_bit_cast(float, v); }
TAMF_DEV int as_i(float v) { return __builtin_bit_cast(int, v); }

// torch.nan_to_num defaults: NaN -> 0, +-inf -> +-FLT_MAX  (interaction_segment_mdm.py:158,166,173)
TAMF_DEV float nan_to_num(float v) {
  if (v != v) return 0.0f;
  if (v == __builtin_inff()) return 3.4028234663852886e38f;
  if (v == -__builtin_inff()) return -3.4028234663852886e38f;
  return v;
}
TAMF_DEV float silu_exact(float x) { return x / (1.0f + expf(-x)); }
// v_exp_f32 / v_rcp_f32 form (each ~1 ulp) for the bf16 modes
TAMF_DEV float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
TAMF_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7 + exp/rcp rounding): ~3x fewer VALU instructions than the
// libm erff; used by the bf16 / bf16x3 epilogues (the f32 parity mode keeps erff)
TAMF_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float y = fmaf(-p * t, e, 1.0f);
  return copysignf(y, x);
}
// (measured and dropped, round 3: GELU(x) = max(x, 0) - |x| q(|x|) with the 7.1.26 polynomial's coefficients halved - three VALU
//  instructions fewer per element and 3.3e-7 instead of 4.7e-7 max abs error on paper, but the FFN1 launch got SLOWER, 81.2 -> 84.9 us
//  (f16x3, tools/kbench.py, two builds alternating on one box): the epilogue's time is not its instruction count)
TAMF_DEV float gelu_erf_fast(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// (measured and dropped, round 5: the f32 mode's GELU through erfc(|z|) = t exp(-z^2 + P(t)) (Numerical Recipes' erfcc fit; ~20 VALU
//  instructions per element against ~75 of libm's erff, the same 4e-7 max abs error in fp32) - the f32 FFN1 launch did not move,
//  221.3 -> 222.4 us (profiles/r05/ab_f32_deferred_c16.txt): again, the epilogue's time is not its instruction count.  And its
//  `2 - t * e` contracted to an fma in one tile form and not in the other: the batch-invariance test caught it - arithmetic shared
//  by two kernel forms must leave the compiler no choice, hence the explicit fmaf / __fmul_rn / contract(off) everywhere.)

// Per-loop inputs of the fused DDPM update.  They live in device memory and are read by the kernel, so the captured graph
// does not depend on them: a new seed / clip range / noise tensor replays the same executable graph.
struct LoopParams {
  const float* noise;   // (n_steps+1, B, F, 1, T) draws in reference call order, or null -> Philox
  float* dump;          // (n_steps, B, F, 1, T) or null
  long noise_draw_stride;
  unsigned long long seed;
  long long clip_base;
};

// ---------------------------------------------------------------------------------------------
// LDS swizzles: a tile row of ROWB bytes is a sequence of 16-byte chunks; an MFMA fragment read takes, for
// lane (r = lane & 15, g = lane >> 4), chunk 4*kc + g of row r.  XOR-ing the chunk index with a function of
// the row makes the ds_read_b128 lane groups conflict-free (tools/lds_bank_sim.py).
// ---------------------------------------------------------------------------------------------
template <int ROWB>
TAMF_DEV int swz_chunk(int row) {
  if constexpr (ROWB == 64) {
    return (4 - ((row >> 2) & 3)) & 3;
  } else if constexpr (ROWB == 128) {
    return (row >> 1) & 7;
  } else {
    return row & 15;
  }
}

// ---------------------------------------------------------------------------------------------
// Sticky status word.  Every CONTEXT owns one (a 4-byte device allocation whose pointer travels in the epilogue structs and the
// small kernels' arguments; read and cleared by tamf_get_status_flags of that context only), so two live contexts on one device
// never see or clear each other's bits.  The process-wide word below serves only the context-less kernel test hooks.
//   bit 0 (TAMF_STATUS_F16_RANGE): a value beyond the fp16 range (|v| > 65504, +-inf included; NaN is not counted) was
//   stored as a split-fp16 operand - its hi part is then inf, its lo part NaN, and the products it enters differ from the
//   reference's fp32 ones.  Raised by the kernels that split activations (Op::store_rc + Op::range_flag); weights are checked on the host
//   (upload_operand).  Attention probabilities (<= 2^8 with the deferred rescale) are split without the check.
// ---------------------------------------------------------------------------------------------
__device__ unsigned g_tamf_status;
TAMF_DEV void f16_range_flag(float absmax, unsigned* status) {
  if (absmax > 65504.0f) atomicOr(status ? status : &g_tamf_status, 1u);
}
// What every epilogue carries besides its own fields (always its LAST member, so the positional initialisers stay short):
//   wscale  2^-k of the weight tensor of this GEMM: f16x3 weights are stored pre-scaled by 2^k (upload_operand) so that their lo
//           planes leave the fp16 subnormal range; the accumulator is multiplied back in the bias add, fma(acc, 2^-k, bias) - exact,
//           a power of two - and 1.0 in every other mode (fma(acc, 1, b) = acc + b bit for bit)
//   status  the context's status word (null: the process word, kernel test hooks)
struct EpiCtl {
  float wscale = 1.0f;
  unsigned* status = nullptr;
};

// ---------------------------------------------------------------------------------------------
// Operand traits.  Every operand matrix is row-major with K contiguous and is consumed in 128-byte row groups:
//     f32     32 floats
//     bf16    64 bf16
//     bf16x3  32 elements as [hi: 32 bf16 | lo: 32 bf16]   (hi = bf16(x), lo = bf16(x - hi), interleaved per 64 bytes)
// A lane (r = lane & 15, g = lane >> 4) holds two 16-byte fragments of row r of a group: f[0] = bytes [16g, 16g+16),
// f[1] = bytes [64+16g, 64+16g+16).  mma(acc, a, b) accumulates the 16 x 16 product of a 16-row A group and a 16-row
// B group over the group's K range: acc layout row = 4*(lane>>4)+reg (A row), col = lane & 15 (B row).
// ---------------------------------------------------------------------------------------------
struct OpF32 {
  typedef float elem_t;
  static constexpr int EB = 4;     // bytes per logical element inside a row
  static constexpr int PREC = 0;
  static constexpr bool SPLIT = false;
  static TAMF_DEV void mma(f32x4& acc, const int4 (&a)[2], const int4 (&b)[2]) {
    // the 4 floats of a fragment are 4 K-steps of v_mfma_f32_16x16x4_f32 (lane group g supplies k = g); A and B use
    // the same permuted K order, so the products pair up exactly (bitwise a k-ordered fp32 fma chain)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(a[f].x), as_f(b[f].x), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(a[f].y), as_f(b[f].y), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(a[f].z), as_f(b[f].z), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(a[f].w), as_f(b[f].w), acc, 0, 0, 0);
    }
  }
  // mma_t(acc, x, w): the TRANSPOSE of mma(acc, w, x) - x is the MFMA's row operand - with the same products in the same
  // order, i.e. the same bits per output element (the V projection stored transposed, tamf_gemm_clip.h)
  static TAMF_DEV void mma_t(f32x4& acc, const int4 (&x)[2], const int4 (&w)[2]) { mma(acc, x, w); }
  // byte offset of logical element idx (row * ld + col; ld % 32 == 0) from the matrix base
  static TAMF_DEV long byte_off(long idx) { return idx * 4; }
  template <int N, bool NT = false>
  static TAMF_DEV void store(elem_t* base, long idx, const float* v) {
    float* p = base + idx;
    if constexpr (N == 8) {
      gst16f(p, v[0], v[1], v[2], v[3]);
      gst16f(p + 4, v[4], v[5], v[6], v[7]);
    } else if constexpr (N == 4) {
      gst16f(p, v[0], v[1], v[2], v[3]);
    } else {
      *(float2*)p = make_float2(v[0], v[1]);
    }
  }
  template <int N, bool NT = false>
  static TAMF_DEV void store_rc(elem_t* base, long idx, const float* v, float&) { store<N, NT>(base, idx, v); }  // (fp32 exponent range: nothing to check)
  static TAMF_DEV void range_flag(float, unsigned*) {}
  static TAMF_DEV void store1(elem_t* base, long idx, float v) { base[idx] = v; }
  static TAMF_DEV float load1(const elem_t* base, long idx) { return base[idx]; }
};

template <int N, bool NT = false>
TAMF_DEV void store_bf16_vec(char* p, const uint32_t* w) {
#ifdef TAMF_H_NT
  if constexpr (NT && N == 8) { gst16_nt(p, w[0], w[1], w[2], w[3]); return; }
  if constexpr (NT && N == 4) { gst8_nt(p, w[0], w[1]); return; }
#endif
  if constexpr (N == 8) {
    gst16(p, w[0], w[1], w[2], w[3]);
  } else if constexpr (N == 4) {
    gst8(p, w[0], w[1]);
  } else {
    *(uint32_t*)p = w[0];
  }
}

struct OpBF16 {
  typedef uint16_t elem_t;
  static constexpr int EB = 2;
  static constexpr int PREC = 1;
  static constexpr bool SPLIT = false;  // one 16-bit plane
  static TAMF_DEV f32x4 mfma1(const int4& a, const int4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static TAMF_DEV void split2(float a, float b, uint32_t& hi, uint32_t& lo) { hi = pack_bf16(a, b); lo = hi; }
  static TAMF_DEV void mma(f32x4& acc, const int4 (&a)[2], const int4 (&b)[2]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
  }
  static TAMF_DEV void mma_t(f32x4& acc, const int4 (&x)[2], const int4 (&w)[2]) { mma(acc, x, w); }
  static TAMF_DEV long byte_off(long idx) { return idx * 2; }
  template <int N, bool NT = false>
  static TAMF_DEV void store(elem_t* base, long idx, const float* v) {
    uint32_t w[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) w[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
    store_bf16_vec<N, NT>((char*)base + idx * 2, w);
  }
  template <int N, bool NT = false>
  static TAMF_DEV void store_rc(elem_t* base, long idx, const float* v, float&) { store<N, NT>(base, idx, v); }  // (fp32 exponent range: nothing to check)
  static TAMF_DEV void range_flag(float, unsigned*) {}
  static TAMF_DEV void store1(elem_t* base, long idx, float v) { base[idx] = (uint16_t)f2bf(v); }
  static TAMF_DEV float load1(const elem_t* base, long idx) { return bf2f(base[idx]); }
};

struct OpBF16X3 {
  typedef uint16_t elem_t;
  static constexpr int EB = 4;  // hi + lo
  static constexpr int PREC = 2;
  static constexpr bool SPLIT = true;  // hi and lo planes, 3 MFMAs per product
  static TAMF_DEV f32x4 mfma1(const int4& a, const int4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static TAMF_DEV void split2(float a, float b, uint32_t& hi, uint32_t& lo) { split_bf16x3(a, b, hi, lo); }
  static TAMF_DEV void mma(f32x4& acc, const int4 (&a)[2], const int4 (&b)[2]) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), al = __builtin_bit_cast(bf16x8, a[1]);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bl = __builtin_bit_cast(bf16x8, b[1]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
  }
  static TAMF_DEV void mma_t(f32x4& acc, const int4 (&x)[2], const int4 (&w)[2]) {  // w_lo.x_hi, w_hi.x_lo, w_hi.x_hi as in mma(acc, w, x)
    acc = mfma1(x[0], w[1], acc);
    acc = mfma1(x[1], w[0], acc);
    acc = mfma1(x[0], w[0], acc);
  }
  // element idx lives in 128-byte group idx / 32: hi at 2 * (idx % 32), lo 64 bytes further
  static TAMF_DEV long byte_off(long idx) { return ((idx >> 5) << 7) + ((idx & 31) << 1); }
  template <int N, bool NT = false>
  static TAMF_DEV void store(elem_t* base, long idx, const float* v) {
    uint32_t wh[N / 2], wl[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) split_bf16x3(v[2 * i], v[2 * i + 1], wh[i], wl[i]);
    char* p = (char*)base + byte_off(idx);
    store_bf16_vec<N, NT>(p, wh);
    store_bf16_vec<N, NT>(p + 64, wl);
  }
  template <int N, bool NT = false>
  static TAMF_DEV void store_rc(elem_t* base, long idx, const float* v, float&) { store<N, NT>(base, idx, v); }  // (fp32 exponent range: nothing to check)
  static TAMF_DEV void range_flag(float, unsigned*) {}
  static TAMF_DEV void store1(elem_t* base, long idx, float v) {
#pragma clang fp contract(off)
    char* p = (char*)base + byte_off(idx);
    const uint32_t hi = f2bf(v);
    *(uint16_t*)p = (uint16_t)hi;
    *(uint16_t*)(p + 64) = (uint16_t)f2bf(v - bf2f(hi));
  }
  static TAMF_DEV float load1(const elem_t* base, long idx) {
    const char* p = (const char*)base + byte_off(idx);
    return bf2f(*(const uint16_t*)p) + bf2f(*(const uint16_t*)(p + 64));
  }
};

// split-fp16 operands: x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 significand bits, i.e. ~2^-22 relative operand
// error (fp32 keeps 2^-24; bf16x3 2^-17), three f16 MFMAs per product (lo.hi + hi.lo + hi.hi, the lo.lo term is below
// 2^-22).  Same 128-byte row groups as bf16x3: [hi: 32 f16 | lo: 32 f16].  Range: |x| <= 65504 (an element beyond it turns
// into inf / NaN exactly like a non-finite activation would); f16 subnormal lo parts (|x| < ~0.12) keep an absolute
// error of 3e-8.
struct OpF16X3 {
  typedef uint16_t elem_t;
  static constexpr int EB = 4;
  static constexpr int PREC = 3;
  static constexpr bool SPLIT = true;
  static TAMF_DEV f32x4 mfma1(const int4& a, const int4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static TAMF_DEV void split2(float a, float b, uint32_t& hi, uint32_t& lo) { split_f16x3(a, b, hi, lo); }
  static TAMF_DEV void mma(f32x4& acc, const int4 (&a)[2], const int4 (&b)[2]) {
    acc = mfma1(a[1], b[0], acc);
    acc = mfma1(a[0], b[1], acc);
    acc = mfma1(a[0], b[0], acc);
  }
  static TAMF_DEV void mma_t(f32x4& acc, const int4 (&x)[2], const int4 (&w)[2]) {  // w_lo.x_hi, w_hi.x_lo, w_hi.x_hi as in mma(acc, w, x)
    acc = mfma1(x[0], w[1], acc);
    acc = mfma1(x[1], w[0], acc);
    acc = mfma1(x[0], w[0], acc);
  }
  static TAMF_DEV long byte_off(long idx) { return ((idx >> 5) << 7) + ((idx & 31) << 1); }
  template <int N, bool NT = false>
  static TAMF_DEV void store(elem_t* base, long idx, const float* v) {
    uint32_t wh[N / 2], wl[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) split_f16x3(v[2 * i], v[2 * i + 1], wh[i], wl[i]);
    char* p = (char*)base + byte_off(idx);
    store_bf16_vec<N, NT>(p, wh);
    store_bf16_vec<N, NT>(p + 64, wl);
  }
  // range-checked store: folds max |v| of the run into the caller's accumulator `am` (v_max3_f32 with |.| modifiers: one
  // instruction per pair, NaN operands drop out); the caller raises the status bit ONCE, after its loops (range_flag) - a
  // compare + branch + atomic inside a row loop keeps the compiler from pipelining it (QKV epilogue: 66 -> 82 us)
  template <int N, bool NT = false>
  static TAMF_DEV void store_rc(elem_t* base, long idx, const float* v, float& am) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) am = fmaxf(fmaxf(am, fabsf(v[2 * i])), fabsf(v[2 * i + 1]));
    store<N, NT>(base, idx, v);
  }
  static TAMF_DEV void range_flag(float am, unsigned* status) { f16_range_flag(am, status); }
  static TAMF_DEV void store1(elem_t* base, long idx, float v) {
#pragma clang fp contract(off)
    char* p = (char*)base + byte_off(idx);
    const _Float16 hi = (_Float16)v;
    *(_Float16*)p = hi;
    *(_Float16*)(p + 64) = (_Float16)(v - (float)hi);
  }
  static TAMF_DEV float load1(const elem_t* base, long idx) {
    const char* p = (const char*)base + byte_off(idx);
    return (float)*(const _Float16*)p + (float)*(const _Float16*)(p + 64);
  }
};

// Position of key k inside a V^T row.  The P.V product takes P straight from the S^T accumulator layout, in which
// lane group g of a 32-key block holds keys {4g..4g+3} and {16+4g..16+4g+3}; for the bf16 modes V^T is therefore stored
// with those 8 keys adjacent (position 8g + 4*(k>>4 & 1) + (k & 3) inside the block), so that the V^T fragment of
// lane group g is the plain 16-byte chunk g of the row group - one ds_read_b128, same addressing as a GEMM fragment.
// f32 MFMAs consume one key per lane group per instruction and keep the natural order.
template <class Op>
TAMF_DEV int vt_key_pos(int k) {
  if constexpr (Op::PREC == 0) return k;
  else return (k & ~31) | (((k & 15) >> 2) << 3) | (((k >> 4) & 1) << 2) | (k & 3);
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
// Cross-lane reductions without LDS round trips (__shfl_xor compiles to ~6 VALU + ds_bpermute_b32 + an exposed LDS
// latency per step): DPP modifiers inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.
// `swap16(v)` returns (value of the even row, value of the odd row) of this lane's row pair in BOTH rows, `swap32(v)`
// (lower half, upper half) in both halves, so a commutative op of the two components is the xor-16 / xor-32 step.
template <int CTRL>
TAMF_DEV float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140 };
struct RedSum { static TAMF_DEV float op(float a, float b) { return a + b; } };
struct RedMax { static TAMF_DEV float op(float a, float b) { return fmaxf(a, b); } };
struct RedMin { static TAMF_DEV float op(float a, float b) { return fminf(a, b); } };
// over the 4 lane groups g = lane >> 4 (lanes l, l^16, l^32, l^48); every lane gets the result.
// (the two results are copied to scalars before the bit casts: __builtin_bit_cast straight from an element of the
// returned vector reads element 0 twice with hipcc 7.2 - tools/micro/reduce_test.hip checks these on the GPU)
template <class R>
TAMF_DEV float groups_reduce(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  unsigned lo = a[0], hi = a[1];
  v = R::op(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
  u = __builtin_bit_cast(unsigned, v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  lo = b[0];
  hi = b[1];
  return R::op(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}
// over all 64 lanes; every lane gets the result
template <class R>
TAMF_DEV float wave_reduce(float v) {
  v = R::op(v, dpp_mov<DPP_XOR1>(v));
  v = R::op(v, dpp_mov<DPP_XOR2>(v));
  v = R::op(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
  v = R::op(v, dpp_mov<DPP_ROW_MIRROR>(v));
  return groups_reduce<R>(v);
}
TAMF_DEV float wave_sum(float v) { return wave_reduce<RedSum>(v); }

// ---------------------------------------------------------------------------------------------
// Deferred LayerNorm (round 5; every arithmetic mode).
//
// The reference's encoder layer is post-LN: x = LayerNorm(u), u = x_prev + sublayer(x_prev) (interaction_segment_mdm.py:63-70, torch
// nn.TransformerEncoderLayer, eps 1e-5).  A LayerNorm needs whole rows, and no tiling of whole rows fills 256 CUs (rounds 2 - 4: the
// 64 x d LayerNorm-fused tile ran at 7 - 20 % of peak, the separate LayerNorm kernels moved 109 MB each).  So the normalised value is
// never materialised: the residual stream holds the UN-normalised sums u (fp32 + operand), the GEMM that produces a row block also
// leaves partial statistics of it, and every consumer applies the normalisation itself -
//   a GEMM over LN(u):     LN(u) . W^T = rstd[m] (u . W''^T) + c2[n],   W'' = W diag(gamma) (I - 1 1^T / d),  c2 = W beta + b
//                          (gain AND centring folded into the weights at tamf_finalize_weights - the rows of W'' sum to zero, so the
//                          row mean drops out of the product by itself; Epi*::ln);
//   the residual add:      u_next[m][n] = ((u[m][n] - mean[m]) rstd[m] gamma[n] + beta[n]) + (acc + bias)           (EpiResid).
// Statistics: the producer's epilogue writes, per row and per block of 32 columns, (S_b, Q_b) = (sum, sum of squares about the
// block's own mean); a consumer stages (mean, rstd) of its tile's rows in LDS before its K loop (ln_stage).  Both are fixed trees -
// the same for the LDS-walking 128 x 128 tiles and for the register epilogue of the clip tiles - so a clip's bits do not depend
// on the batch it is in or on the kernel selection (the batch-invariance tests).
// ---------------------------------------------------------------------------------------------
struct LnStats {
  const float2* part;  // [rows][NB] (S_b, Q_b); null: no LayerNorm in front of this row block (mean 0, rstd 1)
  int NB;              // d / 32: 4, 8 or 16
  float inv_d, eps;
};

// (S, Q) of 32 consecutive columns of one row, held as 8 consecutive values in each of 4 lanes: the lanes of a quad (LDS-walking
// epilogues: GROUPS = false) or the 4 lane groups l, l ^ 16, l ^ 32, l ^ 48 (register epilogue of the clip tiles: GROUPS = true).
// Every one of the 4 lanes gets the result.  One tree, written on PAIRS (w[i] = values 2i, 2i + 1: v_pk_add_f32 / v_pk_mul_f32 take two
// fp32 per instruction, and in f32 every VALU instruction of an epilogue is matrix-pipe time): lane sums
// ((v0 + v2) + (v4 + v6)) + ((v1 + v3) + (v5 + v7)), then (l0 + l1) + (l2 + l3); no fused multiply-adds (the same bits whatever code
// surrounds the call).
template <bool GROUPS>
TAMF_DEV float ln_sum4(float p) {
  if constexpr (GROUPS) {
    return groups_reduce<RedSum>(p);
  } else {
    p = p + dpp_mov<DPP_XOR1>(p);
    return p + dpp_mov<DPP_XOR2>(p);
  }
}
template <bool GROUPS>
TAMF_DEV float2 ln_block_partial(const tamf_f32x2 (&w)[4]) {
#pragma clang fp contract(off)
  const tamf_f32x2 s = (w[0] + w[1]) + (w[2] + w[3]);
  const float S = ln_sum4<GROUPS>(s.x + s.y);
  const float mb = S * 0.03125f;
  const tamf_f32x2 mb2 = {mb, mb};
  tamf_f32x2 q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const tamf_f32x2 dlt = w[j] - mb2;
    q[j] = dlt * dlt;
  }
  const tamf_f32x2 qq = (q[0] + q[1]) + (q[2] + q[3]);
  return make_float2(S, ln_sum4<GROUPS>(qq.x + qq.y));
}

// (mean, rstd) of rows [r0, r0 + rows) into out[0 .. rows) (LDS), by NT threads, 4 lanes (a quad) per row: lane q of the quad takes the
// blocks q NB/4 .. (q + 1) NB/4 - 1 in order, the quad combines as (q0 + q1) + (q2 + q3).  mean = S / d;
// M2 = sum_b (Q_b + 32 (S_b / 32 - mean)^2); rstd = 1 / sqrt(M2 / d + eps).  Rows past row_limit are clamped (their value is never used).
// AFF: what a GEMM behind the LayerNorm needs per row is staged instead - its factor ra = rstd ws (in .x), so that its epilogue
// is out = fma(acc, ra, c2[n]) (tamf_gemm.h: the centring is folded into the weight; ws = the power-of-two weight scale of the launch).
// MAXP = passes of NT / 4 rows that cover the tile: the partials of ALL passes are requested before the first is combined (one
// global-load latency per tile, not one per pass - the first form of this function cost the FFN1 launch 6 us).
struct LnRaw {
  float4 x, y;
};
template <int NT, bool AFF, int MAXP>
TAMF_DEV void ln_stage(const LnStats& s, float ws, int r0, int rows, int row_limit, float2* out, int tid) {
#pragma clang fp contract(off)
  static_assert(NT % 64 == 0, "whole waves");
  constexpr int RPP = NT / 4;
  const int q = tid & 3, rq = tid >> 2;
  if (!s.part) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
      if (q == 0 && rq + p * RPP < rows) out[rq + p * RPP] = AFF ? make_float2(ws, 0.f) : make_float2(0.f, 1.f);
    return;
  }
  const int nbq = s.NB >> 2;
  LnRaw raw[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    raw[p].x = raw[p].y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p * RPP < rows) {  // (uniform)
      int gr = r0 + rq + p * RPP;
      gr = gr < row_limit ? gr : row_limit - 1;
      const float2* ptr = s.part + (long)gr * s.NB + q * nbq;
      if (nbq == 4) {
        raw[p].x = *(const float4*)ptr;
        raw[p].y = *(const float4*)(ptr + 2);
      } else if (nbq == 2) {
        raw[p].x = *(const float4*)ptr;
      } else {
        const float2 t = *ptr;
        raw[p].x = make_float4(t.x, t.y, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p * RPP >= rows) break;  // (uniform: the DPP steps below see whole quads)
    const float sb[4] = {raw[p].x.x, raw[p].x.z, raw[p].y.x, raw[p].y.z};
    const float qb[4] = {raw[p].x.y, raw[p].x.w, raw[p].y.y, raw[p].y.w};
    float S = sb[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < nbq) S = S + sb[i];
    S = ln_sum4<false>(S);
    const float mean = S * s.inv_d;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nbq) {
        const float dm = sb[i] * 0.03125f - mean;
        m2 = m2 + (qb[i] + 32.0f * (dm * dm));
      }
    m2 = ln_sum4<false>(m2);
    const float rstd = 1.0f / sqrtf(m2 * s.inv_d + s.eps);
    const int r = rq + p * RPP;
    if (q == 0 && r < rows) out[r] = AFF ? make_float2(rstd * ws, 0.f) : make_float2(mean, rstd);
  }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller (restated in oracle/mdm_oracle.py:philox_normal)
// ---------------------------------------------------------------------------------------------
TAMF_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                            uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)c0 * 0xD2511F53u;
    const uint64_t p1 = (uint64_t)c2 * 0xCD9E8D57u;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The four normals of Philox block `blk` of clip `clip` at draw `draw`: counter (blk, draw, clip_lo, clip_hi), key
// (seed_lo, seed_hi); outputs (0,1) and (2,3) are two Box-Muller pairs.  Element e of the sampler state (frame-major,
// 128-padded: e = tau*128 + feature) is component e & 3 of block e >> 2.
TAMF_DEV void philox_normal4(uint64_t seed, int64_t clip, uint32_t draw, uint32_t blk, float (&z)[4]) {
  uint32_t r[4];
  philox4x32_10(blk, draw, (uint32_t)((uint64_t)clip & 0xFFFFFFFFu), (uint32_t)((uint64_t)clip >> 32),
                (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    const float u0 = __fadd_rn(__fmul_rn((float)r[2 * pair], 2.3283064365386963e-10f), 1.1641532182693481e-10f);
    const float u1 = __fadd_rn(__fmul_rn((float)r[2 * pair + 1], 2.3283064365386963e-10f), 1.1641532182693481e-10f);
    const float rad = sqrtf(__fmul_rn(-2.0f, logf(u0)));
    const float ang = __fmul_rn(6.283185307179586f, u1);
    z[2 * pair] = __fmul_rn(rad, cosf(ang));
    z[2 * pair + 1] = __fmul_rn(rad, sinf(ang));
  }
}
