// libtamf_hip.so - host side of the C-ABI declared in include/tamf_hip.h: context, weight repacking,
// step-invariant conditioning precompute, the per-step kernel sequence and its hipGraph replay loop.
// gfx950 (MI355X) only.
#include "../../include/tamf_hip.h"
#ifdef TAMF_TEST_HOOKS
#include "../../include/tamf_hip_test.h"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "tamf_attn.h"
#include "tamf_gemm.h"
#include "tamf_gemm_clip.h"
#include "tamf_gemm_deep.h"
#if defined(TAMF_BENCH) && !defined(TAMF_OVERLAP_PROBE)
#define TAMF_OVERLAP_PROBE  // (-DTAMF_OVERLAP_PROBE alone: the probe without the ablation code of -DTAMF_BENCH, which spills in some bf16 / f32 instantiations)
#endif
#ifdef TAMF_BENCH  // the row-block LayerNorm GEMM of round 4 (measured, not faster: DESIGN.md): an A/B partner of the measurement builds only
#endif
#include "tamf_geom.h"
#include "tamf_misc.h"

// run EXPR with `Op` bound to the operand traits of arithmetic mode `prec` (tamf_precision)
#define TAMF_WITH_OP(prec, EXPR)                                         \
  switch (prec) {                                                        \
    case TAMF_PREC_F32: { typedef OpF32 Op; EXPR; } break;               \
    case TAMF_PREC_BF16: { typedef OpBF16 Op; EXPR; } break;             \
    case TAMF_PREC_BF16X3: { typedef OpBF16X3 Op; EXPR; } break;         \
    default: { typedef OpF16X3 Op; EXPR; } break;                        \
  }

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static std::string g_noctx_err;

struct OperandBuf {
  void* p = nullptr;
  float inv_scale = 1.0f;  // f16x3 weights are stored scaled by a power of two (upload_operand); this is its inverse, applied by the epilogue
};

struct LayerW {
  OperandBuf Win, Wout, W1, W2;
  // deferred LayerNorm (tamf_device.h): Win / W1 hold W diag(gamma) (I - 1 1^T / d), the consumers take c2 = W beta + b
  // in place of the bias, the residual adds take (gamma, beta + bias) of the LayerNorm their residual passes through
  float *c2_in = nullptr, *c2_ff = nullptr;
  float *g_att = nullptr, *bb_att = nullptr, *g_ffn = nullptr, *bb_ffn = nullptr;
};

// What a captured step sequence depends on.  Seed, clip range, noise / dump tensors are NOT part of it: the kernels read
// them from the device-side LoopParams block, so one executable graph serves every loop of the same shape.
struct GraphKey {
  int B = -1, T = -1, n_steps = -1, steps_per_graph = 0;
  bool operator==(const GraphKey& o) const {
    return B == o.B && T == o.T && n_steps == o.n_steps && steps_per_graph == o.steps_per_graph;
  }
};

// Guard-band mode, for tests only (GPU AddressSanitizer is not available for gfx950 here): with tamf_test_set_guard_bytes(n > 0) every
// device allocation of contexts created afterwards is n bytes longer at both ends, the margins are filled with GUARD_BYTE, and
// tamf_test_check_guards() verifies them - a kernel that stores below or beyond its buffer (the V^T overrun of round 4) changes a margin
// instead of a neighbouring live buffer, where a later kernel would have overwritten the evidence.
struct GuardRec {
  char* base;
  size_t bytes, guard;
  const char* tag;
  bool ws;
};
static constexpr unsigned char GUARD_BYTE = 0xA5;
static std::atomic<size_t> g_guard_bytes{0};
static thread_local const char* g_alloc_tag = "";

struct tamf_ctx {
  tamf_arch arch{};
  int prec = 0, device = 0, Bmax = 0, Tmax = 0;
  int d = 0, ff = 0, L = 0, H = 0, hd = 0, P = 0, F = 0, has_t = 0;
  int XK = 0;   // padded K of the fused input GEMM
  int XN = 128; // padded N of the output head
  int EB = 4;  // bytes per logical operand element (f32 4, bf16 2, bf16x3 4 = hi + lo)
  std::string err;
  std::vector<void*> allocs;     // weights, tables, schedule: live as long as the context
  std::vector<void*> ws_allocs;  // workspaces dimensioned by (max_batch, max_frames): re-made by tamf_ctx_resize
  bool alloc_ws = false;         // dev_alloc files its allocation under ws_allocs
  std::vector<GuardRec> guards;  // guard-band mode (tamf_test_set_guard_bytes): one record per allocation
  std::map<std::string, std::vector<float>> raw;
  std::map<std::string, std::vector<int64_t>> raw_shape;
  bool finalized = false, cond_set = false;
  int n_t = 0;  // rows of the timestep table
  // schedule
  int n_steps = 0;
  std::vector<float> h_c1, h_c2, h_sigma;
  float *c1 = nullptr, *c2 = nullptr, *sigma = nullptr;
  // weights
  std::vector<LayerW> layers;
  OperandBuf Wfused, Wm2, Wf;
  float *bm2 = nullptr, *bf = nullptr;
  float* pe = nullptr;    // [5000][d]
  float* temb = nullptr;  // [n_t][d]
  // respaced sampling (tamf_set_timestep_map): row i = temb[map[i]], i = the loop's own step index; the loop's kernels read this table
  // instead of temb while a map is set (a single evaluation - tamf_denoise - always indexes temb with the caller's timesteps)
  float* temb_loop = nullptr;
  int temb_loop_cap = 0;
  bool tmap_on = false, in_loop = false;
  int tmap_max = -1;
  OperandBuf Wt1_f32, Wt2_f32;
  float *bt1 = nullptr, *bt2 = nullptr;
  // conditioning (tamf_misc.h, prefix_rows_kernel / cobj_kernel): transposed weights W^T [K][d]; WcT / bc = input_merge.0[:, d:2d] composed with
  // obj_input_process.poseEmbedding (and the fused bias of input_merge.0)
  float *WtxtT = nullptr, *btxt = nullptr, *WshapeT = nullptr, *bshape = nullptr, *WobjT = nullptr, *bobj = nullptr;
  float *WcT = nullptr, *bc = nullptr, *rh = nullptr, *lh = nullptr;
  // activations
  int B = 0, T = 0, S = 0, Sp = 0, Skp = 0, M = 0;
  long Mmax = 0;
  float *xs = nullptr, *cobj = nullptr, *X = nullptr, *pstatic = nullptr;
  OperandBuf xs_op, h1_op, X_op, QK_op, Vt_op, A_op, H_op;
  // f32: an fp32 operand matrix [rows][ld] is byte for byte the fp32 tensor, so the residual stream X and the sampler state xs ARE
  // their own operands (X_op.p = X, xs_op.p = xs) and the kernels that used to write both copies write one (X_st / xs_st = null):
  // 27 MB less per LayerNorm at B = 64 (round 4)
  void *X_st = nullptr, *xs_st = nullptr;
  int* tcur = nullptr;
  unsigned* status = nullptr;  // this context's sticky status word (tamf_device.h): written by its kernels only
  unsigned char* side_dev = nullptr;
  int* objnum_dev = nullptr;  // per-clip object counts of tamf_set_cond_ragged
  unsigned host_status = 0;  // status bits raised on the host (TAMF_STATUS_F16_WEIGHT_RANGE at tamf_finalize_weights); never cleared
  std::string weight_note;   // ... and which tensor raised it
  // deferred LayerNorm: partial row statistics of the residual stream after the attention / feed-forward sublayer, [Mmax][d / 32]
  float2 *stat_att = nullptr, *stat_ffn = nullptr;
  // graph
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  GraphKey graph_key;
  hipEvent_t graph_done = nullptr;  // recorded after the last launch of graph_exec: waited for before it is destroyed
  bool graph_in_flight = false;
  LoopParams* loop_params = nullptr;
  int sched_cap = 0;  // allocated length of c1 / c2 / sigma
  int graph_captures = 0, graph_launches_last_loop = 0;  // tamf_loop_stats
#ifdef TAMF_OVERLAP_PROBE
  // overlap probe (selection bit 256, measurement builds only): see PingPong
  std::vector<hipEvent_t> pp_ev;
  long pp_count = 0;
  bool pp_on = false;
#endif
  int step_kernels = 0;
  // per-launch profiling (tamf_step_profile)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<std::string> prof_names;
  std::vector<double> prof_flops;
};

static std::vector<tamf_ctx*> g_live_ctx;  // contexts of this process (tamf_set_gemm_tuning retires their captured graphs)
static std::mutex g_live_mu;                // (contexts may be created / destroyed from different threads)
// The tuning / selection words (g_krot, g_sel) are process-global and a captured graph has them baked in.  Every entry point that
// enqueues kernels or touches a context's graph holds this lock for its (host-side, microseconds) duration, and so does
// tamf_set_gemm_tuning: a thread inside loop_impl never sees a half-updated selection or a graph being destroyed under it.
static std::recursive_mutex g_launch_mu;
// Round 6: that serialisation exists ONLY in the hooks build (-DTAMF_TEST_HOOKS), where tamf_set_gemm_tuning can change the words.  The
// product library has no setter - g_krot / g_sel are constants there - so its entry points take no process-wide lock: contexts are
// independent (one thread per context, as the header says), the lazily set per-device "kernel attributes prepared" flags and the
// workgroup-slot table are atomics, and the list of live contexts has its own mutex.  Context creation still serialises (TAMF_STATE_LOCK).
#ifdef TAMF_TEST_HOOKS
#define TAMF_LAUNCH_LOCK std::lock_guard<std::recursive_mutex> launch_lock_(g_launch_mu)
#else
#define TAMF_LAUNCH_LOCK ((void)0)
#endif
#define TAMF_STATE_LOCK std::lock_guard<std::recursive_mutex> state_lock_(g_launch_mu)

static int fail(tamf_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  else g_noctx_err = msg;
  return code;
}
#define HIPCHK(ctx, call)                                                                                         \
  do {                                                                                                            \
    hipError_t e_ = (call);                                                                                       \
    if (e_ != hipSuccess)                                                                                         \
      return fail(ctx, TAMF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" +  \
                                         std::to_string(__LINE__) + ")");                                        \
  } while (0)
#define TRY(expr)           \
  do {                      \
    int rc_ = (expr);       \
    if (rc_ != 0) return rc_; \
  } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static std::atomic<int> g_fail_alloc_in{-1};  // test hook (tamf_test_fail_alloc_after): the n-th device allocation from now fails
static int dev_alloc(tamf_ctx* ctx, void** p, size_t bytes, bool zero = false) {
  if (bytes == 0) bytes = 16;
  const size_t guard = g_guard_bytes.load();
  if (g_fail_alloc_in.load() >= 0 && g_fail_alloc_in.fetch_sub(1) == 0)
    return fail(ctx, TAMF_ERR_NOMEM, "hipMalloc failed: injected by tamf_test_fail_alloc_after");
  hipError_t e = hipMalloc(p, bytes + 2 * guard);
  if (e != hipSuccess) return fail(ctx, TAMF_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  (ctx->alloc_ws ? ctx->ws_allocs : ctx->allocs).push_back(*p);
  if (guard) {
    char* base = (char*)*p;
    HIPCHK(ctx, hipMemset(base, GUARD_BYTE, guard));
    HIPCHK(ctx, hipMemset(base + guard + bytes, GUARD_BYTE, guard));
    ctx->guards.push_back(GuardRec{base, bytes, guard, g_alloc_tag, ctx->alloc_ws});
    *p = base + guard;
  }
  if (zero) HIPCHK(ctx, hipMemset(*p, 0, bytes));
  return 0;
}
template <class T>
static int dev_upload(tamf_ctx* ctx, T** p, const T* host, size_t n) {
  TRY(dev_alloc(ctx, (void**)p, n * sizeof(T)));
  HIPCHK(ctx, hipMemcpy(*p, host, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// host float -> bf16 (round to nearest even; NaN kept quiet)
static inline uint16_t h_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float h_bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// upload host fp32 [N][K] as an operand matrix [N][ldk] in precision `prec` (cols >= K zero; ldk % 32 == 0).
// bf16x3 rows are 128-byte groups of 32 elements: [hi: 32 bf16 | lo: 32 bf16] (tamf_device.h "Operand traits").
static int upload_operand(tamf_ctx* ctx, int prec, const float* w, int N, int K, int ldk, OperandBuf* out, const char* what) {
  const size_t n = (size_t)N * ldk;
  if (ldk % 32) return fail(ctx, TAMF_ERR_INVALID, "operand leading dimension must be a multiple of 32");
  int wexp = 0;  // f16x3: the tensor is stored as w * 2^wexp
  out->inv_scale = 1.0f;
  if (prec == TAMF_PREC_F16X3) {
    // Power-of-two pre-scaling of the split-fp16 weights: max |w| lands in [2^14, 2^15), so that the lo planes of everything down to
    // 2^-17 of the tensor's maximum are NORMAL fp16 numbers (22 significand bits; unscaled, PyTorch-default weights of ~0.04 had
    // subnormal lo parts: an absolute 3e-8, i.e. ~19 bits) and a weight beyond the fp16 range is no reason to refuse a checkpoint.
    // The product is scaled back exactly in the epilogue's bias add (EpiCtl::wscale).  Only a non-finite weight is refused.
    // (activations are checked on the device: tamf_device.h, status word)
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)N * K; ++i) {
      if (!std::isfinite(w[i]))
        return fail(ctx, TAMF_ERR_RANGE, std::string("f16x3: weight ") + what + " holds a non-finite value and cannot be stored as split-fp16 operands; use bf16x3 or f32");
      mx = std::fmax(mx, std::fabs(w[i]));
    }
    if (mx > 0.f) {
      int e = 0;
      (void)std::frexp(mx, &e);  // mx = m 2^e, m in [0.5, 1)
      wexp = std::min(120, std::max(-120, 15 - e));
    }
#ifdef TAMF_NO_PRESCALE  // (A/B builds: the unscaled split of rounds 2 - 3; weights beyond 65504 then turn into inf)
    wexp = 0;
#endif
    out->inv_scale = std::ldexp(1.0f, -wexp);
    // One power of two per tensor: a weight smaller than 2^-17.5 of the tensor's maximum has an fp16-SUBNORMAL lo part again (fewer than
    // 22 significand bits).  Ordinary tensors are nowhere near (max / typical ~ 2^3 .. 2^7); one huge outlier over ordinary weights is
    // (a single 7e4 among 0.04s leaves the others ~18 bits).  Reported, not refused: a status bit + the tensor's name.
    size_t small = 0, nonzero = 0;
    for (size_t i = 0; i < (size_t)N * K; ++i)
      if (w[i] != 0.f) {
        ++nonzero;
        if (std::fabs(std::ldexp(w[i], wexp)) < 0.125f) ++small;
      }
    if (nonzero && small * 100 > nonzero) {
      ctx->host_status |= TAMF_STATUS_F16_WEIGHT_RANGE;
      if (ctx->weight_note.empty())
        ctx->weight_note = std::string("f16x3: ") + std::to_string(small * 100 / nonzero) + " % of the non-zero weights of " + what +
                           " are below 2^-17.5 of the tensor's largest magnitude: stored with fewer than 22 significand bits (an outlier dominates the per-tensor scale)";
    }
  }
  if (prec == TAMF_PREC_F32) {
    std::vector<float> h(n, 0.f);
    for (int r = 0; r < N; ++r) memcpy(&h[(size_t)r * ldk], &w[(size_t)r * K], (size_t)K * 4);
    return dev_upload(ctx, (float**)&out->p, h.data(), n);
  }
  if (prec == TAMF_PREC_BF16) {
    std::vector<uint16_t> h(n, 0);
    for (int r = 0; r < N; ++r)
      for (int k = 0; k < K; ++k) h[(size_t)r * ldk + k] = h_f2bf(w[(size_t)r * K + k]);
    return dev_upload(ctx, (uint16_t**)&out->p, h.data(), n);
  }
  std::vector<uint16_t> h(n * 2, 0);
  for (int r = 0; r < N; ++r)
    for (int k = 0; k < K; ++k) {
      const float v = w[(size_t)r * K + k];
      const size_t idx = (size_t)r * ldk + k;
      const size_t o = (idx >> 5) * 64 + (idx & 31);  // in uint16 units: 64 per 128-byte group
      if (prec == TAMF_PREC_F16X3) {
        const float vs = std::ldexp(v, wexp);  // exact
        const _Float16 hi = (_Float16)vs, lo = (_Float16)(vs - (float)hi);
        memcpy(&h[o], &hi, 2);
        memcpy(&h[o + 32], &lo, 2);
      } else {
        const uint16_t hi = h_f2bf(v);
        h[o] = hi;
        h[o + 32] = h_f2bf(v - h_bf2f(hi));
      }
    }
  return dev_upload(ctx, (uint16_t**)&out->p, h.data(), n * 2);
}

// ------------------------------------------------------------------------------------------------
// kernel launchers
// ------------------------------------------------------------------------------------------------
// GemmArgs::krot bits (tamf_gemm.h): L2 touch-prefetch distance, XCD arrangement.  -1 = per-kernel default: the 64-row
// LayerNorm tiles prefetch 4 K tiles ahead into L2 (their weight panel is shared by every workgroup and has been evicted
// from the 4 MB L2 by the other GEMMs of the layer: 186 -> 153 us in situ), the 128 x 128 tiles do not (it costs them
// 5-8 %).  Only the kernel benchmark hook (tamf_bench_gemm) overrides it.
static int g_krot = -1;
// kernel-selection overrides for A/B measurements (tamf_set_gemm_tuning bits 20..): 1 = no clip tiles at all,
// 2 = FFN2 / out-proj on the 128 x 128 tiles, 4 = attention split once more, 8 = FFN1 on the 128 x 128 tiles,
// 16 = residual GEMMs of a few clips on the clip / 128 x 128 tiles too (no 32- / 64-row tiles), 64 = f32: QKV on the 128 x 128 tiles,
// 32 = FFN1 (whole-clip tiles in two or more exact rounds): column-split rounds toggled against TAMF_CLIP_COLSPLIT_DEFAULT,
// 256 = (-DTAMF_OVERLAP_PROBE / -DTAMF_BENCH builds only) OVERLAP PROBE: the launches of a no-graph loop alternate between two streams with no data
//       dependency enforced - garbage samples, the time is an upper bound of what removing the kernel boundaries could gain (PingPong),
// 512 = streaming attention kernel in the 16-bit modes too,
// 1024 = clip tiles from 50 % (not 74 %) of the workgroup slots of their rounds.
// (Rounds 2 - 5 also had 16 (another meaning) / 256 / 2048 / 4096: the LayerNorm-fused 64 x d tile, the GEMM + LayerNorm-kernel form and the row-block kernel
// of the residual GEMMs - gone with the deferred LayerNorm, which every mode uses now; 32 / 128: clip-tile variants of FFN1 / QKV that
// lost their A/B.  The bits are accepted and ignored.)
static int g_sel = 0;
#ifndef TAMF_CLIP_COLSPLIT_DEFAULT  // FFN1's rounds split by columns (selection bit 32 toggles it against this default: A/B)
#define TAMF_CLIP_COLSPLIT_DEFAULT 0
#endif
static inline int krot_for(bool ln_tile) { return g_krot >= 0 ? g_krot : (ln_tile ? (4 << 8) : 0); }

// resident workgroup slots of the chip for the 2-per-CU tiles (MI355X: 256 CUs); one "round" of a launch.  Per DEVICE (one
// process-wide word followed the device of the context created last): written by tamf_ctx_create under the launch lock, read for the
// calling thread's current device - every entry point that enqueues kernels has made its context's device current.
static std::atomic<int> g_wg_slots_dev[64];
static inline int wg_slots() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int v = g_wg_slots_dev[dev & 63].load(std::memory_order_relaxed);
  return v > 0 ? v : 512;
}

template <class Op, int BM, int BN, class Epi, bool CAN_SPLIT = false>
struct GemmLaunch {
  static constexpr int SMEM = GemmSmem<BM, BN>::TOTAL;
  // wave grid: the 64-row LayerNorm tiles own a CU and run 8 waves (2 x 4), the 64 x 512 one 16 waves (2 x 8: four waves
  // per SIMD cover each other's LDS / barrier latency, 36.6 -> 35.1 us for out-proj); the 128 x 128 tiles run 8 waves
  // (4 x 2) with two workgroups per CU
#ifndef TAMF_WGM128  // waves along M of the 128 x 128 tiles: 4 = 8 waves per workgroup (4 x 2), two workgroups per CU = 4 waves per SIMD.
#define TAMF_WGM128 4  // Round 3, two builds alternating on one box: QKV 62.8 -> 60.7 us (f16x3), 31.5 -> 29.6 us (bf16) against the 2 x 2
#endif                 // grid of rounds 1 - 2; the whole step -1 %: a workgroup left alone on its CU during its partner's epilogue keeps 2 waves per SIMD
  static constexpr int WGN = (BM <= 64) ? (BN == 512 ? 8 : 4) : 2;
  static constexpr int WGM = (BM <= 64 || Op::PREC == 0) ? 2 : TAMF_WGM128;  // (f32: the 2 x 2 grid stays - input_merge.2 78 us against 91 us with 4 x 2)
  template <int SPLIT>
  static hipError_t prepare1() {
    return hipFuncSetAttribute((const void*)gemm_kernel<Op, BM, BN, WGM, WGN, Epi, SPLIT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }
  static hipError_t prepare() {  // kernel attributes are per device
    static std::atomic<bool> done[64];  // (zero-initialised; relaxed: the attribute calls are idempotent)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = prepare1<1>();
    if constexpr (CAN_SPLIT) {
      if (e == hipSuccess) e = prepare1<2>();
      if (e == hipSuccess) e = prepare1<4>();
    }
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return e;
  }
  static hipError_t launch(const GemmArgs<Op>& ga, const Epi& epi, hipStream_t st) {
    hipError_t e = prepare();
    if (e != hipSuccess) return e;
    if ((ga.K * Op::EB) % GEMM_BKB != 0 || ga.N % BN != 0 || ga.M <= 0) return hipErrorInvalidValue;
    const int ntn = ga.N / BN, ntm = (ga.M + BM - 1) / BM, tiles = ntn * ntm;
    GemmArgs<Op> gb = ga;
    gb.krot = krot_for(BM <= 64 && BN >= 256);
    // wide-N launches (FFN1: 16 column tiles): XCDs in a 4 x 2 arrangement over the tile grid - each L2 then serves half of W
    // instead of all of it (FETCH_SIZE 162 -> 137 MB per launch, time unchanged); with few column tiles it would re-read A
    if (g_krot < 0 && BM == 128 && ntn >= 8) gb.krot |= 0x10000;  // (also picked up by the persistent grid below: QKV, round 4)
    gb.n_full = tiles;
    int split = 1;
    if constexpr (CAN_SPLIT) {
      // tiles left over after the last full round: cut them 2- or 4-ways while every slice still gets a CU to itself
      // (measured: slices sharing a CU take as long as the whole tiles did, FFN1 93.5 -> 89 us with 2 x 128 slices)
      // (a launch of less than one round is sliced as long as the slices fit the workgroup slots: two slices sharing a CU
      // overlap each other's latencies, one whole tile alone on a CU does not)
      const int n_full = (tiles / wg_slots()) * wg_slots(), rem = tiles - n_full, cus = n_full ? wg_slots() / 2 : wg_slots();
      if (rem > 0) {
        split = (rem * 4 <= cus) ? 4 : (rem * 2 <= cus) ? 2 : 1;
        if (split > 1) gb.n_full = n_full;
      }
    }
    gb.n_tiles = 0;
    int nblk = gb.n_full + (tiles - gb.n_full) * split;
    // more than one round, a partial last round and no slices: a persistent one-round grid balances the CUs (the hardware
    // hands a freed slot to the next workgroup greedily; QKV's 1248 tiles ended up as 4..6 per CU instead of 4..5)
    if (split == 1 && BM == 128 && tiles > wg_slots() && tiles % wg_slots() != 0) {
      gb.n_tiles = tiles;
      nblk = wg_slots();
    }
    const dim3 grid(nblk), block(WGM * WGN * 64);
    if constexpr (CAN_SPLIT) {
      if (split == 4) { hipLaunchKernelGGL((gemm_kernel<Op, BM, BN, WGM, WGN, Epi, 4>), grid, block, SMEM, st, gb, epi); return hipGetLastError(); }
      if (split == 2) { hipLaunchKernelGGL((gemm_kernel<Op, BM, BN, WGM, WGN, Epi, 2>), grid, block, SMEM, st, gb, epi); return hipGetLastError(); }
    }
    hipLaunchKernelGGL((gemm_kernel<Op, BM, BN, WGM, WGN, Epi, 1>), grid, block, SMEM, st, gb, epi);
    return hipGetLastError();
  }
};
// the wide-N GEMMs (QKV, FFN1, input merge) may split their left-over tiles
template <class Epi> struct EpiCanSplit { static constexpr bool value = false; };
template <class Op, bool LN> struct EpiCanSplit<EpiBiasAct<Op, LN>> { static constexpr bool value = true; };
template <class Op, bool LN> struct EpiCanSplit<EpiQKV<Op, LN>> { static constexpr bool value = true; };
template <> struct EpiCanSplit<EpiStoreF32> { static constexpr bool value = true; };

// Small tiles with a deep K pipeline (tamf_gemm_deep.h): launches of a few tiles, one workgroup each
template <class Op, int BM, int BN, int NSTG, class Epi, int WGM = 2, int WGN = 4>
struct GemmDeepLaunch {
  static constexpr int SMEM = GemmSmemDeep<BM, BN, NSTG>::TOTAL;
  static hipError_t prepare() {
    static std::atomic<bool> done[64];  // (zero-initialised; relaxed: the attribute calls are idempotent)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_deep_kernel<Op, BM, BN, WGM, WGN, NSTG, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return e;
  }
  static hipError_t launch(const GemmArgs<Op>& ga, const Epi& epi, hipStream_t st) {
    hipError_t e = prepare();
    if (e != hipSuccess) return e;
    if ((ga.K * Op::EB) % GEMM_BKB != 0 || ga.N % BN != 0 || ga.M <= 0) return hipErrorInvalidValue;
    const int tiles = (ga.N / BN) * ((ga.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_deep_kernel<Op, BM, BN, WGM, WGN, NSTG, Epi>), dim3(tiles), dim3(WGM * WGN * 64), SMEM, st, ga, epi);
    return hipGetLastError();
  }
};

// A few clips per call (one, in the reference's own launcher: launch/sample.py:202-229): the tiles of the big batches would put a handful of
// workgroups on the chip and each would walk its K tiles one L2 round trip at a time.  32 x 128 tiles (64 x 128 while they still fit one
// workgroup per CU) with a four-stage K pipeline instead: a CU on every 32 rows, the same K order per element - the same bits.
// Returns false when the launch is not that small (or selection bit 1 / 16 turns the small tiles off: the A/B partner).
template <class Op, class Epi>
static bool small_m_launch(const GemmArgs<Op>& ga, const Epi& ep, hipStream_t st, hipError_t* e) {
  if ((g_sel & (1 | 16)) || ga.N % 128 != 0) return false;
  const int cus = wg_slots() / 2, ntn = ga.N / 128;
  if (((ga.M + 31) / 32) * ntn <= cus) {
    *e = GemmDeepLaunch<Op, 32, 128, 4, Epi>::launch(ga, ep, st);
    return true;
  }
  if (((ga.M + 63) / 64) * ntn <= cus) {
    *e = GemmDeepLaunch<Op, 64, 128, 4, Epi>::launch(ga, ep, st);
    return true;
  }
  return false;
}

// Clip-aligned tiles (tamf_gemm_clip.h): one M tile = one clip of NSUB MFMA row tiles.  NSUB = 13 serves the bench shape (T = 196:
// S = 201, Sp = 208), NSUB = 11 the length the reference's dataset emits (slice_max_len = 160, dataset/interaction_segment.py:291:
// S = 165, Sp = 168 - the 11th row tile holds 8 rows, the rest of it is clamped on the load side and skipped on the store side);
// a clip that is up to one row tile shorter than the template runs on it with that tile wasted.  Used when K gives an even number
// of K tiles and the tile count fills the chip's rounds well enough; everything else runs on the 128 x 128 tiles.
// row tiles of the X waves (the K tile's loaders): they carry the DMA issue, so they get fewer of the row tiles - 6 of 13 / 5 of 11
// at 256 columns (5 : 8 spills the Y waves), 2 at 128 columns (a loader wave is blocked ~850 of the ~1 900 cycles of a 128-column
// interval while its pieces queue; FFN2 at B = 64: 73.7 us with 4 : 9, 69.5 us with 2 : 11)
#ifndef TAMF_CLIP_XSUB_N2  // (build-time knobs of the A/B runs)
#define TAMF_CLIP_XSUB_N2 2
#endif
#ifndef TAMF_CLIP_XSUB_N4  // X row tiles of a 13-row-tile clip at 256 columns; an 11-row-tile clip gets one less
#define TAMF_CLIP_XSUB_N4 6
#endif
#ifndef TAMF_CLIP_XSUB_PARTS  // X : Y row tiles of the 7- / 6-row-tile parts (A/B knob; FFN2 at B = 32: 2 : 5 46 us, 3 : 4 60 us, 4 : 3 slower still)
#define TAMF_CLIP_XSUB_PARTS 2
#endif
#ifndef TAMF_CLIP_YSUB_F32_N4  // f32 at 256 columns: row tiles of the Y waves (6: they read the next K tile's fragments inside their MFMA stream,
#define TAMF_CLIP_YSUB_F32_N4 6  // clip_mma_read_y, which does not fit the registers with 7; in f32 nothing of X overlaps Y's MFMAs, so the split is free)
#endif
template <int NI, int NSUB, int PARTS, int PREC>
struct ClipXsub {
  static constexpr int value = PARTS > 1 ? TAMF_CLIP_XSUB_PARTS
                               : NI >= 4 ? (PREC == 0 ? NSUB - TAMF_CLIP_YSUB_F32_N4 : TAMF_CLIP_XSUB_N4 - (13 - NSUB + 1) / 2)
                                         : TAMF_CLIP_XSUB_N2;
};
// NSUB: row tiles of a TILE.  PARTS = 1: the tile is a whole clip; PARTS = 2: every clip is cut into its first NSUB row tiles and
// the rest (NSUB or NSUB - 1 of them), each a tile of its own - for launches whose whole-clip tiles would fill at most half of the CUs
template <class Op, int NI, class Epi, int NSUB = 13, int PARTS = 1>
struct ClipLaunch {
  static constexpr int XSUB = ClipXsub<NI, NSUB, PARTS, Op::PREC>::value;
  typedef ClipCfg<NSUB, NI, XSUB, Epi::LANE_CHUNK> C;
  static constexpr int CLIP_ROWS_MAX = PARTS * C::MT;  // padded rows of the longest clip these tiles hold
  static hipError_t prepare() {
    static std::atomic<bool> done[64];  // (zero-initialised; relaxed: the attribute calls are idempotent)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)clip_gemm_kernel<Op, NSUB, NI, XSUB, Epi>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::BYTES);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return e;
  }
  static bool shape_ok(int Sp, int N, int K) {
    if (g_sel & 1) return false;  // kernel benchmark hook: force the 128 x 128 tiles (A/B runs)
    if (PARTS == 1 ? (Sp > C::MT || Sp <= C::MT - 32) : (Sp > 2 * C::MT || Sp <= 2 * C::MT - 32)) return false;
    if (N % C::BN != 0 || (K * Op::EB) % GEMM_BKB != 0) return false;
    const int KT = (K * Op::EB) / GEMM_BKB;
    return KT >= 2 && !(KT & 1);
  }
  static bool applies(int n_clips, int Sp, int N, int K, int min_util = 74) {
    static_assert(PARTS == 1, "whole-clip tiles");
    if (!shape_ok(Sp, N, K)) return false;
    const int cus = wg_slots() / 2, tiles = n_clips * (N / C::BN), rounds = (tiles + cus - 1) / cus;
    return tiles * 100 >= rounds * cus * ((g_sel & 1024) ? 50 : min_util);  // >= 74 % of the workgroup slots of its rounds are used (A/B: 50 %)
  }
  static hipError_t launch(const Op*, const typename Op::elem_t* A, int lda, const typename Op::elem_t* W, int ldw, int n_clips,
                           int Sp, int N, int K, const Epi& epi, hipStream_t st) {
    if (!shape_ok(Sp, N, K)) return hipErrorInvalidValue;  // (the kernel relies on it: fewer than 32 padding rows per tile, clip_issue)
    hipError_t e = prepare();
    if (e != hipSuccess) return e;
    static_assert(PARTS == 1 || PARTS == 2, "whole clips or their two row parts");
    constexpr int split_rows = PARTS == 1 ? 0 : NSUB * 16;
    ClipGemmArgs<Op> ga{A, lda, W, ldw, n_clips, Sp, N, K, PARTS * n_clips * (N / C::BN), split_rows, 0,
                        g_krot >= 0 ? ((g_krot >> 12) & 15) | (((g_krot >> 17) & 3) << 4) : 0};
    const int cus = wg_slots() / 2;
    {  // column-split rounds (tamf_gemm_clip.h, ClipGemmArgs::colsplit): whole clips, two or more EXACT rounds, column tiles divisible
      const int ntn = N / C::BN, rounds = ga.n_tiles / cus;
      const bool fits = PARTS == 1 && ga.n_tiles > cus && ga.n_tiles % cus == 0 && ntn % rounds == 0 && NI == 4;
      if (fits && (TAMF_CLIP_COLSPLIT_DEFAULT != 0) != ((g_sel & 32) != 0)) ga.colsplit = ntn;
    }
    hipLaunchKernelGGL((clip_gemm_kernel<Op, NSUB, NI, XSUB, Epi>), dim3(ga.n_tiles < cus ? ga.n_tiles : cus), dim3(512), C::BYTES, st, ga, epi);
    return hipGetLastError();
  }
  // row-part tiles: when whole-clip tiles would use at most half of the CUs and the parts fit one round
  static bool applies_parts(int n_clips, int Sp, int N, int K) {
    static_assert(PARTS == 2, "row-part tiles");
    if (!shape_ok(Sp, N, K) || Sp <= C::MT) return false;
    const int cus = wg_slots() / 2, whole = n_clips * (N / C::BN);
    return whole * 2 <= cus;
  }
};
// Bind NS (whole-clip row tiles: 13 or 11) and NSP (row tiles of the first of two row parts: 7 or 6) for a clip of Sp padded rows
// and run the statement(s) that follow Sp_ (variadic: template argument lists carry commas); nothing runs for other clip lengths
#define TAMF_CLIP_NSUB(Sp_, ...)                                                                \
  if ((Sp_) > 176 && (Sp_) <= 208) { constexpr int NS = 13, NSP = 7; (void)NSP; (void)NS; __VA_ARGS__; } \
  else if ((Sp_) > 144 && (Sp_) <= 176) { constexpr int NS = 11, NSP = 6; (void)NSP; (void)NS; __VA_ARGS__; }

// share of the workgroup slots of its rounds that a launch of `tiles` workgroups uses
static inline double round_util(int tiles, int slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); }

// 128 x 128 tiles, or 64 x 128 tiles (8 waves) for the short-K launches that fill at most half of the workgroup slots (the
// input-merge GEMMs at <= 32 clips: 20.7 against 23.4 us at B = 32).  Measured and left on the big tiles: QKV (624 tiles at
// B = 32: 43 against 37 us) and FFN2 (K = 2048: 58 against 56 us).
template <class Epi> struct EpiHasSmallTile { static constexpr bool value = false; };
template <class Op> struct EpiHasSmallTile<EpiBiasAct<Op>> { static constexpr bool value = true; };
template <class Op> struct EpiHasSmallTile<EpiSeqRows<Op>> { static constexpr bool value = true; };
template <class Op, class Epi>
static hipError_t gemm128(const GemmArgs<Op>& ga, const Epi& epi, hipStream_t st) {
  if constexpr (EpiHasSmallTile<Epi>::value) {
    const int t128 = ((ga.M + 127) / 128) * (ga.N / 128);
    if (!(g_sel & 1) && ga.N % 128 == 0 && t128 * 2 <= wg_slots() && ga.K * Op::EB <= 2048)
      return GemmLaunch<Op, 64, 128, Epi>::launch(ga, epi, st);
  }
  return GemmLaunch<Op, 128, 128, Epi, EpiCanSplit<Epi>::value>::launch(ga, epi, st);
}
// the clip-tile kernels of one clip length (NS whole-clip row tiles, NSP row tiles of a row part)
template <class Op, int NS, int NSP>
static hipError_t prepare_clip() {
  hipError_t e;
  // the encoder layers: FFN1 with the row factors of the deferred LayerNorm, the two residual GEMMs (whole clips / row parts)
  if ((e = ClipLaunch<Op, 4, EpiBiasAct<Op, true>, NS>::prepare()) != hipSuccess) return e;
  if ((e = ClipLaunch<Op, 2, EpiResid<Op>, NS>::prepare()) != hipSuccess) return e;
  if ((e = ClipLaunch<Op, 2, EpiResid<Op>, NSP, 2>::prepare()) != hipSuccess) return e;
  if constexpr (Op::PREC == 0) {  // f32: the QKV projection on clip tiles (Q | K columns; V columns transposed)
    if ((e = ClipLaunch<Op, 2, EpiQK<Op, true>, NS>::prepare()) != hipSuccess) return e;
    if ((e = ClipLaunch<Op, 4, EpiQK<Op, true>, NS>::prepare()) != hipSuccess) return e;
    if ((e = ClipLaunch<Op, 2, EpiVt<Op, true>, NS>::prepare()) != hipSuccess) return e;
  }
  // plain fp32 product (tamf_test_gemm, tamf_bench_gemm)
  if ((e = ClipLaunch<Op, 2, EpiStoreF32, NS>::prepare()) != hipSuccess) return e;
  if ((e = ClipLaunch<Op, 2, EpiStoreF32, NSP, 2>::prepare()) != hipSuccess) return e;
  return hipSuccess;
}
template <class Op>
static hipError_t prepare_all() {
  hipError_t e;
  if ((e = GemmLaunch<Op, 128, 128, EpiBiasAct<Op>, true>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiQKV<Op>, true>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiSeqRows<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 64, 128, EpiHead<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiStoreF32, true>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiBiasAct<Op, true>, true>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiQKV<Op, true>, true>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 128, 128, EpiResid<Op>>::prepare()) != hipSuccess) return e;
  // a few clips per call (small_m_launch): the encoder layers' GEMMs on 32- / 64-row tiles with the deep K pipeline
  if ((e = GemmDeepLaunch<Op, 64, 128, 4, EpiResid<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 128, 4, EpiResid<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 64, 6, EpiResid<Op>, 2, 2>::prepare()) != hipSuccess) return e;
  if constexpr (Op::PREC != 0) {  // 20 - 39 clips per call, 16-bit modes (launch_resid)
    if ((e = GemmDeepLaunch<Op, 64, 128, 3, EpiResid<Op>>::prepare()) != hipSuccess) return e;
  }
  if ((e = GemmDeepLaunch<Op, 64, 128, 4, EpiQKV<Op, true>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 128, 4, EpiQKV<Op, true>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 64, 128, 4, EpiBiasAct<Op, true>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 128, 4, EpiBiasAct<Op, true>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 64, 128, 4, EpiSeqRows<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 128, 4, EpiSeqRows<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 64, 128, 4, EpiHead<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmDeepLaunch<Op, 32, 128, 4, EpiHead<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 64, 128, EpiBiasAct<Op>>::prepare()) != hipSuccess) return e;
  if ((e = GemmLaunch<Op, 64, 128, EpiSeqRows<Op>>::prepare()) != hipSuccess) return e;
  if ((e = prepare_clip<Op, 13, 7>()) != hipSuccess) return e;
  if ((e = prepare_clip<Op, 11, 6>()) != hipSuccess) return e;
  if ((e = ClipLaunch<Op, 4, EpiBiasAct<Op>>::prepare()) != hipSuccess) return e;  // (tamf_bench_gemm: FFN1 without the row factors, T = 196)
  if ((e = ClipLaunch<Op, 4, EpiStoreF32>::prepare()) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)attn_kernel<Op, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               AttnCfg<Op, 64>::SMEM)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)attn_kernel<Op, 128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               AttnCfg<Op, 128>::SMEM)) != hipSuccess) return e;
  {
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 64, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 64, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 128, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 128, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)attn_res_kernel<Op, 128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  }
  return hipSuccess;
}

#ifndef TAMF_ATTN_KSPLIT  // (0: A/B builds without the key split of the last query tile)
#define TAMF_ATTN_KSPLIT 1
#endif
template <class Op>
static hipError_t launch_attn(const AttnArgs<Op>& aa, int B, int hd, hipStream_t st) {
  const int nqt = (aa.Sp + 15) / 16;
  const int nw_max = 16;  // one workgroup per (clip, head) up to 256 queries: K/V streamed once
  int chunks = (nqt + nw_max - 1) / nw_max;
  // fewer (clip, head) pairs than CUs (B = 32: 128): split the queries of a pair over workgroups until the chip is filled;
  // K/V are then streamed once per workgroup, which costs less than idle CUs (a query's result does not depend on the split)
  while (chunks * B * aa.H < wg_slots() / 2 && chunks * 2 <= nqt) chunks *= 2;
  if ((g_sel & 4) && chunks * 2 <= nqt) chunks *= 2;  // A/B: one more split (two workgroups per CU at B = 64)
  const int nw = (nqt + chunks - 1) / chunks;
  dim3 grid(chunks, B * aa.H);
  constexpr int smem64 = AttnCfg<Op, 64>::SMEM, smem128 = AttnCfg<Op, 128>::SMEM;
  // up to 224 keys: K resident in LDS, exact two-pass softmax, three barriers (tamf_attn.h AttnRes); the streaming kernel serves
  // longer sequences and the A/B switch (selection bit 512)
  {
    if (!(g_sel & 512)) {
#define TAMF_TRY_RES(HD_, NKB_)                                                                            \
  if (hd == HD_ && AttnRes<Op, HD_, NKB_>::fits(aa.S, aa.Sp) && nw <= AttnRes<Op, HD_, NKB_>::MAXW) {     \
    int lds = AttnRes<Op, HD_, NKB_>::smem(aa.S, aa.Sp);                                                   \
    AttnArgs<Op> a2 = aa;                                                                                  \
    int nwl = nw;                                                                                          \
    /* f32, a clip of 4 n + 1 query tiles (13 at T = 196): its last tile is key-split over four waves (tamf_attn.h) - decided by */ \
    /* the clip's length alone, so that a clip's result does not depend on the batch size or the query split.  f32 only: the    */ \
    /* 16-bit modes' waves overlap on a SIMD, the split buys them nothing at B = 64 and costs 1 - 3 % at B <= 32                 */ \
    /* (whether the split applies is decided on the WORST case of the query split - one workgroup, nqt - 1 tile-owning waves: the  */ \
    /*  partial area grows with them and the four extra waves then sit at nqt + 3 - so it cannot apply at one batch size and not  */ \
    /*  at another)                                                                                                     */ \
    if (TAMF_ATTN_KSPLIT && Op::PREC == 0 && nqt % 4 == 1 && nqt >= 5 && nqt + 3 <= AttnRes<Op, HD_, NKB_>::MAXW &&            \
        AttnRes<Op, HD_, NKB_>::ksplit_extra(aa.S, aa.Sp, nqt - 1) >= 0) {                                 \
      const int nwq = (nqt - 1 + chunks - 1) / chunks, nlast = nqt - 1 - (chunks - 1) * nwq > 0 ? nqt - 1 - (chunks - 1) * nwq : 0; \
      /* the four key-split waves go to the SIMDs with the fewest tiles in the last workgroup (wave w runs on SIMD w mod 4) */     \
      const int r = nlast % 4, nlight = r ? 4 - r : 4;                                                     \
      int hw = 0, top = nwq, used = 0;                                                                     \
      for (int j = 0; j < 4; ++j) {                                                                        \
        const int simd = (r ? r : 0) + j % nlight;                                                         \
        int idx = nlast + ((simd - nlast % 4) + 4) % 4;                                                    \
        while (used & (1 << idx)) idx += 4;                                                                \
        used |= 1 << idx; hw |= idx << (4 * j); if (idx + 1 > top) top = idx + 1;                          \
      }                                                                                                    \
      const int extra = AttnRes<Op, HD_, NKB_>::ksplit_extra(aa.S, aa.Sp, nwq);                            \
      if (extra >= 0 && top <= AttnRes<Op, HD_, NKB_>::MAXW) { a2.ksplit = 1; a2.nwq = nwq; a2.hw = hw; nwl = top; lds += extra; } \
    }                                                                                                      \
    hipLaunchKernelGGL((attn_res_kernel<Op, HD_, NKB_>), grid, dim3(nwl * 64), lds, st, a2);               \
    return hipGetLastError();                                                                              \
  }
      TAMF_TRY_RES(64, 4)   // up to 128 keys
      TAMF_TRY_RES(64, 6)   // up to 192 keys: the dataset's clips (T = 160: S = 165, 11 key tiles of the 12)
      TAMF_TRY_RES(64, 7)   // up to 224 keys (T = 196)
      TAMF_TRY_RES(128, 4)
      TAMF_TRY_RES(128, 6)
      TAMF_TRY_RES(128, 7)
#undef TAMF_TRY_RES
    }
  }
  if (hd == 64) {
    hipLaunchKernelGGL((attn_kernel<Op, 64>), grid, dim3(nw * 64), smem64, st, aa);
  } else if (hd == 128) {
    hipLaunchKernelGGL((attn_kernel<Op, 128>), grid, dim3(nw * 64), smem128, st, aa);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static inline dim3 grid1d(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// Keys per V^T row (the row stride of Vt_op): the sequence rounded up to whole 32-key blocks - and at least the rows of the clip tile
// that writes V^T in f32 (EpiVt stores every row tile of its 208- / 176-row tile, zeros past the clip: T = 174 is S = 179, 192 keys in
// blocks, but 13 row tiles = 208 stored positions - they ran into the next feature's row and, for the last one, past the buffer)
static inline int vt_row_keys(int S, int Sp) {
  const int tile = (Sp > 176 && Sp <= 208) ? 208 : (Sp > 144 && Sp <= 176) ? 176 : 0;
  return round_up(S > tile ? S : tile, 32);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" const char* tamf_last_error(const tamf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_noctx_err.c_str(); }

static int alloc_operand(tamf_ctx* ctx, OperandBuf* ob, long elems, bool zero = false) {
  return dev_alloc(ctx, &ob->p, (size_t)elems * ctx->EB, zero);
}

// the kernels address operands with 32-bit byte offsets and int element indices: refuse shapes beyond them
static int check_dims(const tamf_arch* arch, int max_batch, int max_frames) {
  if (max_batch <= 0 || max_frames <= 0 || max_frames > 4990) return fail(nullptr, TAMF_ERR_INVALID, "bad max_batch/max_frames");
  const long long sp = (max_frames + 5 + 7) / 8 * 8, mmax = (long long)max_batch * sp;
  const long long widest = std::max<long long>(arch->ff_size, 3LL * arch->latent_dim);
  if (mmax * widest * 4 >= (1LL << 32) || (long long)max_batch * max_frames * 896 * 4 >= (1LL << 32))
    return fail(nullptr, TAMF_ERR_INVALID, "max_batch x max_frames too large for one context (32-bit operand offsets): split the batch");
  return 0;
}

// every buffer whose size depends on (max_batch, max_frames): sampler state, token rows, operand planes, V^T, scratch
static int alloc_workspaces(tamf_ctx* ctx, int max_batch, int max_frames) {
  const int d = ctx->d;
  const int Smax = max_frames + ctx->P, Spmax = round_up(Smax, 8), Skpmax = round_up(Smax > 208 ? Smax : 208, 32);  // (>= vt_row_keys of every shape)
  const long BT = (long)max_batch * max_frames;
  const long Mmax = (long)max_batch * Spmax;
  ctx->Mmax = Mmax;
  ctx->Bmax = max_batch;
  ctx->Tmax = max_frames;
  ctx->alloc_ws = true;
  int rc = 0;
#define A(call)             \
  do {                      \
    g_alloc_tag = #call;    \
    if (rc == 0) rc = call; \
  } while (0)
  A(dev_alloc(ctx, (void**)&ctx->xs, BT * ctx->XK * 4, true));
  if (ctx->prec == TAMF_PREC_F32) ctx->xs_op.p = ctx->xs;
  else { A(alloc_operand(ctx, &ctx->xs_op, BT * ctx->XK, true)); ctx->xs_st = ctx->xs_op.p; }
  A(dev_alloc(ctx, (void**)&ctx->cobj, BT * d * 4));
  A(alloc_operand(ctx, &ctx->h1_op, BT * d));
  A(dev_alloc(ctx, (void**)&ctx->X, Mmax * d * 4, true));
  if (ctx->prec == TAMF_PREC_F32) ctx->X_op.p = ctx->X;
  else { A(alloc_operand(ctx, &ctx->X_op, Mmax * d, true)); ctx->X_st = ctx->X_op.p; }
  A(alloc_operand(ctx, &ctx->QK_op, Mmax * 2 * d, true));
  A(alloc_operand(ctx, &ctx->Vt_op, (long)max_batch * d * Skpmax, true));
  A(alloc_operand(ctx, &ctx->A_op, Mmax * d, true));
  A(alloc_operand(ctx, &ctx->H_op, Mmax * ctx->ff, true));
  A(dev_alloc(ctx, (void**)&ctx->pstatic, (long)max_batch * ctx->P * d * 4, true));
  A(dev_alloc(ctx, (void**)&ctx->tcur, (long)max_batch * 4, true));
  A(dev_alloc(ctx, (void**)&ctx->status, 16, true));
  A(dev_alloc(ctx, (void**)&ctx->side_dev, max_batch, true));
  A(dev_alloc(ctx, (void**)&ctx->objnum_dev, (long)max_batch * 4, true));
  A(dev_alloc(ctx, (void**)&ctx->stat_att, Mmax * (d / 32) * 8, true));  // block statistics of the residual stream (deferred LayerNorm)
  A(dev_alloc(ctx, (void**)&ctx->stat_ffn, Mmax * (d / 32) * 8, true));
  A(dev_alloc(ctx, (void**)&ctx->loop_params, sizeof(LoopParams), true));
#undef A
  g_alloc_tag = "(weights / tables)";
  ctx->alloc_ws = false;
  return rc;
}

extern "C" int tamf_ctx_create(const tamf_arch* arch, int32_t max_batch, int32_t max_frames, int32_t precision,
                               int32_t device, tamf_ctx** out) {
  if (!arch || !out) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  *out = nullptr;
  const int d = arch->latent_dim;
  if (!(d == 128 || d == 256 || d == 512)) return fail(nullptr, TAMF_ERR_INVALID, "latent_dim must be 128, 256 or 512");
  if (arch->num_heads <= 0 || d % arch->num_heads) return fail(nullptr, TAMF_ERR_INVALID, "bad num_heads");
  const int hd = d / arch->num_heads;
  if (!(hd == 64 || hd == 128)) return fail(nullptr, TAMF_ERR_INVALID, "head dim (latent_dim/num_heads) must be 64 or 128");
  if (arch->ff_size <= 0 || arch->ff_size % 128) return fail(nullptr, TAMF_ERR_INVALID, "ff_size must be a multiple of 128");
  if (arch->num_layers <= 0 || arch->num_layers > 64) return fail(nullptr, TAMF_ERR_INVALID, "bad num_layers");
  if (arch->input_dim <= 0 || arch->input_dim > 128) return fail(nullptr, TAMF_ERR_INVALID, "input_dim must be in [1,128]");
  if (precision < 0 || precision > TAMF_PREC_F16X3) return fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  if (arch->kind != TAMF_KIND_G && arch->kind != TAMF_KIND_R) return fail(nullptr, TAMF_ERR_INVALID, "unknown model kind");
  int ndev = 0;
  hipError_t de = hipGetDeviceCount(&ndev);
  if (de != hipSuccess || ndev <= 0)
    return fail(nullptr, TAMF_ERR_HIP, std::string("no HIP device visible (") + hipGetErrorString(de) +
                                           "): libtamf_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, TAMF_ERR_INVALID, "device index out of range");

  if (int rc = check_dims(arch, max_batch, max_frames)) return rc;
  tamf_ctx* ctx = new tamf_ctx();
  ctx->arch = *arch;
  ctx->prec = precision;
  ctx->device = device;
  ctx->Bmax = max_batch;
  ctx->Tmax = max_frames;
  ctx->d = d;
  ctx->ff = arch->ff_size;
  ctx->L = arch->num_layers;
  ctx->H = arch->num_heads;
  ctx->hd = hd;
  ctx->F = arch->input_dim;
  ctx->has_t = arch->kind == TAMF_KIND_G ? 1 : 0;
  ctx->P = arch->kind == TAMF_KIND_G ? 5 : 3;
  ctx->XK = arch->kind == TAMF_KIND_G ? 128 : round_up(arch->input_dim + arch->h2o_dim, 64);
  ctx->EB = precision == TAMF_PREC_BF16 ? 2 : 4;
  ctx->layers.resize(ctx->L);

  auto bail = [&](int rc) {
    std::string e = ctx->err;
    tamf_ctx_destroy(ctx);
    g_noctx_err = e;
    return rc;
  };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(ctx, TAMF_ERR_HIP, "hipSetDevice failed"));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      g_wg_slots_dev[device & 63].store(2 * cus, std::memory_order_relaxed);
  }
  hipError_t pe;
  {
    TAMF_STATE_LOCK;  // (one context at a time sets the kernel attributes of a device: they are idempotent, this only avoids the duplicate work)
    pe = prepare_all<OpF32>();
    if (pe == hipSuccess && precision != TAMF_PREC_F32) TAMF_WITH_OP(precision, pe = prepare_all<Op>());
  }
  if (pe != hipSuccess)
    return bail(fail(ctx, TAMF_ERR_HIP, std::string("kernel attribute setup failed: ") + hipGetErrorString(pe)));
  if (hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(ctx, TAMF_ERR_HIP, "hipStreamCreate failed"));

  if (int rc = alloc_workspaces(ctx, max_batch, max_frames)) return bail(rc);
  if (hipEventCreateWithFlags(&ctx->graph_done, hipEventDisableTiming) != hipSuccess)
    return bail(fail(ctx, TAMF_ERR_HIP, "hipEventCreate failed"));
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ctx.push_back(ctx);
  }
  *out = ctx;
  return 0;
}

extern "C" void tamf_ctx_destroy(tamf_ctx* ctx) {
  if (!ctx) return;
  TAMF_LAUNCH_LOCK;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), ctx), g_live_ctx.end());
  }
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->graph_done) (void)hipEventDestroy(ctx->graph_done);
  if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->graph) (void)hipGraphDestroy(ctx->graph);
  if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
#ifdef TAMF_OVERLAP_PROBE
  for (hipEvent_t e : ctx->pp_ev) (void)hipEventDestroy(e);
#endif
  for (void* p : ctx->allocs) (void)hipFree(p);
  for (void* p : ctx->ws_allocs) (void)hipFree(p);
  delete ctx;
}

static int retire_graph(tamf_ctx* ctx);

// the members alloc_workspaces() fills: saved / restored as a unit by tamf_ctx_resize
struct WorkspaceSet {
  float *xs, *cobj, *X, *pstatic;
  OperandBuf xs_op, h1_op, X_op, QK_op, Vt_op, A_op, H_op;
  void *X_st, *xs_st;
  int* tcur;
  unsigned* status;
  unsigned char* side_dev;
  int* objnum_dev;
  float2 *stat_att, *stat_ffn;
  LoopParams* loop_params;
  long Mmax;
  int Bmax, Tmax;
  std::vector<void*> ws_allocs;
  std::vector<GuardRec> ws_guards;
};
static WorkspaceSet ws_take(tamf_ctx* c) {  // moves the workspace set out of the context (its lists of workspace allocations / guards become empty)
  WorkspaceSet w{c->xs, c->cobj, c->X, c->pstatic, c->xs_op, c->h1_op, c->X_op, c->QK_op, c->Vt_op, c->A_op, c->H_op, c->X_st, c->xs_st, c->tcur,
                 c->status, c->side_dev, c->objnum_dev, c->stat_att, c->stat_ffn, c->loop_params, c->Mmax, c->Bmax, c->Tmax, {}, {}};
  w.ws_allocs.swap(c->ws_allocs);
  for (const GuardRec& g : c->guards)
    if (g.ws) w.ws_guards.push_back(g);
  c->guards.erase(std::remove_if(c->guards.begin(), c->guards.end(), [](const GuardRec& g) { return g.ws; }), c->guards.end());
  return w;
}
static void ws_put(tamf_ctx* c, WorkspaceSet& w) {
  c->xs = w.xs; c->cobj = w.cobj; c->X = w.X; c->pstatic = w.pstatic;
  c->xs_op = w.xs_op; c->h1_op = w.h1_op; c->X_op = w.X_op; c->QK_op = w.QK_op; c->Vt_op = w.Vt_op; c->A_op = w.A_op; c->H_op = w.H_op;
  c->X_st = w.X_st; c->xs_st = w.xs_st; c->tcur = w.tcur; c->status = w.status; c->side_dev = w.side_dev; c->objnum_dev = w.objnum_dev;
  c->stat_att = w.stat_att; c->stat_ffn = w.stat_ffn; c->loop_params = w.loop_params;
  c->Mmax = w.Mmax; c->Bmax = w.Bmax; c->Tmax = w.Tmax;
  c->ws_allocs.swap(w.ws_allocs);
  c->guards.insert(c->guards.end(), w.ws_guards.begin(), w.ws_guards.end());
}

// Transactional: the new workspaces are allocated FIRST, beside the old ones; only when every allocation has succeeded
// are the old ones freed.  On failure (the likely one: a larger batch that does not fit) the partial new set is freed, the old set -
// pointers, dimensions, conditioning - is put back and the context keeps working at its old size; the captured graph is gone either
// way (it holds the old pointers' kernels: harmless, but it is re-captured on the next loop).  The sticky status word travels with
// the context: bits raised and not yet read survive the resize.
extern "C" int tamf_ctx_resize(tamf_ctx* ctx, int32_t max_batch, int32_t max_frames) {
  TAMF_LAUNCH_LOCK;
  if (!ctx) return fail(ctx, TAMF_ERR_INVALID, "null ctx");
  if (int rc = check_dims(&ctx->arch, max_batch, max_frames)) {
    ctx->err = g_noctx_err;
    return rc;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  TRY(retire_graph(ctx));
  if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
  ctx->graph_key = GraphKey{};
  WorkspaceSet old = ws_take(ctx);
  int rc = alloc_workspaces(ctx, max_batch, max_frames);
  if (rc == 0 && old.status && ctx->status &&
      hipMemcpy(ctx->status, old.status, 16, hipMemcpyDeviceToDevice) != hipSuccess)
    rc = fail(ctx, TAMF_ERR_HIP, "tamf_ctx_resize: carrying the status word over failed");
  if (rc != 0) {
    const std::string why = ctx->err;
    WorkspaceSet part = ws_take(ctx);  // whatever alloc_workspaces got before it failed
    for (void* p : part.ws_allocs) (void)hipFree(p);
    ws_put(ctx, old);
    ctx->alloc_ws = false;
    ctx->err = "tamf_ctx_resize(" + std::to_string(max_batch) + ", " + std::to_string(max_frames) + ") failed, the context keeps its " +
               std::to_string(ctx->Bmax) + " x " + std::to_string(ctx->Tmax) + " workspaces: " + why;
    return rc;
  }
  for (void* p : old.ws_allocs) (void)hipFree(p);
  ctx->cond_set = false;
  ctx->B = ctx->T = ctx->S = ctx->Sp = ctx->Skp = ctx->M = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static std::map<std::string, std::vector<int64_t>> expected_shapes(const tamf_ctx* c) {
  std::map<std::string, std::vector<int64_t>> m;
  const int64_t d = c->d, ff = c->ff;
  auto lin = [&](const std::string& n, int64_t o, int64_t i) {
    m[n + ".weight"] = {o, i};
    m[n + ".bias"] = {o};
  };
  m["hand_side_process.rh_embed"] = {d};
  m["hand_side_process.lh_embed"] = {d};
  lin("hand_shape_process.shape_embed", d, c->arch.hand_shape_dim);
  lin("obj_embed_process.embedding", d, c->arch.obj_embed_dim);
  lin("input_process.poseEmbedding", d, c->arch.input_dim);
  lin("obj_input_process.poseEmbedding", d, c->arch.obj_input_dim);
  if (c->arch.kind == TAMF_KIND_R) {
    lin("h2o_dist_input_process.poseEmbedding", d, c->arch.h2o_dim);
    lin("input_merge.0", d, 3 * d);
  } else {
    lin("input_merge.0", d, 2 * d);
    lin("embed_timestep.time_embed.0", d, d);
    lin("embed_timestep.time_embed.2", d, d);
    lin("embed_text", d, c->arch.clip_dim);
  }
  lin("input_merge.2", d, d);
  m["sequence_pos_encoder.pe"] = {5000, 1, d};
  for (int l = 0; l < c->L; ++l) {
    const std::string p = "seqTransEncoder.layers." + std::to_string(l);
    m[p + ".self_attn.in_proj_weight"] = {3 * d, d};
    m[p + ".self_attn.in_proj_bias"] = {3 * d};
    lin(p + ".self_attn.out_proj", d, d);
    lin(p + ".linear1", ff, d);
    lin(p + ".linear2", d, ff);
    m[p + ".norm1.weight"] = {d};
    m[p + ".norm1.bias"] = {d};
    m[p + ".norm2.weight"] = {d};
    m[p + ".norm2.bias"] = {d};
  }
  lin("output_process.poseFinal", c->arch.input_dim, d);
  return m;
}

extern "C" int tamf_load_weight(tamf_ctx* ctx, const char* name, const float* host_data, const int64_t* shape,
                                int32_t ndim) {
  if (!ctx || !name || !host_data || !shape || ndim <= 0) return fail(ctx, TAMF_ERR_INVALID, "null/invalid argument");
  if (ctx->finalized) return fail(ctx, TAMF_ERR_STATE, "weights already finalised");
  const auto exp = expected_shapes(ctx);
  const auto it = exp.find(name);
  if (it == exp.end()) return 0;  // strict=False: unexpected keys are ignored (e.g. clip_model.*, aliased pe buffer)
  std::vector<int64_t> shp(shape, shape + ndim);
  if (shp != it->second) return fail(ctx, TAMF_ERR_INVALID, std::string("shape mismatch for ") + name);
  int64_t n = 1;
  for (auto v : shp) n *= v;
  ctx->raw[name].assign(host_data, host_data + n);
  ctx->raw_shape[name] = shp;
  return 0;
}

template <class Op>
static OperandBuf ob_cast(const OperandBuf& o) { return o; }

static int upload_f32(tamf_ctx* ctx, const std::string& name, float** p) {
  const auto& v = ctx->raw.at(name);
  return dev_upload(ctx, p, v.data(), v.size());
}

extern "C" int tamf_finalize_weights(tamf_ctx* ctx, int32_t max_timesteps, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx) return fail(ctx, TAMF_ERR_INVALID, "null ctx");
  if (ctx->finalized) return fail(ctx, TAMF_ERR_STATE, "weights already finalised");
  if (max_timesteps <= 0 || max_timesteps > 5000) return fail(ctx, TAMF_ERR_INVALID, "max_timesteps must be in [1,5000]");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  for (const auto& kv : expected_shapes(ctx))
    if (!ctx->raw.count(kv.first)) return fail(ctx, TAMF_ERR_MISSING, "missing checkpoint tensor: " + kv.first);
  const int d = ctx->d, ff = ctx->ff, F = ctx->F, prec = ctx->prec;
  auto R = [&](const std::string& n) -> const float* { return ctx->raw.at(n).data(); };

  for (int l = 0; l < ctx->L; ++l) {
    const std::string p = "seqTransEncoder.layers." + std::to_string(l);
    LayerW& w = ctx->layers[l];
    {
      // Deferred LayerNorm: fold the gain AND the centring of the LayerNorm in front of a GEMM into its weight, W'' = W diag(gamma) (I - 1 1^T / d)
      // (fp64: every row of W diag(gamma) minus its own mean), c2 = W beta + bias; the residual adds take gamma and beta + bias of the
      // LayerNorm their residual input passes through.  Layer 0's attention block reads the encoder input itself: the plain weight.
      const std::string pp = "seqTransEncoder.layers." + std::to_string(l - 1);
      const float* g_in = l ? R(pp + ".norm2.weight") : nullptr;   // LayerNorm in front of this layer's attention block
      const float* be_in = l ? R(pp + ".norm2.bias") : nullptr;
      const float* g1 = R(p + ".norm1.weight");                    // ... in front of its feed-forward block
      const float* be1 = R(p + ".norm1.bias");
      auto fold = [&](const float* W, const float* bias, int N, const float* g, const float* be, OperandBuf* ob, float** c2,
                      const std::string& what) -> int {
        std::vector<float> wf((size_t)N * d), v2(N);
        for (int n = 0; n < N; ++n) {
          double acc = bias[n], rowsum = 0.0;
          for (int k = 0; k < d; ++k) {
            if (g) rowsum += (double)W[(size_t)n * d + k] * g[k];
            if (be) acc += (double)W[(size_t)n * d + k] * be[k];
          }
          const double shift = g ? rowsum / d : 0.0;  // W'' = W diag(gamma) (I - 1 1^T / d): every row minus its own mean
          for (int k = 0; k < d; ++k) wf[(size_t)n * d + k] = g ? (float)((double)W[(size_t)n * d + k] * g[k] - shift) : W[(size_t)n * d + k];
          v2[n] = (float)acc;
        }
        TRY(upload_operand(ctx, prec, wf.data(), N, d, d, ob, what.c_str()));
        return dev_upload(ctx, c2, v2.data(), v2.size());
      };
      TRY(fold(R(p + ".self_attn.in_proj_weight"), R(p + ".self_attn.in_proj_bias"), 3 * d, g_in, be_in, &w.Win, &w.c2_in,
               p + ".self_attn.in_proj_weight"));
      TRY(fold(R(p + ".linear1.weight"), R(p + ".linear1.bias"), ff, g1, be1, &w.W1, &w.c2_ff, p + ".linear1.weight"));
      std::vector<float> ga(d), bb(d);
      for (int n = 0; n < d; ++n) {
        ga[n] = g_in ? g_in[n] : 1.0f;
        bb[n] = (be_in ? be_in[n] : 0.0f) + R(p + ".self_attn.out_proj.bias")[n];
      }
      TRY(dev_upload(ctx, &w.g_att, ga.data(), ga.size()));
      TRY(dev_upload(ctx, &w.bb_att, bb.data(), bb.size()));
      for (int n = 0; n < d; ++n) {
        ga[n] = g1[n];
        bb[n] = be1[n] + R(p + ".linear2.bias")[n];
      }
      TRY(dev_upload(ctx, &w.g_ffn, ga.data(), ga.size()));
      TRY(dev_upload(ctx, &w.bb_ffn, bb.data(), bb.size()));
    }
    TRY(upload_operand(ctx, prec, R(p + ".self_attn.out_proj.weight"), d, d, d, &w.Wout, (p + ".self_attn.out_proj.weight").c_str()));
    TRY(upload_operand(ctx, prec, R(p + ".linear2.weight"), d, ff, ff, &w.W2, (p + ".linear2.weight").c_str()));
  }
  // fused input weight: input_merge.0[:, :d] . poseEmbedding (and, for R, input_merge.0[:, 2d:] . h2o embedding)
  {
    const bool isR = ctx->arch.kind == TAMF_KIND_R;
    const int mk = isR ? 3 * d : 2 * d;
    const float* Wm1 = R("input_merge.0.weight");
    const float* bm1 = R("input_merge.0.bias");
    const float* Wp = R("input_process.poseEmbedding.weight");
    const float* bp = R("input_process.poseEmbedding.bias");
    const int qd = ctx->arch.obj_input_dim;
    const float* Wq = R("obj_input_process.poseEmbedding.weight");
    const float* bq = R("obj_input_process.poseEmbedding.bias");
    std::vector<float> fused((size_t)d * ctx->XK, 0.f), cb(d), wct((size_t)qd * d);
    for (int n = 0; n < d; ++n) {
      double acc_b = bm1[n];
      for (int j = 0; j < d; ++j) acc_b += (double)Wm1[(size_t)n * mk + j] * bp[j];
      for (int k = 0; k < F; ++k) {
        double a = 0;
        for (int j = 0; j < d; ++j) a += (double)Wm1[(size_t)n * mk + j] * Wp[(size_t)j * F + k];
        fused[(size_t)n * ctx->XK + k] = (float)a;
      }
      // the object half, hoisted out of the loop (cobj_kernel): W_m1[:, d:2d] . (W_q m + b_q) = Wc m + ..., Wc = W_m1[:, d:2d] . W_q
      for (int k = 0; k < qd; ++k) {
        double a = 0;
        for (int j = 0; j < d; ++j) a += (double)Wm1[(size_t)n * mk + d + j] * Wq[(size_t)j * qd + k];
        wct[(size_t)k * d + n] = (float)a;
      }
      for (int j = 0; j < d; ++j) acc_b += (double)Wm1[(size_t)n * mk + d + j] * bq[j];
      if (isR) {
        const int Hd = ctx->arch.h2o_dim;
        const float* Wh = R("h2o_dist_input_process.poseEmbedding.weight");
        const float* bh = R("h2o_dist_input_process.poseEmbedding.bias");
        for (int j = 0; j < d; ++j) acc_b += (double)Wm1[(size_t)n * mk + 2 * d + j] * bh[j];
        for (int k = 0; k < Hd; ++k) {
          double a = 0;
          for (int j = 0; j < d; ++j) a += (double)Wm1[(size_t)n * mk + 2 * d + j] * Wh[(size_t)j * Hd + k];
          fused[(size_t)n * ctx->XK + F + k] = (float)a;
        }
      }
      cb[n] = (float)acc_b;
    }
    TRY(upload_operand(ctx, prec, fused.data(), d, ctx->XK, ctx->XK, &ctx->Wfused, "input_merge.0.weight x poseEmbedding.weight (fused)"));
    TRY(dev_upload(ctx, &ctx->WcT, wct.data(), wct.size()));
    TRY(dev_upload(ctx, &ctx->bc, cb.data(), cb.size()));
  }
  TRY(upload_operand(ctx, prec, R("input_merge.2.weight"), d, d, d, &ctx->Wm2, "input_merge.2.weight"));
  TRY(upload_f32(ctx, "input_merge.2.bias", &ctx->bm2));
  {
    std::vector<float> wf((size_t)ctx->XN * d, 0.f), bfp(ctx->XN, 0.f);
    memcpy(wf.data(), R("output_process.poseFinal.weight"), (size_t)F * d * 4);
    memcpy(bfp.data(), R("output_process.poseFinal.bias"), (size_t)F * 4);
    {  // the encoder's last LayerNorm, deferred into the head: W_f diag(gamma) (I - 1 1^T / d), c2 = W_f beta + b_f
      const std::string pl = "seqTransEncoder.layers." + std::to_string(ctx->L - 1);
      const float *g = R(pl + ".norm2.weight"), *be = R(pl + ".norm2.bias");
      for (int n = 0; n < F; ++n) {
        double acc = bfp[n], rowsum = 0.0;
        for (int k = 0; k < d; ++k) {
          acc += (double)wf[(size_t)n * d + k] * be[k];
          rowsum += (double)wf[(size_t)n * d + k] * g[k];
        }
        for (int k = 0; k < d; ++k) wf[(size_t)n * d + k] = (float)((double)wf[(size_t)n * d + k] * g[k] - rowsum / d);
        bfp[n] = (float)acc;
      }
    }
    TRY(upload_operand(ctx, prec, wf.data(), ctx->XN, d, d, &ctx->Wf, "output_process.poseFinal.weight"));
    TRY(dev_upload(ctx, &ctx->bf, bfp.data(), bfp.size()));
  }
  TRY(upload_f32(ctx, "sequence_pos_encoder.pe", &ctx->pe));
  auto upload_T = [&](const std::string& name, int N, int K, float** p) -> int {  // W [N][K] -> W^T [K][N] (prefix_rows_kernel)
    const float* W = R(name);
    std::vector<float> t((size_t)N * K);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = W[(size_t)n * K + k];
    return dev_upload(ctx, p, t.data(), t.size());
  };
  if (ctx->arch.hand_shape_dim > 64 || ctx->arch.obj_input_dim > COBJ_QMAX || d > 1024)
    return fail(ctx, TAMF_ERR_INVALID, "hand_shape_dim > 64, obj_input_dim > 16 or latent_dim > 1024");
  TRY(upload_T("hand_shape_process.shape_embed.weight", d, ctx->arch.hand_shape_dim, &ctx->WshapeT));
  TRY(upload_f32(ctx, "hand_shape_process.shape_embed.bias", &ctx->bshape));
  TRY(upload_T("obj_embed_process.embedding.weight", d, ctx->arch.obj_embed_dim, &ctx->WobjT));
  TRY(upload_f32(ctx, "obj_embed_process.embedding.bias", &ctx->bobj));
  TRY(upload_f32(ctx, "hand_side_process.rh_embed", &ctx->rh));
  TRY(upload_f32(ctx, "hand_side_process.lh_embed", &ctx->lh));
  if (ctx->has_t) {
    TRY(upload_T("embed_text.weight", d, ctx->arch.clip_dim, &ctx->WtxtT));
    TRY(upload_f32(ctx, "embed_text.bias", &ctx->btxt));
    TRY(upload_operand(ctx, TAMF_PREC_F32, R("embed_timestep.time_embed.0.weight"), d, d, d, &ctx->Wt1_f32, "embed_timestep.time_embed.0.weight"));
    TRY(upload_operand(ctx, TAMF_PREC_F32, R("embed_timestep.time_embed.2.weight"), d, d, d, &ctx->Wt2_f32, "embed_timestep.time_embed.2.weight"));
    TRY(upload_f32(ctx, "embed_timestep.time_embed.0.bias", &ctx->bt1));
    TRY(upload_f32(ctx, "embed_timestep.time_embed.2.bias", &ctx->bt2));
    // timestep-embedding table: temb[t] = nan_to_num(W2 silu(W1 pe[t] + b1) + b2) + pe[0]   (exact fp32 MFMA)
    ctx->n_t = max_timesteps;
    float* tmp = nullptr;
    TRY(dev_alloc(ctx, (void**)&ctx->temb, (size_t)ctx->n_t * d * 4));
    TRY(dev_alloc(ctx, (void**)&tmp, (size_t)ctx->n_t * d * 4));
    GemmArgs<OpF32> g1{ctx->pe, d, (const float*)ctx->Wt1_f32.p, d, ctx->n_t, d, d, 0};
    EpiBiasAct<OpF32> e1{ctx->bt1, nullptr, 0, tmp, d, ACT_SILU};
    HIPCHK(ctx, gemm128<OpF32>(g1, e1, st));
    GemmArgs<OpF32> g2{tmp, d, (const float*)ctx->Wt2_f32.p, d, ctx->n_t, d, d, 0};
    EpiSeqRows<OpF32> e2{ctx->bt2, ctx->pe, 0, ctx->temb, nullptr, d, 0x7FFFFFFF, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0};
    HIPCHK(ctx, gemm128<OpF32>(g2, e2, st));
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->raw.clear();
  ctx->finalized = true;
  if (!ctx->weight_note.empty()) ctx->err = ctx->weight_note;  // (readable through tamf_last_error after the successful call)
  return 0;
}

// destroy the executable graph once its last replay has finished (its launches may still be queued on the caller's stream)
static int retire_graph(tamf_ctx* ctx) {
  if (ctx->graph_in_flight) {
    HIPCHK(ctx, hipEventSynchronize(ctx->graph_done));
    ctx->graph_in_flight = false;
  }
  if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
  if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
  ctx->graph_key = GraphKey();
  return 0;
}

extern "C" int tamf_set_schedule(tamf_ctx* ctx, int32_t n_steps, const double* c1, const double* c2, const double* logvar) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !c1 || !c2 || !logvar || n_steps <= 0) return fail(ctx, TAMF_ERR_INVALID, "null/invalid argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->h_c1.resize(n_steps);
  ctx->h_c2.resize(n_steps);
  ctx->h_sigma.resize(n_steps);
  for (int i = 0; i < n_steps; ++i) {
    ctx->h_c1[i] = (float)c1[i];  // float64 -> float32 cast of _extract_into_tensor (gaussian_diffusion.py:1275)
    ctx->h_c2[i] = (float)c2[i];
    ctx->h_sigma[i] = expf(0.5f * (float)logvar[i]);  // th.exp(0.5 * log_variance) on float32 (:459)
  }
  ctx->n_steps = n_steps;
  if (ctx->tmap_on) {  // a new schedule is a new sampler: its timestep map (if any) is set after it
    TRY(retire_graph(ctx));
    ctx->tmap_on = false;
    ctx->tmap_max = -1;
  }
  if (n_steps > ctx->sched_cap) {  // (re)allocated only when a longer schedule arrives; a captured graph holds these pointers
    TRY(retire_graph(ctx));
    const int cap = std::max(n_steps, 1000);
    TRY(dev_alloc(ctx, (void**)&ctx->c1, (size_t)cap * 4));
    TRY(dev_alloc(ctx, (void**)&ctx->c2, (size_t)cap * 4));
    TRY(dev_alloc(ctx, (void**)&ctx->sigma, (size_t)cap * 4));
    ctx->sched_cap = cap;
  }
  // plain (synchronous) copies: ordered after earlier work on the null stream semantics of hipMemcpy
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipMemcpy(ctx->c1, ctx->h_c1.data(), (size_t)n_steps * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->c2, ctx->h_c2.data(), (size_t)n_steps * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->sigma, ctx->h_sigma.data(), (size_t)n_steps * 4, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int tamf_set_timestep_map(tamf_ctx* ctx, int32_t n_steps, const int32_t* map_host) {
  TAMF_LAUNCH_LOCK;
  if (!ctx) return fail(ctx, TAMF_ERR_INVALID, "null ctx");
  if (!ctx->finalized || !ctx->has_t) return fail(ctx, TAMF_ERR_STATE, "tamf_set_timestep_map needs a G context with finalised weights");
  if (ctx->n_steps <= 0 || n_steps != ctx->n_steps) return fail(ctx, TAMF_ERR_STATE, "tamf_set_timestep_map: set the schedule of the same length first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  TRY(retire_graph(ctx));  // (a captured loop holds the table pointer of its capture time)
  if (!map_host) {
    ctx->tmap_on = false;
    ctx->tmap_max = -1;
    return 0;
  }
  int mx = -1;
  for (int i = 0; i < n_steps; ++i) {
    if (map_host[i] < 0 || map_host[i] >= ctx->n_t)
      return fail(ctx, TAMF_ERR_INVALID, "timestep map entry " + std::to_string(map_host[i]) + " outside the timestep table (max_timesteps = " + std::to_string(ctx->n_t) + ")");
    if (i && map_host[i] <= map_host[i - 1]) return fail(ctx, TAMF_ERR_INVALID, "timestep map must be strictly increasing (respace.py:76-82)");
    mx = std::max(mx, (int)map_host[i]);
  }
  if (n_steps > ctx->temb_loop_cap) {
    const int cap = std::max(n_steps, 1000);
    TRY(dev_alloc(ctx, (void**)&ctx->temb_loop, (size_t)cap * ctx->d * 4));
    ctx->temb_loop_cap = cap;
  }
  int* map_dev = nullptr;
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipMalloc((void**)&map_dev, (size_t)n_steps * 4));
  hipError_t e = hipMemcpy(map_dev, map_host, (size_t)n_steps * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gather_rows_kernel, grid1d((long)n_steps * ctx->d), dim3(256), 0, nullptr, ctx->temb_loop, ctx->temb, map_dev, n_steps, ctx->d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(map_dev);
  HIPCHK(ctx, e);
  ctx->tmap_on = true;
  ctx->tmap_max = mx;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conditioning (step-invariant precompute, always exact fp32)
// ------------------------------------------------------------------------------------------------
extern "C" int tamf_set_cond(tamf_ctx* ctx, int32_t B, int32_t T, int32_t nobj, const float* text_emb_dev,
                             const uint8_t* hand_side_host, const float* shape_dev, const float* obj_emb_dev,
                             const float* obj_traj_dev, void* stream) {
  return tamf_set_cond_ragged(ctx, B, T, nobj, nullptr, text_emb_dev, hand_side_host, shape_dev, obj_emb_dev, obj_traj_dev, stream);
}

extern "C" int tamf_set_cond_ragged(tamf_ctx* ctx, int32_t B, int32_t T, int32_t nobj, const int32_t* obj_num_host,
                                    const float* text_emb_dev, const uint8_t* hand_side_host, const float* shape_dev,
                                    const float* obj_emb_dev, const float* obj_traj_dev, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx) return fail(ctx, TAMF_ERR_INVALID, "null ctx");
  if (!ctx->finalized) return fail(ctx, TAMF_ERR_STATE, "weights not finalised");
  if (B <= 0 || B > ctx->Bmax || T <= 0 || T > ctx->Tmax || nobj <= 0) return fail(ctx, TAMF_ERR_INVALID, "B/T/nobj out of range");
  if (!hand_side_host || !shape_dev || !obj_emb_dev || !obj_traj_dev || (ctx->has_t && !text_emb_dev))
    return fail(ctx, TAMF_ERR_INVALID, "null conditioning tensor");
  for (int b = 0; b < B; ++b)
    if (hand_side_host[b] > 1) return fail(ctx, TAMF_ERR_INVALID, "unexpected hand_side: " + std::to_string(hand_side_host[b]));
  if (obj_num_host)
    for (int b = 0; b < B; ++b)
      if (obj_num_host[b] < 1 || obj_num_host[b] > nobj)
        return fail(ctx, TAMF_ERR_INVALID, "obj_num[" + std::to_string(b) + "] = " + std::to_string(obj_num_host[b]) + " outside [1, nobj]");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const int* cnt = nullptr;
  if (obj_num_host) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->objnum_dev, obj_num_host, sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
    cnt = ctx->objnum_dev;
  }
  const int d = ctx->d, P = ctx->P, ht = ctx->has_t, nrows = P - ht;
  ctx->B = B;
  ctx->T = T;
  ctx->S = T + P;
  ctx->Sp = round_up(ctx->S, 8);
  ctx->Skp = vt_row_keys(ctx->S, ctx->Sp);
  ctx->M = B * ctx->Sp;
  ctx->cond_set = false;
  HIPCHK(ctx, hipMemcpyAsync(ctx->side_dev, hand_side_host, B, hipMemcpyHostToDevice, st));
  // two launches (tamf_misc.h): the static prefix rows of every clip, and the hoisted object half of input_merge.0 per frame
  const int sd = ctx->arch.hand_shape_dim, od = ctx->arch.obj_embed_dim, qd = ctx->arch.obj_input_dim, cd = ht ? ctx->arch.clip_dim : 0;
  PrefixArgs pa{};
  pa.text = ht ? text_emb_dev : nullptr; pa.WtxtT = ctx->WtxtT; pa.btxt = ctx->btxt;
  pa.side = ctx->side_dev; pa.rh = ctx->rh; pa.lh = ctx->lh;
  pa.shape = shape_dev; pa.WshapeT = ctx->WshapeT; pa.bshape = ctx->bshape;
  pa.oemb = obj_emb_dev; pa.WobjT = ctx->WobjT; pa.bobj = ctx->bobj; pa.cnt = cnt;
  pa.pe = ctx->pe; pa.pstatic = ctx->pstatic;
  pa.B = B; pa.d = d; pa.T = T; pa.sd = sd; pa.nobj = nobj; pa.od = od; pa.clip_dim = cd; pa.nrows = nrows; pa.ht = ht;
  const size_t psm = (size_t)(std::max(std::max(cd, od), sd) + 1024) * sizeof(float);
  hipLaunchKernelGGL(prefix_rows_kernel, dim3(B, nrows), dim3(1024), psm, st, pa);
  hipLaunchKernelGGL(cobj_kernel, dim3((unsigned)(((long)B * T + COBJ_ROWS - 1) / COBJ_ROWS)), dim3(256), 0, st, obj_traj_dev, ctx->WcT, ctx->bc,
                     ctx->cobj, B, nobj, T, qd, d, cnt);
  HIPCHK(ctx, hipGetLastError());
  ctx->cond_set = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// one denoiser evaluation = the kernel sequence below (captured into a hipGraph by the sampling loop)
// ------------------------------------------------------------------------------------------------
#ifdef TAMF_OVERLAP_PROBE
// OVERLAP PROBE (-DTAMF_OVERLAP_PROBE measurement builds only; tamf_set_gemm_tuning selection bit 256; plain launches, no hipGraph).  What would a step gain
// if launch k + 1 could start on the CUs that launch k has left, instead of behind the kernel boundary - the most that per-clip ready
// flags / a persistent per-clip pipeline could recover?  The launches of the loop alternate between the caller's stream and the
// context's second stream, and NOTHING enforces their data dependencies: the SAMPLES ARE GARBAGE, only the time means something, and
// it is an upper bound (a real consumer waits for its producers' tiles; here it waits for nothing).  Order of dispatch is kept sane:
// launch k becomes eligible when launch k - 2 (same stream) and launch k - 3 are complete - i.e. never before launch k - 1 became
// eligible - so at most two launches share the chip, the older one placed first (gfx950 / ROCm 7.2 ignores hipExtAnyOrderLaunch on
// one stream: tools/micro/anyorder.hip, profiles/r06/anyorder_c01.txt).
struct PingPong {
  tamf_ctx* c;
  hipStream_t a, b;
  hipStream_t next() {
    if (!c->pp_on) return a;
    const long k = c->pp_count++;
    hipStream_t s = (k & 1) ? b : a;
    const size_t NEV = c->pp_ev.size() - 1;  // (the last event is the loop's fork / join)
    (void)hipEventRecord(c->pp_ev[k % NEV], s);  // fires when everything launched on s before launch k is complete
    if (k > 0) (void)hipStreamWaitEvent(s, c->pp_ev[(k - 1) % NEV], 0);  // ... and launch k waits for what launch k - 1 waited for
    return s;
  }
};
#define TAMF_NEXT_STREAM pp_.next()
#else
#define TAMF_NEXT_STREAM st
#endif
// residual GEMM of the deferred-LayerNorm form (out-proj, FFN2; 16-bit modes): the clip's row parts where whole clips would fill at
// most half of the CUs, whole-clip tiles where they fill at least half of their rounds' slots, 128 x 128 tiles for every other shape -
// the same K order per element and the same statistics trees in all three, i.e. the same bits
template <class Op>
static hipError_t launch_resid(const GemmArgs<Op>& ga, const EpiResid<Op>& ep, int B, int Sp, hipStream_t st) {
  // a few clips per call: whole-clip or row-part tiles would put 2 N / 128 workgroups on a clip - FFN2 of ONE clip ran 34 us (f32: 100 us)
  // on 8 CUs, 40 - 55 % of the step (profiles/r05/small_batch_resid_c25.txt, ..._c27.txt)
  // one to ~ 6 clips: 32 x 64 tiles (4 waves, six stages) while they fit one workgroup per CU - half the L2 -> LDS fill per workgroup and K
  // interval of the 32 x 128 tile, which is what a one-clip FFN2 was left bound by: 19.1 -> 14.2 us (f16x3), 10.3 -> 7.3 (bf16), 46.9 -> 25.5
  // (f32); the step of one clip 434 -> 385 / 256 -> 226 / 900 -> 686 us (profiles/r05/small_batch_32x64_c37.txt; same bits)
  if (!(g_sel & (1 | 16)) && ga.N % 64 == 0 && ((ga.M + 31) / 32) * (ga.N / 64) <= wg_slots() / 2)
    return GemmDeepLaunch<Op, 32, 64, 6, EpiResid<Op>, 2, 2>::launch(ga, ep, st);
  {
    hipError_t e = hipSuccess;
    if (small_m_launch<Op>(ga, ep, st, &e)) return e;
  }
  // Between "a few clips" and the whole-clip tiles (20 - 39 clips): 64 x 128 tiles with THREE stages - 72 KB, two workgroups per CU - while
  // they fit the workgroup slots.  bf16, whose K tiles are the shortest: 0.815 against 0.850 ms per step at 32 clips on the row-part tiles
  // (-4 %; profiles/r05/mid_batch_resid_c33.txt, variants alternating); the split modes gain up to ~ 1.25 tiles per CU only (24 clips:
  // -3 %; 32 clips: 1.510 against 1.510), f32 loses (MFMA-bound: the row parts multiply with fewer staged bytes per flop).  At 64 clips
  // the whole-clip tiles win by a wide margin (b64_resid_deep_c34.txt).  Same K order per element: the same bits.
  if (!(g_sel & (1 | 16)) && Op::PREC != 0 && ga.N % 128 == 0) {
    const int t64 = ((ga.M + 63) / 64) * (ga.N / 128);
    if (Op::SPLIT ? t64 * 4 <= wg_slots() / 2 * 5 : t64 <= wg_slots()) return GemmDeepLaunch<Op, 64, 128, 3, EpiResid<Op>>::launch(ga, ep, st);
  }
  if (!(g_sel & 2)) {
    TAMF_CLIP_NSUB(Sp, {
      if (ClipLaunch<Op, 2, EpiResid<Op>, NSP, 2>::applies_parts(B, Sp, ga.N, ga.K))
        return ClipLaunch<Op, 2, EpiResid<Op>, NSP, 2>::launch(nullptr, ga.A, ga.lda, ga.W, ga.ldw, B, Sp, ga.N, ga.K, ep, st);
      if (ClipLaunch<Op, 2, EpiResid<Op>, NS>::applies(B, Sp, ga.N, ga.K, 50))
        return ClipLaunch<Op, 2, EpiResid<Op>, NS>::launch(nullptr, ga.A, ga.lda, ga.W, ga.ldw, B, Sp, ga.N, ga.K, ep, st);
    })
  }
  return gemm128<Op>(ga, ep, st);
}

template <class Op>
static int enqueue_step(tamf_ctx* ctx, hipStream_t st, const EpiHead<Op>& head_in, int t_off = 0) {
  typedef typename Op::elem_t E;
  const int d = ctx->d, ff = ctx->ff, B = ctx->B, T = ctx->T, S = ctx->S, Sp = ctx->Sp, M = ctx->M, P = ctx->P;
  int nk = 0;
#ifdef TAMF_OVERLAP_PROBE
  PingPong pp_{ctx, st, ctx->cap_stream};
#endif
  // algorithmic FLOPs of the reference work each launch stands for (true S, not padded rows)
  const double BS = (double)B * S, BT = (double)B * T, dd = d, F = ctx->F;
  auto mark = [&](const char* name, double flops) {
    ++nk;
    if (!ctx->prof_on) return;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, st);
    ctx->prof_ev.push_back(ev);
    ctx->prof_names.push_back(name);
    ctx->prof_flops.push_back(flops);
  };
  {  // input_merge.0 on [pose | (h2o)] with the hoisted object term, SiLU
    hipStream_t sk = TAMF_NEXT_STREAM;
    GemmArgs<Op> ga{(const E*)ctx->xs_op.p, ctx->XK, (const E*)ctx->Wfused.p, ctx->XK, B * T, d, ctx->XK, 0};
    EpiBiasAct<Op> ep{nullptr, ctx->cobj, d, (E*)ctx->h1_op.p, d, ACT_SILU, {ctx->Wfused.inv_scale, ctx->status}};
    HIPCHK(ctx, gemm128<Op>(ga, ep, sk));
    mark("gemm_input_merge0", BT * (2.0 * F * dd + 2.0 * dd * (ctx->arch.kind == TAMF_KIND_R ? 3 : 2) * dd +
                                    (ctx->arch.kind == TAMF_KIND_R ? 2.0 * ctx->arch.h2o_dim * dd : 0.0)));
  }
  {  // input_merge.2 + nan_to_num + positional rows -> token rows of X
    hipStream_t sk = TAMF_NEXT_STREAM;
    GemmArgs<Op> ga{(const E*)ctx->h1_op.p, d, (const E*)ctx->Wm2.p, d, B * T, d, d, 0};
    // (+ the prefix and pad rows of every clip, written by the tile that holds the clip's first frame)
    const float* temb_now = (ctx->in_loop && ctx->tmap_on) ? ctx->temb_loop : ctx->temb;
    EpiSeqRows<Op> ep{ctx->bm2, ctx->pe + (long)P * d, d, ctx->X, (E*)ctx->X_st, d, T, Sp, P, ctx->pstatic, temb_now, ctx->tcur, ctx->has_t, S, t_off, {ctx->Wm2.inv_scale, ctx->status}};
    hipError_t es = hipSuccess;
    if (small_m_launch<Op>(ga, ep, sk, &es)) HIPCHK(ctx, es);
    else HIPCHK(ctx, gemm128<Op>(ga, ep, sk));
    mark("gemm_input_merge2", BT * 2.0 * dd * dd + (ctx->has_t ? B * 4.0 * dd * dd : 0.0));
  }
  const float qscale = 1.4426950408889634f / sqrtf((float)ctx->hd);
  {
    // Deferred LayerNorm (tamf_device.h): the residual stream X / X_op holds the UN-normalised sums; five launches per layer (f32: six,
    // the QKV projection as two clip launches)
    {
      const int NB = d / 32;
      const float inv_d = 1.0f / (float)d;
      for (int l = 0; l < ctx->L; ++l) {
        LayerW& w = ctx->layers[l];
        const LnStats ln_in{l ? ctx->stat_ffn : nullptr, NB, inv_d, 1e-5f};  // the LayerNorm in front of the attention block (layer 0: none)
        const LnStats ln_ff{ctx->stat_att, NB, inv_d, 1e-5f};               // ... in front of the feed-forward block
        {
          hipStream_t sk = TAMF_NEXT_STREAM;
          GemmArgs<Op> ga{(const E*)ctx->X_op.p, d, (const E*)w.Win.p, d, M, 3 * d, d, 0};
          bool on_clip = false;
          if constexpr (Op::PREC == 0) {  // f32: the Q | K columns and the V columns as two clip launches (see the other branch)
            if (!(g_sel & 64) && ctx->hd == 128) {
              TAMF_CLIP_NSUB(Sp, {
                if (ClipLaunch<Op, 2, EpiQK<Op, true>, NS>::applies(B, Sp, 2 * d, d) && ClipLaunch<Op, 2, EpiVt<Op, true>, NS>::applies(B, Sp, d, d)) {
                  EpiQK<Op, true> eq{w.c2_in, (E*)ctx->QK_op.p, d, qscale, ACT_NONE, {w.Win.inv_scale, ctx->status}, ln_in};
                  if (ClipLaunch<Op, 4, EpiQK<Op, true>, NS>::applies(B, Sp, 2 * d, d))
                    HIPCHK(ctx, (ClipLaunch<Op, 4, EpiQK<Op, true>, NS>::launch(nullptr, ga.A, d, ga.W, d, B, Sp, 2 * d, d, eq, sk)));
                  else
                    HIPCHK(ctx, (ClipLaunch<Op, 2, EpiQK<Op, true>, NS>::launch(nullptr, ga.A, d, ga.W, d, B, Sp, 2 * d, d, eq, sk)));
                  const E* Wv = (const E*)((const char*)w.Win.p + (size_t)2 * d * d * Op::EB);
                  EpiVt<Op, true> ev{w.c2_in + 2 * d, (E*)ctx->Vt_op.p, ctx->H, ctx->hd, ctx->Skp, ACT_NONE, {w.Win.inv_scale, ctx->status}, ln_in};
                  mark("gemm_qk", BS * 2.0 * dd * 2 * dd);
                  sk = TAMF_NEXT_STREAM;
                  HIPCHK(ctx, (ClipLaunch<Op, 2, EpiVt<Op, true>, NS>::launch(nullptr, ga.A, d, Wv, d, B, Sp, d, d, ev, sk)));
                  mark("gemm_v", BS * 2.0 * dd * dd);
                  on_clip = true;
                }
              })
            }
          }
          if (!on_clip) {
            EpiQKV<Op, true> ep{w.c2_in, (E*)ctx->QK_op.p, (E*)ctx->Vt_op.p, d, ctx->H, ctx->hd, Sp, ctx->Skp, qscale, {w.Win.inv_scale, ctx->status}, ln_in};
            hipError_t es = hipSuccess;
            if (small_m_launch<Op>(ga, ep, sk, &es)) HIPCHK(ctx, es);
            else HIPCHK(ctx, gemm128<Op>(ga, ep, sk));
            mark("gemm_qkv", BS * 2.0 * dd * 3 * dd);
          }
        }
        {
          hipStream_t sk = TAMF_NEXT_STREAM;
          AttnArgs<Op> aa{(const E*)ctx->QK_op.p, (const E*)ctx->Vt_op.p, (E*)ctx->A_op.p, S, Sp, ctx->Skp, d, ctx->H, 0};
          HIPCHK(ctx, launch_attn<Op>(aa, B, ctx->hd, sk));
          mark("attention", 4.0 * B * (double)S * S * dd);
        }
        {
          hipStream_t sk = TAMF_NEXT_STREAM;
          GemmArgs<Op> ga{(const E*)ctx->A_op.p, d, (const E*)w.Wout.p, d, M, d, d, 0};
          EpiResid<Op> ep{w.bb_att, w.g_att, ctx->X, (E*)ctx->X_st, d, ctx->stat_att, ACT_NONE, {w.Wout.inv_scale, ctx->status}, ln_in};
          HIPCHK(ctx, launch_resid<Op>(ga, ep, B, Sp, sk));
          mark("gemm_outproj", BS * 2.0 * dd * dd);
        }
        {
          hipStream_t sk = TAMF_NEXT_STREAM;
          GemmArgs<Op> ga{(const E*)ctx->X_op.p, d, (const E*)w.W1.p, d, M, ff, d, 0};
          EpiBiasAct<Op, true> ep{w.c2_ff, nullptr, 0, (E*)ctx->H_op.p, ff, ACT_GELU, {w.W1.inv_scale, ctx->status}, ln_ff};
          bool on_clip = false;
          if (!(g_sel & 8)) {
            TAMF_CLIP_NSUB(Sp, {
              if (ClipLaunch<Op, 4, EpiBiasAct<Op, true>, NS>::applies(B, Sp, ff, d)) {
                HIPCHK(ctx, (ClipLaunch<Op, 4, EpiBiasAct<Op, true>, NS>::launch(nullptr, ga.A, d, ga.W, d, B, Sp, ff, d, ep, sk)));
                on_clip = true;
              }
            })
          }
          if (!on_clip) {
            hipError_t es = hipSuccess;
            if (small_m_launch<Op>(ga, ep, sk, &es)) HIPCHK(ctx, es);
            else HIPCHK(ctx, gemm128<Op>(ga, ep, sk));
          }
          mark("gemm_ffn1_gelu", BS * 2.0 * dd * ff);
        }
        {
          hipStream_t sk = TAMF_NEXT_STREAM;
          GemmArgs<Op> ga{(const E*)ctx->H_op.p, ff, (const E*)w.W2.p, ff, M, d, ff, 0};
          EpiResid<Op> ep{w.bb_ffn, w.g_ffn, ctx->X, (E*)ctx->X_st, d, ctx->stat_ffn, ACT_NONE, {w.W2.inv_scale, ctx->status}, ln_ff};
          HIPCHK(ctx, launch_resid<Op>(ga, ep, B, Sp, sk));
          mark("gemm_ffn2", BS * 2.0 * dd * ff);
        }
      }
    }
  }
  {
    // N = 128 is a single column tile: 64-row tiles (8 waves) double the workgroups that share the Philox-heavy epilogue
    hipStream_t sk = TAMF_NEXT_STREAM;
    GemmArgs<Op> ga{(const E*)ctx->X_op.p, d, (const E*)ctx->Wf.p, d, M, ctx->XN, d, 0};
    EpiHead<Op> head = head_in;
    head.t_off = t_off;
    hipError_t es = hipSuccess;
    if (small_m_launch<Op>(ga, head, sk, &es)) HIPCHK(ctx, es);  // (a few clips per call)
    else HIPCHK(ctx, (GemmLaunch<Op, 64, 128, EpiHead<Op>>::launch(ga, head, sk)));
    mark("gemm_head_ddpm", BT * 2.0 * dd * F);
  }
  ctx->step_kernels = nk;
  return 0;
}

template <class Op>
static EpiHead<Op> make_head(tamf_ctx* ctx, int mode) {
  EpiHead<Op> h{};
  h.bias = ctx->bf;
  h.mode = mode;
  h.F = ctx->F;
  h.T = ctx->T;
  h.Sp = ctx->Sp;
  h.P = ctx->P;
  h.XK = ctx->XK;
  h.xs = ctx->xs;
  h.xs_op = (typename Op::elem_t*)ctx->xs_st;
  h.tcur = ctx->tcur;
  h.c1 = ctx->c1;
  h.c2 = ctx->c2;
  h.sigma = ctx->sigma;
  h.n_steps = ctx->n_steps;
  h.lp = ctx->loop_params;
  h.ctl = EpiCtl{ctx->Wf.inv_scale, ctx->status};
  h.ln = LnStats{ctx->stat_ffn, ctx->d / 32, 1.0f / (float)ctx->d, 1e-5f};
  return h;
}

template <class Op>
static int denoise_impl(tamf_ctx* ctx, const float* x, const int64_t* t_dev, float* out, hipStream_t st) {
  typedef typename Op::elem_t E;
  const int B = ctx->B, T = ctx->T;
  hipLaunchKernelGGL((state_in_kernel<Op>), grid1d((long)B * T * (ctx->XK / 8)), dim3(256), 0, st, x, ctx->xs, (E*)ctx->xs_st,
                     B, ctx->F, T, ctx->XK, 0, 0ull, 0ll, ctx->status);
  hipLaunchKernelGGL(set_t_kernel, grid1d(B), dim3(256), 0, st, ctx->tcur, (const long long*)t_dev, 0, B, ctx->n_t > 0 ? ctx->n_t : 1);
  EpiHead<Op> h = make_head<Op>(ctx, HEAD_X0);
  h.x0_out = out;
  TRY(enqueue_step<Op>(ctx, st, h));
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

extern "C" int tamf_denoise(tamf_ctx* ctx, const float* x_dev, const int64_t* t_dev, float* x0_out_dev, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !x_dev || !t_dev || !x0_out_dev) return fail(ctx, TAMF_ERR_INVALID, "null argument");
  if (!ctx->cond_set) return fail(ctx, TAMF_ERR_STATE, "conditioning not set");
  if (ctx->arch.kind != TAMF_KIND_G) return fail(ctx, TAMF_ERR_STATE, "tamf_denoise needs a G context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  TAMF_WITH_OP(ctx->prec, return denoise_impl<Op>(ctx, x_dev, t_dev, x0_out_dev, st));
  return 0;
}

template <class Op>
static int refine_impl(tamf_ctx* ctx, const float* x_in, const float* h2o, float* out, hipStream_t st) {
  typedef typename Op::elem_t E;
  const int B = ctx->B, T = ctx->T;
  hipLaunchKernelGGL((refine_in_kernel<Op>), grid1d((long)B * T * (ctx->XK / 8)), dim3(256), 0, st, x_in, h2o, (E*)ctx->xs_op.p,
                     B * T, ctx->F, ctx->arch.h2o_dim, ctx->XK, ctx->status);
  EpiHead<Op> h = make_head<Op>(ctx, HEAD_RESIDUAL);
  h.x0_out = out;
  h.x_in = x_in;
  TRY(enqueue_step<Op>(ctx, st, h));
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

extern "C" int tamf_refine(tamf_ctx* ctx, const float* sample_pose_repr_dev, const float* h2o_dist_dev, float* out_dev,
                           void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !sample_pose_repr_dev || !h2o_dist_dev || !out_dev) return fail(ctx, TAMF_ERR_INVALID, "null argument");
  if (!ctx->cond_set) return fail(ctx, TAMF_ERR_STATE, "conditioning not set");
  if (ctx->arch.kind != TAMF_KIND_R) return fail(ctx, TAMF_ERR_STATE, "tamf_refine needs an R context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  TAMF_WITH_OP(ctx->prec, return refine_impl<Op>(ctx, sample_pose_repr_dev, h2o_dist_dev, out_dev, st));
  return 0;
}

extern "C" int tamf_ddpm_step(tamf_ctx* ctx, const float* x_t_dev, const float* x0_dev, int32_t t, const float* noise_dev,
                              float* x_out_dev, int64_t n, void* stream) {
  if (!ctx || !x_t_dev || !x0_dev || !x_out_dev || n <= 0) return fail(ctx, TAMF_ERR_INVALID, "null/invalid argument");
  if (ctx->n_steps <= 0) return fail(ctx, TAMF_ERR_STATE, "schedule not set");
  if (t < 0 || t >= ctx->n_steps) return fail(ctx, TAMF_ERR_INVALID, "t out of range");
  if (t != 0 && !noise_dev) return fail(ctx, TAMF_ERR_INVALID, "noise required for t != 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ddpm_step_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_t_dev, x0_dev, noise_dev, x_out_dev,
                     (long)n, ctx->h_c1[t], ctx->h_c2[t], ctx->h_sigma[t], t);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

template <class Op>
static int loop_impl(tamf_ctx* ctx, const float* noise, uint64_t seed, int64_t clip_base, float* out, float* dump,
                     int use_graph, hipStream_t st) {
  typedef typename Op::elem_t E;
  const int B = ctx->B, T = ctx->T, N = ctx->n_steps;
  // draw 0 = x_T
  hipLaunchKernelGGL((state_in_kernel<Op>), grid1d((long)B * T * (ctx->XK / 8)), dim3(256), 0, st, noise, ctx->xs,
                     (E*)ctx->xs_st, B, ctx->F, T, ctx->XK, noise ? 0 : 1, (unsigned long long)seed,
                     (long long)clip_base, ctx->status);
  hipLaunchKernelGGL(set_t_kernel, grid1d(B), dim3(256), 0, st, ctx->tcur, (const long long*)nullptr, N - 1, B, ctx->n_t > 0 ? ctx->n_t : 1);
  EpiHead<Op> h = make_head<Op>(ctx, HEAD_DDPM);
  struct InLoop {  // the step's timestep-embedding table is the respaced one (if a map is set) while the LOOP enqueues or captures
    tamf_ctx* c;
    explicit InLoop(tamf_ctx* c_) : c(c_) { c->in_loop = true; }
    ~InLoop() { c->in_loop = false; }
  } in_loop_(ctx);
  hipLaunchKernelGGL(set_loop_params_kernel, dim3(1), dim3(64), 0, st, ctx->loop_params, noise, dump, (long)B * ctx->F * T,
                     (unsigned long long)seed, (long long)clip_base);
  if (!use_graph) {
#ifdef TAMF_OVERLAP_PROBE
    ctx->pp_on = (g_sel & 256) != 0;
    if (ctx->pp_on) {  // the second stream starts behind the loop's set-up kernels
      if (ctx->pp_ev.empty()) {
        ctx->pp_ev.resize(64);
        for (auto& e : ctx->pp_ev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      ctx->pp_count = 0;
      HIPCHK(ctx, hipEventRecord(ctx->pp_ev[63], st));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->cap_stream, ctx->pp_ev[63], 0));
    }
#endif
    for (int i = 0; i < N; ++i) {
      TRY(enqueue_step<Op>(ctx, st, h));
#ifdef TAMF_OVERLAP_PROBE
      if (ctx->pp_on) { PingPong pp_{ctx, st, ctx->cap_stream}; hipLaunchKernelGGL(advance_t_kernel, grid1d(B), dim3(256), 0, pp_.next(), ctx->tcur, B, 1); continue; }
#endif
      hipLaunchKernelGGL(advance_t_kernel, grid1d(B), dim3(256), 0, st, ctx->tcur, B, 1);
    }
#ifdef TAMF_OVERLAP_PROBE
    if (ctx->pp_on) {  // join: the caller's stream continues behind both
      HIPCHK(ctx, hipEventRecord(ctx->pp_ev[63], ctx->cap_stream));
      HIPCHK(ctx, hipStreamWaitEvent(st, ctx->pp_ev[63], 0));
      ctx->pp_on = false;
    }
#endif
  } else {
    // G consecutive steps per graph (the largest divisor of N up to 16: 10 for N = 1000 -> 100 graph launches per loop);
    // the sequence is step-agnostic (device-side step counter) and seed-agnostic (LoopParams), so it is captured once per
    // (B, T, N) and replayed by every later loop
    int G = 1;
    for (int g = 2; g <= 16; ++g)
      if (N % g == 0) G = g;
    GraphKey key;
    key.B = B; key.T = T; key.n_steps = N; key.steps_per_graph = G;
    if (!(key == ctx->graph_key) || !ctx->graph_exec) {
      TRY(retire_graph(ctx));
      HIPCHK(ctx, hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed));
      int rc = 0;
      // step g of the graph works at t = counter - g (a kernel argument of its input-merge and head GEMMs); the counter itself
      // moves once per graph launch
      for (int g = 0; g < G && rc == 0; ++g) rc = enqueue_step<Op>(ctx, ctx->cap_stream, h, g);
      hipLaunchKernelGGL(advance_t_kernel, grid1d(B), dim3(256), 0, ctx->cap_stream, ctx->tcur, B, G);
      hipError_t ee = hipStreamEndCapture(ctx->cap_stream, &ctx->graph);
      if (rc) return rc;
      HIPCHK(ctx, ee);
      HIPCHK(ctx, hipGraphInstantiate(&ctx->graph_exec, ctx->graph, nullptr, nullptr, 0));
      ctx->graph_key = key;
      ++ctx->graph_captures;
    }
    for (int i = 0; i < N / G; ++i) HIPCHK(ctx, hipGraphLaunch(ctx->graph_exec, st));
    ctx->graph_launches_last_loop = N / G;
    HIPCHK(ctx, hipEventRecord(ctx->graph_done, st));
    ctx->graph_in_flight = true;
  }
  hipLaunchKernelGGL(state_out_kernel, grid1d((long)B * ctx->F * T), dim3(256), 0, st, ctx->xs, out, B, ctx->F, T, ctx->XK);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

extern "C" int tamf_sample_loop(tamf_ctx* ctx, const float* noise_dev, uint64_t seed, int64_t clip_id_base, float* x0_out_dev,
                                float* dump_dev, int32_t use_graph, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !x0_out_dev) return fail(ctx, TAMF_ERR_INVALID, "null argument");
  if (!ctx->cond_set) return fail(ctx, TAMF_ERR_STATE, "conditioning not set");
  if (ctx->n_steps <= 0) return fail(ctx, TAMF_ERR_STATE, "schedule not set");
  if (ctx->n_steps > ctx->n_t) return fail(ctx, TAMF_ERR_STATE, "schedule longer than the timestep table (max_timesteps)");
  if (ctx->arch.kind != TAMF_KIND_G) return fail(ctx, TAMF_ERR_STATE, "tamf_sample_loop needs a G context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  TAMF_WITH_OP(ctx->prec, return loop_impl<Op>(ctx, noise_dev, seed, clip_id_base, x0_out_dev, dump_dev, use_graph, st));
  return 0;
}

extern "C" int tamf_get_status_flags(tamf_ctx* ctx, uint32_t* flags, int32_t clear, void* stream) {
  if (!ctx || !flags) return fail(ctx, TAMF_ERR_INVALID, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
  // this context's own word (per-context since round 4: another context on the device neither sees nor clears it)
  unsigned v = 0;
  HIPCHK(ctx, hipMemcpy(&v, ctx->status, sizeof(v), hipMemcpyDeviceToHost));
  *flags = v | ctx->host_status;  // (the host-side bits describe the loaded weights: they are not cleared)
  if (clear && v) HIPCHK(ctx, hipMemset(ctx->status, 0, sizeof(v)));
  return 0;
}

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_set_guard_bytes(int64_t bytes) {
  if (bytes < 0 || bytes > (1 << 20) || bytes % 256) return fail(nullptr, TAMF_ERR_INVALID, "guard bytes must be a multiple of 256 in [0, 1 MiB]");
  g_guard_bytes.store((size_t)bytes);
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_fail_alloc_after(int32_t n) {
  g_fail_alloc_in.store(n);
  return 0;
}
#endif  // TAMF_TEST_HOOKS
#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_poke(tamf_ctx* ctx, int32_t alloc_index, int64_t offset, int32_t nbytes) {
  if (!ctx || alloc_index < 0 || (size_t)alloc_index >= ctx->guards.size() || nbytes <= 0) return fail(ctx, TAMF_ERR_INVALID, "bad argument");
  const GuardRec& g = ctx->guards[alloc_index];
  if (offset < -(int64_t)g.guard || offset + nbytes > (int64_t)(g.bytes + g.guard)) return fail(ctx, TAMF_ERR_INVALID, "outside the allocation and its margins");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemset(g.base + g.guard + offset, 0, nbytes));
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_check_guards(tamf_ctx* ctx, int32_t* n_checked) {
  if (!ctx) return fail(ctx, TAMF_ERR_INVALID, "null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  if (n_checked) *n_checked = (int32_t)ctx->guards.size();
  std::vector<unsigned char> host;
  int bad = 0;
  std::string report;
  for (size_t i = 0; i < ctx->guards.size(); ++i) {
    const GuardRec& g = ctx->guards[i];
    host.resize(g.guard);
    for (int side = 0; side < 2; ++side) {
      const char* src = side ? g.base + g.guard + g.bytes : g.base;
      HIPCHK(ctx, hipMemcpy(host.data(), src, g.guard, hipMemcpyDeviceToHost));
      size_t first = g.guard, last = 0, n = 0;
      for (size_t k = 0; k < g.guard; ++k)
        if (host[k] != GUARD_BYTE) {
          if (first == g.guard) first = k;
          last = k;
          ++n;
        }
      if (n) {
        ++bad;
        if (report.size() < 1500)
          report += std::string(report.empty() ? "" : "; ") + "allocation #" + std::to_string(i) + " [" + g.tag + "] of " +
                    std::to_string(g.bytes) + " bytes: " + std::to_string(n) + " bytes written " +
                    (side ? "BEYOND its end (offsets +" + std::to_string(first) + " .. +" + std::to_string(last) + ")"
                          : "BELOW its start (offsets -" + std::to_string(g.guard - first) + " .. -" + std::to_string(g.guard - last) + ")");
      }
    }
  }
  if (bad) return fail(ctx, TAMF_ERR_STATE, "out-of-bounds device stores: " + report);
  return 0;
}
#endif  // TAMF_TEST_HOOKS

extern "C" int tamf_step_kernel_count(const tamf_ctx* ctx) { return ctx ? ctx->step_kernels : 0; }

extern "C" int tamf_loop_stats(const tamf_ctx* ctx, int32_t* graph_captures, int32_t* graph_launches_last_loop) {
  if (!ctx) return TAMF_ERR_INVALID;
  if (graph_captures) *graph_captures = ctx->graph_captures;
  if (graph_launches_last_loop) *graph_launches_last_loop = ctx->graph_launches_last_loop;
  return 0;
}

template <class Op>
static int profile_impl(tamf_ctx* ctx, hipStream_t st) {
  EpiHead<Op> h = make_head<Op>(ctx, HEAD_DDPM);
  hipLaunchKernelGGL(set_loop_params_kernel, dim3(1), dim3(64), 0, st, ctx->loop_params, (const float*)nullptr, (float*)nullptr,
                     (long)ctx->B * ctx->F * ctx->T, 1ull, 0ll);
  return enqueue_step<Op>(ctx, st, h);
}

// Runs `body` (which enqueues launches through enqueue_step) with an event behind every launch and reports per launch: milliseconds,
// algorithmic FLOPs, name.
template <class Body>
static int profile_run(tamf_ctx* ctx, int32_t max_n, float* ms_host, double* flops_host, char* names_host, hipStream_t st, Body body) {
  ctx->prof_ev.clear();
  ctx->prof_names.clear();
  ctx->prof_flops.clear();
  hipEvent_t ev0;
  HIPCHK(ctx, hipEventCreate(&ev0));
  // cost of an event record between two launches: measured on back-to-back records with nothing in between and taken
  // off every interval below, so that the per-launch times agree with the profiler's kernel durations
  hipEvent_t cal[6];
  for (hipEvent_t& e : cal) HIPCHK(ctx, hipEventCreate(&e));
  ctx->prof_on = true;
  for (hipEvent_t& e : cal) (void)hipEventRecord(e, st);
  (void)hipEventRecord(ev0, st);
  int rc = body();
  ctx->prof_on = false;
  hipError_t se = hipStreamSynchronize(st);
  int n = 0;
  if (rc == 0 && se == hipSuccess) {
    float gaps[5];
    for (int i = 0; i < 5; ++i) {
      gaps[i] = 0.f;
      (void)hipEventElapsedTime(&gaps[i], cal[i], cal[i + 1]);
    }
    std::sort(gaps, gaps + 5);
    const float ev_cost = gaps[2];
    hipEvent_t prev = ev0;
    for (size_t i = 0; i < ctx->prof_ev.size() && n < max_n; ++i, ++n) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, prev, ctx->prof_ev[i]);
      ms = ms > ev_cost ? ms - ev_cost : ms;
      ms_host[n] = ms;
      flops_host[n] = ctx->prof_flops[i];
      snprintf(names_host + (size_t)n * 48, 48, "%s", ctx->prof_names[i].c_str());
      prev = ctx->prof_ev[i];
    }
  }
  for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : cal) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ev0);
  ctx->prof_ev.clear();
  if (rc) return rc;
  HIPCHK(ctx, se);
  return n;
}

extern "C" int tamf_step_profile(tamf_ctx* ctx, int32_t max_n, float* ms_host, double* flops_host, char* names_host,
                                 void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !ms_host || !flops_host || !names_host || max_n <= 0) return fail(ctx, TAMF_ERR_INVALID, "null/invalid argument");
  if (!ctx->cond_set || ctx->n_steps <= 0) return fail(ctx, TAMF_ERR_STATE, "conditioning / schedule not set");
  if (ctx->arch.kind != TAMF_KIND_G) return fail(ctx, TAMF_ERR_STATE, "needs a G context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_t_kernel, grid1d(ctx->B), dim3(256), 0, st, ctx->tcur, (const long long*)nullptr, ctx->n_steps / 2, ctx->B, ctx->n_t > 0 ? ctx->n_t : 1);
  return profile_run(ctx, max_n, ms_host, flops_host, names_host, st, [&]() {
    int rc = 0;
    TAMF_WITH_OP(ctx->prec, rc = profile_impl<Op>(ctx, st));
    return rc;
  });
}

extern "C" int tamf_refine_profile(tamf_ctx* ctx, const float* sample_pose_repr_dev, const float* h2o_dist_dev, float* out_dev,
                                   int32_t max_n, float* ms_host, double* flops_host, char* names_host, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (!ctx || !sample_pose_repr_dev || !h2o_dist_dev || !out_dev || !ms_host || !flops_host || !names_host || max_n <= 0)
    return fail(ctx, TAMF_ERR_INVALID, "null/invalid argument");
  if (!ctx->cond_set) return fail(ctx, TAMF_ERR_STATE, "conditioning not set");
  if (ctx->arch.kind != TAMF_KIND_R) return fail(ctx, TAMF_ERR_STATE, "tamf_refine_profile needs an R context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  return profile_run(ctx, max_n, ms_host, flops_host, names_host, st, [&]() {
    int rc = 0;
    TAMF_WITH_OP(ctx->prec, rc = refine_impl<Op>(ctx, sample_pose_repr_dev, h2o_dist_dev, out_dev, st));
    return rc;
  });
}

// ------------------------------------------------------------------------------------------------
// kernel-level test hooks
// ------------------------------------------------------------------------------------------------
struct TmpBufs {
  std::vector<void*> v;
  ~TmpBufs() {
    for (void* p : v) (void)hipFree(p);
  }
  void* get(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    v.push_back(p);
    return p;
  }
};

template <class Op>
static int test_gemm_impl(int M, int N, int K, const float* a, const float* w, const float* bias, int act, float* c,
                          const float* gamma, const float2* stats_in, float2* stats_out, bool resid, hipStream_t st) {
  typedef typename Op::elem_t E;
  if (prepare_all<Op>() != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "prepare failed");
  const int Kp = round_up(K, 64);
  TmpBufs tb;
  E* ao = (E*)tb.get((size_t)M * Kp * Op::EB);
  E* wo = (E*)tb.get((size_t)N * Kp * Op::EB);
  E* yo = (E*)tb.get((size_t)M * N * Op::EB);
  if (!ao || !wo || !yo) return fail(nullptr, TAMF_ERR_NOMEM, "hipMalloc failed");
  hipLaunchKernelGGL((pack_operand_kernel<Op>), grid1d((long)M * (Kp / 8)), dim3(256), 0, st, a, ao, (long)M, K, Kp);
  hipLaunchKernelGGL((pack_operand_kernel<Op>), grid1d((long)N * (Kp / 8)), dim3(256), 0, st, w, wo, (long)N, K, Kp);
  GemmArgs<Op> ga{ao, Kp, wo, Kp, M, N, Kp, 0};
  hipError_t e;
  if (resid) {
    // the residual GEMM of an encoder sublayer with the LayerNorm of its input deferred (EpiResid): c holds u on entry, u_next on return
    EpiResid<Op> ep{bias, gamma, c, Op::PREC == 0 ? nullptr : yo, N, stats_out, ACT_NONE, {}, LnStats{stats_in, N / 32, 1.0f / (float)N, 1e-5f}};
    // (the selection of the step: launch_resid for clip-aligned M - small tiles with the deep K pipeline when there are few of them - and
    // for every other M the small tiles where they fit, so that ragged tile edges of that kernel are tested too, else 128 x 128)
    const int sp = M % 208 == 0 ? 208 : (M % 168 == 0 ? 168 : 0);
    if (sp) e = launch_resid<Op>(ga, ep, M / sp, sp, st);
    else if (!small_m_launch<Op>(ga, ep, st, &e)) e = gemm128<Op>(ga, ep, st);
  } else {
    EpiStoreF32 ep{bias, c, N, act};
    // M = n * 208 rows (T = 196) or n * 168 rows (T = 160): the clip-aligned tiles the encoder layers use (same selection as
    // enqueue_step; 32 clips or fewer: the row-part tiles)
    const int nc = M / 208;
    bool done = false;
    if (M % 208 == 0 && N % 256 == 0 && ClipLaunch<Op, 4, EpiStoreF32>::applies(nc, 208, N, Kp)) {
      e = ClipLaunch<Op, 4, EpiStoreF32>::launch(nullptr, ao, Kp, wo, Kp, nc, 208, N, Kp, ep, st);
      done = true;
    } else {
      const int sp = M % 208 == 0 ? 208 : (M % 168 == 0 ? 168 : 0);
      if (sp) {
        TAMF_CLIP_NSUB(sp, {
          if (ClipLaunch<Op, 2, EpiStoreF32, NSP, 2>::applies_parts(M / sp, sp, N, Kp)) {
            e = ClipLaunch<Op, 2, EpiStoreF32, NSP, 2>::launch(nullptr, ao, Kp, wo, Kp, M / sp, sp, N, Kp, ep, st);
            done = true;
          } else if (ClipLaunch<Op, 2, EpiStoreF32, NS>::applies(M / sp, sp, N, Kp)) {
            e = ClipLaunch<Op, 2, EpiStoreF32, NS>::launch(nullptr, ao, Kp, wo, Kp, M / sp, sp, N, Kp, ep, st);
            done = true;
          }
        })
      }
    }
    if (!done) e = gemm128<Op>(ga, ep, st);
  }
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, std::string("gemm launch: ") + hipGetErrorString(e));
  if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "sync failed");
  e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, std::string("gemm run: ") + hipGetErrorString(e));
  return 0;
}

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_gemm(int32_t precision, int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev,
                              const float* bias_dev, int32_t act, float* c_dev, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (M <= 0 || N <= 0 || K <= 0 || N % 128) return fail(nullptr, TAMF_ERR_INVALID, "N must be a multiple of 128");
  hipStream_t st = (hipStream_t)stream;
  if (precision < 0 || precision > TAMF_PREC_F16X3) return fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  TAMF_WITH_OP(precision, return test_gemm_impl<Op>(M, N, K, a_dev, w_dev, bias_dev, act, c_dev, nullptr, nullptr, nullptr, false, st));
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_gemm_resid(int32_t precision, int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev,
                                    const float* bb_dev, const float* gamma_dev, const float* stats_in_dev, float* x_dev,
                                    float* stats_out_dev, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (M <= 0 || K <= 0 || !(N == 128 || N == 256 || N == 512)) return fail(nullptr, TAMF_ERR_INVALID, "N must be 128/256/512");
  if (!a_dev || !w_dev || !bb_dev || !gamma_dev || !x_dev || !stats_out_dev) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  if (precision < 0 || precision > TAMF_PREC_F16X3) return fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  TAMF_WITH_OP(precision, return test_gemm_impl<Op>(M, N, K, a_dev, w_dev, bb_dev, 0, x_dev, gamma_dev, (const float2*)stats_in_dev,
                                                    (float2*)stats_out_dev, true, st));
  return 0;
}
#endif  // TAMF_TEST_HOOKS

template <class Op>
static int test_attn_impl(int B, int S, int H, int hd, const float* qkv, float* out, hipStream_t st) {
  typedef typename Op::elem_t E;
  const int d = H * hd, Sp = round_up(S, 8), Skp = round_up(S, 32);
  const long M = (long)B * Sp;
  TmpBufs tb;
  const size_t qk_n = (size_t)M * 2 * d, vt_n = (size_t)B * d * Skp, o_n = (size_t)M * d;
  E* qk = (E*)tb.get(qk_n * Op::EB);
  E* vt = (E*)tb.get(vt_n * Op::EB);
  E* oo = (E*)tb.get(o_n * Op::EB);
  float* of = (float*)tb.get(o_n * 4);
  if (!qk || !vt || !oo || !of) return fail(nullptr, TAMF_ERR_NOMEM, "hipMalloc failed");
  (void)hipMemsetAsync(vt, 0, vt_n * Op::EB, st);
  const float qscale = 1.4426950408889634f / sqrtf((float)hd);
  hipLaunchKernelGGL((qkv_pack_kernel<Op>), grid1d(M * 3 * d), dim3(256), 0, st, qkv, qk, vt, B, S, Sp, Skp, H, hd, qscale);
  AttnArgs<Op> aa{qk, vt, oo, S, Sp, Skp, d, H, 0};
  hipError_t e = launch_attn<Op>(aa, B, hd, st);
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, std::string("attn launch: ") + hipGetErrorString(e));
  hipLaunchKernelGGL((unpack_operand_kernel<Op>), grid1d(M * d), dim3(256), 0, st, oo, of, M, d, d);
  // compact [B][Sp][d] -> [B][S][d]
  for (int b = 0; b < B; ++b)
    (void)hipMemcpyAsync(out + (size_t)b * S * d, of + (size_t)b * Sp * d, (size_t)S * d * 4, hipMemcpyDeviceToDevice, st);
  if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "sync failed");
  e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, std::string("attn run: ") + hipGetErrorString(e));
  return 0;
}

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_attention(int32_t precision, int32_t B, int32_t S, int32_t H, int32_t hd, const float* qkv_dev,
                                   float* out_dev, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (B <= 0 || S <= 0 || H <= 0 || !(hd == 64 || hd == 128)) return fail(nullptr, TAMF_ERR_INVALID, "bad attention shape");
  hipStream_t st = (hipStream_t)stream;
  if (precision < 0 || precision > TAMF_PREC_F16X3) return fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  TAMF_WITH_OP(precision, return test_attn_impl<Op>(B, S, H, hd, qkv_dev, out_dev, st));
  return 0;
}
#endif  // TAMF_TEST_HOOKS

// random operand fill for the kernel benchmarks (values in [-1, 1))
template <class Op>
__global__ void fill_operand_kernel(typename Op::elem_t* out, long n, unsigned salt) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned h = (unsigned)(i + j) * 2654435761u + salt;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    v[j] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
  }
  Op::template store<8>(out, i, v);
}

template <class Op>
static int bench_gemm_impl(int epi_kind, int M, int N, int K, int iters, float* ms_out, hipStream_t st) {
  typedef typename Op::elem_t E;
  if (prepare_all<Op>() != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "prepare failed");
  TmpBufs tb;
  const long an = (long)M * K, wn = (long)N * K, on = (long)M * N;
  E* a = (E*)tb.get((size_t)an * Op::EB);
  E* w = (E*)tb.get((size_t)wn * Op::EB);
  E* o = (E*)tb.get((size_t)on * Op::EB);
  E* o2 = (E*)tb.get((size_t)on * Op::EB);
  float* x = (float*)tb.get((size_t)on * 4);
  float* vec = (float*)tb.get((size_t)N * 4 * 4);
  if (!a || !w || !o || !o2 || !x || !vec) return fail(nullptr, TAMF_ERR_NOMEM, "hipMalloc failed");
  hipLaunchKernelGGL((fill_operand_kernel<Op>), grid1d(an / 8), dim3(256), 0, st, a, an, 1u);
  hipLaunchKernelGGL((fill_operand_kernel<Op>), grid1d(wn / 8), dim3(256), 0, st, w, wn, 2u);
  hipLaunchKernelGGL((fill_operand_kernel<OpF32>), grid1d(on / 8), dim3(256), 0, st, x, on, 3u);
  hipLaunchKernelGGL((fill_operand_kernel<OpF32>), grid1d(N * 4 / 8), dim3(256), 0, st, vec, (long)N * 4, 4u);
  GemmArgs<Op> ga{a, K, w, K, M, N, K, 0};
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "event");
  hipError_t e = hipSuccess;
  for (int it = -2; it < iters && e == hipSuccess; ++it) {
    if (it == 0) (void)hipEventRecord(e0, st);
    if (epi_kind == 2) {
      e = hipErrorInvalidValue;  // (rounds 2 - 5: the LayerNorm-fused 64 x d tile; gone - kind 12 is the residual GEMM now)
    } else if (epi_kind == 3) {
      EpiStoreF32 ep{vec, x, N, ACT_NONE};
      if (M % 208 == 0 && ClipLaunch<Op, 2, EpiStoreF32>::applies(M / 208, 208, N, K))
        e = ClipLaunch<Op, 2, EpiStoreF32>::launch(nullptr, a, K, w, K, M / 208, 208, N, K, ep, st);
      else
        e = gemm128<Op>(ga, ep, st);
    } else if (epi_kind == 1) {
      const int d = N / 3;
      EpiQKV<Op> ep{vec, o, o2, d, d / 128, 128, 208, 224, 0.1f};
      e = gemm128<Op>(ga, ep, st);
    } else if (epi_kind >= 10 && epi_kind <= 12) {
      // the deferred-LayerNorm forms: 10 = FFN1 with the row terms, 11 = QKV with the row terms, 12 = residual GEMM
      {
        const LnStats ln{(const float2*)x, K / 32, 1.0f / (float)K, 1e-5f};  // (any finite numbers: x is M x N >= M x K / 16 floats)
        if (epi_kind == 10) {
          EpiBiasAct<Op, true> ep{vec, nullptr, 0, o, N, ACT_GELU, {}, ln};
          if (M % 208 == 0 && ClipLaunch<Op, 4, EpiBiasAct<Op, true>>::applies(M / 208, 208, N, K))
            e = ClipLaunch<Op, 4, EpiBiasAct<Op, true>>::launch(nullptr, a, K, w, K, M / 208, 208, N, K, ep, st);
          else e = gemm128<Op>(ga, ep, st);
        } else if (epi_kind == 11) {
          const int d = N / 3;
          EpiQKV<Op, true> ep{vec, o, o2, d, d / 128, 128, 208, 224, 0.1f, {}, ln};
          e = gemm128<Op>(ga, ep, st);
        } else {
          EpiResid<Op> ep{vec, vec + N, x, o, N, (float2*)o2, ACT_NONE, {}, LnStats{nullptr, N / 32, 1.0f / (float)N, 1e-5f}};
          e = M % 208 == 0 ? launch_resid<Op>(ga, ep, M / 208, 208, st) : gemm128<Op>(ga, ep, st);
        }
      }
    } else {
      EpiBiasAct<Op> ep{vec, nullptr, 0, o, N, ACT_GELU};
      if (M % 208 == 0 && ClipLaunch<Op, 4, EpiBiasAct<Op>>::applies(M / 208, 208, N, K)) {
        e = ClipLaunch<Op, 4, EpiBiasAct<Op>>::launch(nullptr, a, K, w, K, M / 208, 208, N, K, ep, st);  // (ablations: ClipGemmArgs::abl)
      } else {
        if (g_krot >= 0 && (g_krot & 0x20000)) ep.ldo = 0;  // ablation: every row stores to the same (L2-resident) row - no HBM writes
        if (g_krot >= 0 && (g_krot & 0x8000)) ep.act = ACT_NONE;  // ablation: no GELU
        e = gemm128<Op>(ga, ep, st);
      }
    }
  }
  (void)hipEventRecord(e1, st);
  hipError_t se = hipStreamSynchronize(st);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess || se != hipSuccess)
    return fail(nullptr, TAMF_ERR_HIP, std::string("bench gemm: ") + hipGetErrorString(e != hipSuccess ? e : se));
  *ms_out = ms / iters;
  return 0;
}

// attention alone on random operands resident in HBM (tools/attn_bench.py): average ms of `iters` launches
template <class Op>
static int bench_attn_impl(int B, int S, int H, int hd, int iters, int abl, float* ms_out, hipStream_t st) {
  typedef typename Op::elem_t E;
  if (prepare_all<Op>() != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "prepare failed");
  const int d = H * hd, Sp = round_up(S, 8), Skp = round_up(S, 32);
  const long M = (long)B * Sp, qk_n = M * 2 * d, vt_n = (long)B * d * Skp, o_n = M * d;
  TmpBufs tb;
  E* qk = (E*)tb.get((size_t)qk_n * Op::EB);
  E* vt = (E*)tb.get((size_t)vt_n * Op::EB);
  E* oo = (E*)tb.get((size_t)o_n * Op::EB);
  if (!qk || !vt || !oo) return fail(nullptr, TAMF_ERR_NOMEM, "hipMalloc failed");
  hipLaunchKernelGGL((fill_operand_kernel<Op>), grid1d(qk_n / 8), dim3(256), 0, st, qk, qk_n, 5u);
  hipLaunchKernelGGL((fill_operand_kernel<Op>), grid1d(vt_n / 8), dim3(256), 0, st, vt, vt_n, 6u);
  AttnArgs<Op> aa{qk, vt, oo, S, Sp, Skp, d, H, abl};
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "event");
  hipError_t e = hipSuccess;
  for (int it = -2; it < iters && e == hipSuccess; ++it) {
    if (it == 0) (void)hipEventRecord(e0, st);
    e = launch_attn<Op>(aa, B, hd, st);
  }
  (void)hipEventRecord(e1, st);
  hipError_t se = hipStreamSynchronize(st);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess || se != hipSuccess)
    return fail(nullptr, TAMF_ERR_HIP, std::string("bench attention: ") + hipGetErrorString(e != hipSuccess ? e : se));
  *ms_out = ms / iters;
  return 0;
}

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_bench_attention(int32_t precision, int32_t B, int32_t S, int32_t H, int32_t hd, int32_t iters, int32_t abl,
                                    int32_t tuning, float* ms_out, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (B <= 0 || S <= 0 || H <= 0 || !(hd == 64 || hd == 128) || iters <= 0 || !ms_out) return fail(nullptr, TAMF_ERR_INVALID, "bad argument");
  if (precision < 0 || precision > TAMF_PREC_F16X3) return fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  const int saved_rot = g_krot, saved_sel = g_sel;
  tamf_set_gemm_tuning(tuning);
  int rc = 0;
  TAMF_WITH_OP(precision, rc = bench_attn_impl<Op>(B, S, H, hd, iters, abl, ms_out, (hipStream_t)stream));
  g_krot = saved_rot;
  g_sel = saved_sel;
  return rc;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_bench_mfma_rate(int32_t precision, int32_t millis, float* tflops_out, float* mhz_out, void* stream) {
  if (precision < 0 || precision > TAMF_PREC_F16X3 || millis <= 0 || millis > 20000 || !tflops_out) return fail(nullptr, TAMF_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* sink = nullptr;
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return fail(nullptr, TAMF_ERR_HIP, "no device");
  const int grid = 2 * cus, block = 512, iters = 4000;  // 2 workgroups x 8 waves per CU = 4 waves per SIMD; about 1 ms per launch
  if (hipMalloc(&sink, (size_t)grid * block * sizeof(float)) != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, "hipMalloc failed");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() {
    if (precision == TAMF_PREC_F32) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(grid), dim3(block), 0, st, sink, iters);
    else if (precision == TAMF_PREC_F16X3) hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(grid), dim3(block), 0, st, sink, iters);
    else hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(grid), dim3(block), 0, st, sink, iters);
  };
  // the first third of the time lets the power management settle, the rest is timed
  float ms1 = 0.f, ms = 0.f;
  launch();
  hipEventRecord(e0, st);
  launch();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms1, e0, e1);
  const int n_all = (int)(millis / (ms1 > 1e-3f ? ms1 : 1e-3f)) + 3, n_settle = n_all / 3, n = n_all - n_settle;
  for (int i = 0; i < n_settle; ++i) launch();
  hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1, st);
  hipError_t e = hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(sink);
  if (e != hipSuccess || !(ms > 0.f)) return fail(nullptr, TAMF_ERR_HIP, e != hipSuccess ? hipGetErrorString(e) : "no time measured");
  const double mfmas = (double)n * grid * (block / 64) * (double)iters * 8;
  const double flop_per_mfma = precision == TAMF_PREC_F32 ? 2.0 * 16 * 16 * 4 : 2.0 * 16 * 16 * 32;
  *tflops_out = (float)(mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
  // the clock this rate implies if the pipe issued one MFMA per 16 cycles (32 for the fp32 shape: 8 passes of 4 cycles): a LOWER bound of sclk
  if (mhz_out) *mhz_out = (float)(mfmas / (cus * 4.0) * (precision == TAMF_PREC_F32 ? 32.0 : 16.0) / (ms * 1e-3) / 1e6);
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_bench_gemm(int32_t precision, int32_t epi_kind, int32_t krot, int32_t M, int32_t N, int32_t K,
                               int32_t iters, float* ms_out, void* stream) {
  TAMF_LAUNCH_LOCK;
  if (M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !ms_out) return fail(nullptr, TAMF_ERR_INVALID, "bad argument");
  if (epi_kind == 1 && (N % 384 || M % 208)) return fail(nullptr, TAMF_ERR_INVALID, "qkv bench needs N = 3d, M multiple of 208");
  const int saved_rot = g_krot, saved_sel = g_sel;
  tamf_set_gemm_tuning(krot);  // -1 = per-kernel defaults; >= 0 = GemmArgs::krot bits (tamf_gemm.h) + selection overrides
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (precision < 0 || precision > TAMF_PREC_F16X3) rc = fail(nullptr, TAMF_ERR_INVALID, "unknown precision");
  else TAMF_WITH_OP(precision, rc = bench_gemm_impl<Op>(epi_kind, M, N, K, iters, ms_out, st));
  g_krot = saved_rot;
  g_sel = saved_sel;
  return rc;
}
#endif  // TAMF_TEST_HOOKS

// ------------------------------------------------------------------------------------------------
// geometry either side of the trunks
// ------------------------------------------------------------------------------------------------
extern "C" int tamf_pose_decode(const float* pose_repr_dev, int64_t n_frames, int32_t n_joints, float* tsl_out_dev,
                                float* quat_out_dev, void* stream) {
  if (!pose_repr_dev || !quat_out_dev || n_frames <= 0 || n_joints <= 0) return fail(nullptr, TAMF_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(pose_decode_kernel, grid1d(n_frames * n_joints), dim3(256), 0, (hipStream_t)stream, pose_repr_dev,
                     tsl_out_dev, quat_out_dev, (long)n_frames, n_joints, 3 + 6 * n_joints);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}

static int h2o_launch(const float* hand_verts_dev, const float* obj_traj_dev, const float* obj_points_dev,
                      const int32_t* obj_num_dev, int32_t B, int32_t T, int32_t V, int32_t nobj, int32_t P, float* h2o_out_dev,
                      float* frame_min_dev, void* stream) {
  if (!hand_verts_dev || !obj_traj_dev || !obj_points_dev || (!h2o_out_dev && !frame_min_dev))
    return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  if (B <= 0 || T <= 0 || V <= 0 || nobj <= 0 || P <= 0) return fail(nullptr, TAMF_ERR_INVALID, "bad shape");
  if (V > 256 * H2O_VPT) return fail(nullptr, TAMF_ERR_INVALID, "at most 1024 hand vertices per frame (MANO has 778)");
  if (B > 65535) return fail(nullptr, TAMF_ERR_INVALID, "batch too large for one launch");
  hipLaunchKernelGGL(h2o_dist_kernel, dim3(T, B), dim3(256), 0, (hipStream_t)stream, hand_verts_dev, obj_traj_dev,
                     obj_points_dev, (const int*)obj_num_dev, h2o_out_dev, frame_min_dev, T, V, nobj, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}

extern "C" int tamf_h2o_dist(const float* hand_verts_dev, const float* obj_traj_dev, const float* obj_points_dev,
                             const int32_t* obj_num_dev, int32_t B, int32_t T, int32_t V, int32_t nobj, int32_t P,
                             float* h2o_out_dev, void* stream) {
  if (!h2o_out_dev) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  return h2o_launch(hand_verts_dev, obj_traj_dev, obj_points_dev, obj_num_dev, B, T, V, nobj, P, h2o_out_dev, nullptr, stream);
}

extern "C" int tamf_contact_min_dist(const float* hand_verts_dev, const float* obj_traj_dev, const float* obj_points_dev,
                                     const int32_t* obj_num_dev, int32_t B, int32_t T, int32_t V, int32_t nobj, int32_t P,
                                     float* min_dist_out_dev, void* stream) {
  if (!min_dist_out_dev) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  return h2o_launch(hand_verts_dev, obj_traj_dev, obj_points_dev, obj_num_dev, B, T, V, nobj, P, nullptr, min_dist_out_dev, stream);
}

extern "C" int tamf_transform_points(const void* obj_traj_dev, const void* obj_points_dev, int32_t n_obj, int32_t T, int32_t P,
                                     int32_t is_f64, void* out_dev, void* stream) {
  if (!obj_traj_dev || !obj_points_dev || !out_dev) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  if (n_obj <= 0 || T <= 0 || P <= 0 || n_obj > 65535) return fail(nullptr, TAMF_ERR_INVALID, "bad shape");
  if (is_f64)
    hipLaunchKernelGGL((transform_points_kernel<double>), dim3(T, n_obj), dim3(256), 0, (hipStream_t)stream, (const double*)obj_traj_dev,
                       (const double*)obj_points_dev, (double*)out_dev, T, P);
  else
    hipLaunchKernelGGL((transform_points_kernel<float>), dim3(T, n_obj), dim3(256), 0, (hipStream_t)stream, (const float*)obj_traj_dev,
                       (const float*)obj_points_dev, (float*)out_dev, T, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}

extern "C" int tamf_vertex_normals(const float* verts_dev, int64_t n_mesh, int32_t V, const int32_t* csr_off_dev,
                                    const int32_t* csr_ent_dev, float* normals_out_dev, void* stream) {
  if (!verts_dev || !csr_off_dev || !csr_ent_dev || !normals_out_dev) return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  if (n_mesh <= 0 || V <= 0) return fail(nullptr, TAMF_ERR_INVALID, "bad shape");
  hipLaunchKernelGGL(vertex_normals_kernel, grid1d((long)n_mesh * V), dim3(256), 0, (hipStream_t)stream, verts_dev, csr_off_dev,
                     csr_ent_dev, normals_out_dev, (long)n_mesh, V);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}

extern "C" int tamf_mesh_contains(const double* verts_dev, const int32_t* faces_dev, int32_t n_faces, const double* points_dev,
                                  int64_t n_points, const double* scale3, const double* translate3, int32_t resolution,
                                  double* tri_workspace_dev, uint8_t* contains_out_dev, void* stream) {
  if (!verts_dev || !faces_dev || !points_dev || !scale3 || !translate3 || !tri_workspace_dev || !contains_out_dev)
    return fail(nullptr, TAMF_ERR_INVALID, "null argument");
  if (n_faces <= 0 || n_points <= 0 || resolution <= 1) return fail(nullptr, TAMF_ERR_INVALID, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mesh_prepare_kernel, grid1d(n_faces), dim3(256), 0, st, verts_dev, (const int*)faces_dev, n_faces, scale3[0],
                     scale3[1], scale3[2], translate3[0], translate3[1], translate3[2], tri_workspace_dev);
  hipLaunchKernelGGL(mesh_contains_kernel, grid1d(n_points), dim3(256), 0, st, tri_workspace_dev, n_faces, points_dev,
                     (long)n_points, scale3[0], scale3[1], scale3[2], translate3[0], translate3[1], translate3[2],
                     (double)resolution, contains_out_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_set_gemm_tuning(int32_t krot) {
  TAMF_LAUNCH_LOCK;
  // low 20 bits: GemmArgs::krot bits (all ones = keep the per-kernel defaults); bits 20..30: kernel-selection overrides (g_sel).
  // The words are process-global and a captured loop graph has the selection of its capture time baked in, so every live
  // context's graph is retired here: the next tamf_sample_loop re-captures with the new selection (same as tamf_denoise).
  // (selection bit 2048 - the row-block kernels - has no room above bit 30: it is "low 20 bits = 0x7FFFF", i.e. all ones but bit 19)
  int sel = krot >= 0 ? (krot >> 20) & 0x7FF : 0;
  const int low = krot & 0xFFFFF;
  int rot = -1;
  if (krot >= 0 && low == 0x7FFFF) sel |= 2048;
  else if (krot >= 0 && low != 0xFFFFF) rot = low;
  if (sel != g_sel || rot != g_krot) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (tamf_ctx* c : g_live_ctx) (void)retire_graph(c);
  }
  g_sel = sel;
  g_krot = rot;
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_test_philox(uint64_t seed, int64_t clip_id_base, int32_t draw, int32_t B, int32_t n_feat, int32_t T,
                                float* out_dev, void* stream) {
  if (B <= 0 || n_feat <= 0 || T <= 0 || !out_dev) return fail(nullptr, TAMF_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(philox_fill_kernel, grid1d((long)B * n_feat * T), dim3(256), 0, (hipStream_t)stream, out_dev,
                     (unsigned long long)seed, (long long)clip_id_base, (unsigned)draw, B, n_feat, T);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, TAMF_ERR_HIP, hipGetErrorString(e));
  return 0;
}
#endif  // TAMF_TEST_HOOKS

#ifdef TAMF_TIMELINE
// debug builds only (not part of include/tamf_hip.h): which = 0 GEMM (5 u64 per workgroup), 1 attention (4 u64)
#ifdef TAMF_TEST_HOOKS  // (test / measurement hook: include/tamf_hip_test.h, libtamf_hip_hooks.so only)
extern "C" int tamf_debug_timeline(int which, void* dst, size_t bytes) {
  if (which == 0) return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_ts), bytes, 0, hipMemcpyDeviceToHost);
  if (which == 2) return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_clip_ts), bytes, 0, hipMemcpyDeviceToHost);
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_ts), bytes, 0, hipMemcpyDeviceToHost);
}
#endif  // TAMF_TEST_HOOKS
#endif
