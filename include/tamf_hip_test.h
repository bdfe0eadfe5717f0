/* tamf_hip_test.h - test hooks and measurement entry points of libtamf_hip_hooks.so (gfx950 / MI355X).
 *
 * NOT part of the drop-in surface (include/tamf_hip.h, libtamf_hip.so; INTEGRATION.md binds none of this).  The same sources compiled
 * with -DTAMF_TEST_HOOKS give libtamf_hip_hooks.so, which exports everything libtamf_hip.so does plus the functions below: kernel-level
 * test entry points (the same kernels the step runs), guard bands around device allocations, allocation-failure injection, kernel
 * benchmarks and the process-global kernel-selection override that the A/B measurements and the "every selection gives the same
 * bits" tests use.  Loaded by tests/ and tools/ (oakink2_tamf_amd.hip_backend.hooks() / use_test_hooks()) and by bench.py for the
 * register-only MFMA probe; never by the product path.
 */
#ifndef TAMF_HIP_TEST_H
#define TAMF_HIP_TEST_H

#include "tamf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel-level test hooks (used by tests/ only; same kernels the step uses) ------------------- */
/* C[M,N] = A[M,K] . W[N,K]^T + bias, optional activation (0 none, 1 silu, 2 gelu_erf); all f32 device buffers;
 * operands are converted to `precision` on the fly by the library's own pack kernels. */
int tamf_test_gemm(int32_t precision, int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev,
                   const float* bias_dev, int32_t act, float* c_dev, void* stream);
/* The residual GEMM of an encoder sublayer as the step runs it (the LayerNorm of its INPUT deferred, EpiResid; N == latent width,
 * 128 / 256 / 512; reference: nn.TransformerEncoderLayer's x + sublayer(x) behind norm1 / norm2, interaction_segment_mdm.py:63-70):
 *   x[m][n] <- ((x[m][n] - mean[m]) rstd[m] gamma[n] + bb[n]) + (A . W^T)[m][n]        (in place, f32)
 * with (mean, rstd) of row m from stats_in[m][N / 32] = (S_b, Q_b), the sum and the sum of squares about its own mean of every
 * 32-column block of the row on entry (null: no LayerNorm in front, gamma = ones expected), and stats_out the same of the row on return. */
int tamf_test_gemm_resid(int32_t precision, int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev,
                         const float* bb_dev, const float* gamma_dev, const float* stats_in_dev, float* x_dev,
                         float* stats_out_dev, void* stream);
/* out[b,s,h*hd+e] = softmax(q k^T / sqrt(hd)) v per (b,h); qkv_dev: (B, S, 3*H*hd) f32 packed [q|k|v]. */
int tamf_test_attention(int32_t precision, int32_t B, int32_t S, int32_t H, int32_t hd, const float* qkv_dev,
                        float* out_dev, void* stream);
/* Guard bands (the out-of-bounds check that stands in for GPU AddressSanitizer, which gfx950 lacks here).  After
 * tamf_test_set_guard_bytes(n) - n a multiple of 256, 0 switches it off - every device allocation of contexts created by THIS process
 * from then on (activations, operand planes, V^T, scratch, weights, tables) is n bytes longer at both ends and the margins hold a
 * pattern.  tamf_test_check_guards synchronises the device and verifies all margins of `ctx`: 0 if intact, TAMF_ERR_STATE with the
 * offending allocations (creation expression, byte offsets) in tamf_last_error otherwise; *n_checked (may be NULL) = number of guarded
 * allocations.  Test hooks: no product path calls them. */
int tamf_test_set_guard_bytes(int64_t bytes);
int tamf_test_check_guards(tamf_ctx* ctx, int32_t* n_checked);
/* The checker's own test: zero nbytes at `offset` from the start of guarded allocation #alloc_index (negative / beyond-the-end offsets
 * reach into its margins). */
int tamf_test_poke(tamf_ctx* ctx, int32_t alloc_index, int64_t offset, int32_t nbytes);
/* Failure injection for the allocation paths: the (n + 1)-th device allocation this process makes from now on fails with
 * TAMF_ERR_NOMEM (n = 0: the next one; -1 disarms).  tests/test_hip_guardbands.py uses it to prove that a tamf_ctx_resize which
 * runs out of memory leaves the context working at its old size. */
int tamf_test_fail_alloc_after(int32_t n);
/* Philox normal draws exactly as the sampling loop generates them: out (B, n_feat, 1, T). */
int tamf_test_philox(uint64_t seed, int64_t clip_id_base, int32_t draw, int32_t B, int32_t n_feat, int32_t T,
                     float* out_dev, void* stream);

/* ---- kernel benchmarks / tuning (tools/kbench.py) ------------------------------------------------ */
/* Average milliseconds of `iters` launches of one GEMM on random operands already resident in HBM.
 * epi_kind: 0 = bias + GELU (FFN1 shape; clip tiles when M is a multiple of 208); 1 = 128x128 tile, QKV epilogue (N = 3d);
 * 3 = bias, fp32 output; the forms the step runs (deferred LayerNorm, csrc/tamf_device.h): 10 = FFN1 with the row factors,
 * 11 = QKV with the row factors, 12 = residual GEMM (EpiResid).  (2 = the LayerNorm-fused 64xN tile of rounds 1 - 5: gone, invalid.)
 * krot: -1 = default tuning, >= 0 = GemmArgs::krot bits (csrc/tamf_gemm.h: K-loop rotation, L2 touch-prefetch distance, ablation flags). */
int tamf_bench_gemm(int32_t precision, int32_t epi_kind, int32_t krot, int32_t M, int32_t N, int32_t K,
                    int32_t iters, float* ms_out, void* stream);
/* Average milliseconds of `iters` launches of the attention kernel alone on random Q | K / V^T operands resident in HBM (B clips,
 * S tokens, H heads of hd).  tuning: tamf_set_gemm_tuning word for the call (-1 = defaults; selection bit 512 = streaming kernel);
 * abl: ablation bits of csrc/tamf_attn.h AttnArgs::abl (honoured by -DTAMF_BENCH builds only). */
int tamf_bench_attention(int32_t precision, int32_t B, int32_t S, int32_t H, int32_t hd, int32_t iters, int32_t abl, int32_t tuning,
                         float* ms_out, void* stream);
/* What the matrix pipe of the current device sustains by itself: register-only MFMA loops (the mode's instruction: v_mfma_f32_16x16x4_f32,
 * _16x16x32_bf16 or _16x16x32_f16) with random full-mantissa operands on every SIMD for about `millis` ms; the last two thirds are
 * timed.  tflops_out: dense TFLOP/s (2 * 16 * 16 * K per MFMA); mhz_out (optional): the shader clock that rate implies at one MFMA per
 * 16 cycles per SIMD (32 for the fp32 shape).  bench.py reports it beside the nominal peak: with real operand bits MI355X reaches its
 * power management well below 2.5 PFLOP/s. */
int tamf_bench_mfma_rate(int32_t precision, int32_t millis, float* tflops_out, float* mhz_out, void* stream);
/* MEASUREMENT HOOK, not part of the drop-in surface (tools/ only; INTEGRATION.md does not bind it).  Overrides the GEMM tuning /
 * kernel-selection bits for every subsequent launch (-1 restores the per-kernel defaults); process-global; serialised against
 * every entry point that enqueues kernels (one process-wide lock), and
 * retires the captured loop graphs of all live contexts so that the next tamf_sample_loop re-captures with the new selection.
 * The ablation bits (no loads / no MFMAs / no epilogue) only exist in -DTAMF_BENCH builds of the library. */
int tamf_set_gemm_tuning(int32_t krot);

#ifdef __cplusplus
}
#endif
#endif /* TAMF_HIP_TEST_H */
