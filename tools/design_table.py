"""Per-kernel table of one round's measurements, as DESIGN.md section 6 shows it:  python tools/design_table.py profiles/r05 > profiles/r05/kernel_table.md

Inputs (all written by tools/collect_round_profiles.sh on the GPU box and copied into profiles/rNN/):
  step_profile_hipevents_<mode>.json   bench.py --profile-out: per-launch HIP-event times of one DDPM step + algorithmic GFLOP
  hbm_traffic_<mode>.json              rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel class and launch (fabric side, corrected)
  rocprofv3_kernel_stats_<mode>_*.csv  (optional) rocprofv3 --kernel-trace --stats: average duration per kernel symbol
Columns: launches per step, us per launch (HIP events), algorithmic GFLOP (the reference's unfused formulas on the true S = T + 5 rows,
SURVEY.md 8d), EXECUTED GFLOP (what the launch really multiplies: padded rows, the K = 128 fusion of input_merge.0, the 128-column head,
208 queries x 224 keys per (clip, head) in attention) with its own fraction of the peak - which can never exceed 100 % -, TFLOP/s,
fraction of the dense MFMA peak of the mode
(2.5 PFLOP/s for the 16-bit MFMAs - the split modes issue three per product and are still quoted against it - 157.3 TFLOP/s fp32),
counted fabric bytes per launch, algorithmic bytes per launch (every operand read once, every output written once, weights once),
and their ratio."""
import json
import os
import sys

PEAK = {"f32": 157.3, "f16x3": 2500.0, "bf16x3": 2500.0, "bf16": 2500.0}
EB = {"f32": 4, "f16x3": 4, "bf16x3": 4, "bf16": 2}  # bytes per operand element


def algorithmic_bytes(kernel, mode, B, T, d=512, ff=2048, P=5):
    S = T + P
    Sp = (S + 7) // 8 * 8
    M, eb = B * Sp, EB[mode]
    act = lambda cols: M * cols * eb       # an operand matrix of M rows
    f32 = lambda cols: M * cols * 4
    opw = lambda cols: 0 if mode == "f32" else act(cols)  # a separate operand copy of an fp32 result (f32: the fp32 buffer IS the operand)
    t = {
        "gemm_qkv": act(d) + 3 * d * d * eb + act(3 * d),
        "gemm_qk": act(d) + 2 * d * d * eb + act(2 * d),
        "gemm_v": act(d) + d * d * eb + act(d),
        "attention": act(3 * d) + act(d),
        "gemm_ffn1_gelu": act(d) + ff * d * eb + act(ff),
        # residual GEMMs (deferred LayerNorm): operand in, weight, residual fp32 in and out, operand out (16-bit modes)
        "gemm_outproj": act(d) + d * d * eb + 2 * f32(d) + opw(d),
        "gemm_ffn2": act(ff) + d * ff * eb + 2 * f32(d) + opw(d),
        "outproj_residual_ln": 3 * f32(d),  # (rounds 2 - 4, f32 until round 5: product + residual in, state out)
        "ffn2_residual_ln": 3 * f32(d),
        "gemm_outproj_ln": act(d) + d * d * eb + 2 * f32(d) + act(d),
        "gemm_ffn2_ln": act(ff) + d * ff * eb + 2 * f32(d) + act(d),
        "gemm_input_merge0": B * T * (128 * eb + d * 4 + d * eb) + d * 128 * eb,
        "gemm_input_merge2": B * T * d * eb + d * d * eb + f32(d) + opw(d),
        "gemm_head_ddpm": act(d) + 128 * d * eb + B * T * 128 * (4 + 4 + (0 if mode == "f32" else eb)),
    }
    return t.get(kernel)


def executed_gflop(kernel, B, T, d=512, ff=2048, P=5, H=4):
    """MACs x 2 the launch really performs (padding and fusions included)"""
    S = T + P
    Sp = (S + 7) // 8 * 8
    M, BT = B * Sp, B * T
    up = lambda v, m: (v + m - 1) // m * m
    q, kk = up(S, 16), up(S, 32)  # queries of a (clip, head) in whole 16-row tiles, keys in whole 32-key blocks
    t = {
        "gemm_qkv": 2.0 * M * 3 * d * d, "gemm_qk": 2.0 * M * 2 * d * d, "gemm_v": 2.0 * M * d * d,
        "attention": 4.0 * B * H * q * kk * (d // H),
        "gemm_outproj": 2.0 * M * d * d, "gemm_ffn1_gelu": 2.0 * M * d * ff, "gemm_ffn2": 2.0 * M * ff * d,
        "gemm_input_merge0": 2.0 * up(BT, 128) * 128 * d,   # pose (99 -> 128) through the composed input_process . input_merge.0 weight
        "gemm_input_merge2": 2.0 * up(BT, 128) * d * d,
        "gemm_head_ddpm": 2.0 * up(M, 64) * d * 128,          # N = 99 -> 128 columns, every token row tile
    }
    v = t.get(kernel)
    return v / 1e9 if v else None


def table(pdir, mode):
    prof = json.load(open(os.path.join(pdir, f"step_profile_hipevents_{mode}.json")))
    try:
        traffic = json.load(open(os.path.join(pdir, f"hbm_traffic_{mode}.json")))
    except OSError:
        traffic = {"kernels": {}}
    B, T = prof["B"], prof["T"]
    out = [f"**{mode}** - B = {B}, T = {T}; event sum of one step {prof['step_ms_eventsum'] * 1e3:.0f} us"
           + (f"; counters taken at `{traffic.get('commit')}`" if traffic.get("commit") else ""), "",
           "| launch | per step | us | algorithmic GFLOP | TFLOP/s | of peak | executed GFLOP | executed, of peak | counted MB | algorithmic MB | counted / algorithmic |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k in prof["kernels"]:
        name = k["kernel"]
        tr = traffic["kernels"].get(name, {}).get("traffic_bytes_per_launch")
        ab = algorithmic_bytes(name, mode, B, T)
        ex = executed_gflop(name, B, T)
        out.append("| `%s` | %d | %.1f | %.2f | %.0f | %.1f %% | %s | %s | %s | %s | %s |" % (
            name, k["launches_per_step"], k["avg_ms"] * 1e3, k["algorithmic_gflop_per_launch"], k["tflops"], 100.0 * k["tflops"] / PEAK[mode],
            "%.2f" % ex if ex else "-", "%.1f %%" % (100.0 * ex / (k["avg_ms"] * 1e-3) / 1e3 / PEAK[mode]) if ex else "-",
            "%.0f" % (tr / 1e6) if tr else "-", "%.0f" % (ab / 1e6) if ab else "-", "%.2f" % (tr / ab) if tr and ab else "-"))
    return "\n".join(out)


def main():
    pdir = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05"
    parts = []
    for mode in ("f16x3", "f32", "bf16", "bf16x3"):
        if os.path.exists(os.path.join(pdir, f"step_profile_hipevents_{mode}.json")):
            parts.append(table(pdir, mode))
    print("\n\n".join(parts))


if __name__ == "__main__":
    main()
