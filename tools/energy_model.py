"""Energy model of a DDPM step from tools/energy_model.sh's measurements ("power").

  python tools/energy_model.py [dir with energy_cases.txt + energy_power_samples.txt] [--json out.json]

1. joules per unit of every single-resource case: (mean package power of the case - idle power) / rate.
2. the step's resource counts (MFMA instructions, fabric bytes from the committed PMC files, bytes staged L2 -> LDS, LDS fragment
   bytes, VALU instructions - formulas below, B = 64, T = 196, arch_mdm_l), priced with (1).
3. predicted step time = (sum of dynamic joules) / (cap - idle power) if the board is at its cap, against the measured loops.
"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/r04"
cases = []
for line in open(os.path.join(d, "energy_cases.txt")):
    p = line.split()
    if p and p[0] == "CASE":
        cases.append(dict(name=p[1], t0=float(p[2]), t1=float(p[3]), rate=float(p[4]), unit=p[5]))
rows = [list(map(float, l.split())) for l in open(os.path.join(d, "energy_power_samples.txt")) if l[0] != "#"]
ncard = (len(rows[0]) - 1) // 2


def card_samples(k):
    return [(r[0], r[1 + 2 * k], r[2 + 2 * k]) for r in rows if len(r) == 1 + 2 * ncard]


def window_mean(smp, c, skip=1.0):
    xs = [(w, m) for t, w, m in smp if c["t0"] + skip <= t <= c["t1"] - 0.3]
    return (sum(w for w, _ in xs) / len(xs), sum(m for _, m in xs) / len(xs), len(xs)) if xs else (float("nan"), float("nan"), 0)


# the card of THIS job: the one whose power rises most from the idle case to the MFMA case (other cards belong to other jobs)
_c = {c["name"]: c for c in cases}
card = max(range(ncard), key=lambda k: window_mean(card_samples(k), _c["mfma_f16_random"])[0] - window_mean(card_samples(k), _c["idle"])[0])
samples = card_samples(card)
print(f"{ncard} card(s) sampled; this job's card: column {card}")


def mean_power(c, skip=1.0):
    return window_mean(samples, c, skip)


by = {}
for c in cases:
    c["watts"], c["mhz"], c["n"] = mean_power(c)
    by[c["name"]] = c
idle = min(by["idle"]["watts"], by["idle2"]["watts"])
print(f"idle power {by['idle']['watts']:.0f} W before / {by['idle2']['watts']:.0f} W after the cases (hot)  -> idle = {idle:.0f} W")
print(f"{'case':26s} {'W':>7s} {'sclk':>6s} {'rate':>12s} unit      joules per unit (dynamic)")
jpu = {}
for c in cases:
    if c["rate"] > 0 and not c["name"].startswith("loop_"):
        jpu[c["name"]] = (c["watts"] - idle) / c["rate"]
        print(f"{c['name']:26s} {c['watts']:7.0f} {c['mhz']:6.0f} {c['rate']:12.4e} {c['unit']:9s} {jpu[c['name']] * 1e12:10.3f} pJ")
    elif c["name"].startswith("loop_"):
        print(f"{c['name']:26s} {c['watts']:7.0f} {c['mhz']:6.0f} {c['rate']:12.4e} {c['unit']:9s} ({1e3 / c['rate']:.3f} ms per step)")

# ---- resource counts of one DDPM step, B = 64, T = 196 (Sp = 208, M = 13312 rows), arch_mdm_l (d = 512, ff = 2048, 8 layers) ----
B, Sp, d, ff, L, Hh, hd = 64, 208, 512, 2048, 8, 4, 128
M = B * Sp


def counts(mode):
    split = mode in ("f16x3", "bf16x3")
    per_prod = 3 if split else 1
    eb = 2 if mode == "bf16" else 4              # operand bytes per element
    kel = 128 // eb if mode != "f32" else 32     # elements per 128-byte K tile
    flop_mfma = 2 * 16 * 16 * (4 if mode == "f32" else 32)
    # MFMA instructions (wave level): padded rows, every GEMM of the step; attention scores over 14 key tiles, P.V over 7 blocks of 32
    gemm_macs = L * M * (3 * d * d + d * d + 2 * d * ff) + B * 196 * (128 * d + d * d + d * 128)
    attn_macs = L * B * Hh * (Sp * 224 * hd + Sp * 224 * hd)
    mfma = (gemm_macs + attn_macs) * 2 / flop_mfma * per_prod
    # bytes staged L2 -> LDS per layer: tile rows x 128 B x K tiles x tiles
    def staged(tiles, rows, K):
        return tiles * rows * 128 * (K * eb // 128)
    if mode == "bf16":
        lds_stage = staged(1248, 256, d) + staged(208, 64 + 512, d) + staged(512, 208 + 256, d) + staged(208, 64 + 512, ff)
    else:
        lds_stage = staged(1248, 256, d) + staged(208, 64 + 512, d) + staged(512, 208 + 256, d) + staged(256, 208 + 128, ff)
    attn_stage = B * Hh * (Sp * hd * eb + 7 * hd * 32 * eb)
    lds_stage = L * (lds_stage + attn_stage)
    # LDS fragment bytes read: every MFMA product reads (rows_a + rows_b) / (rows_a * rows_b) fragments of 2 KiB per wave tile;
    # wave tiles: 128 x 128 kernel 2 x 4 MFMA tiles (8 waves 4 x 2), LN tile 2 x 4, clip FFN1 6.5 x 4, clip FFN2 6.5 x 2 (X / Y mean)
    def frag_bytes(macs, mr, nc):
        products = macs / (16 * 16 * kel)  # (row tile, col tile, K tile) triples
        return products * (mr + nc) / (mr * nc) * 2048
    lds_read = L * (frag_bytes(M * 3 * d * d, 2, 4) + frag_bytes(M * d * d, 2, 4) + frag_bytes(M * d * ff, 6.5, 4)
                    + frag_bytes(M * d * ff, 6.5, 2) + frag_bytes(B * Hh * Sp * 224 * hd * 2, 14, 1))
    return dict(mfma=mfma, lds_stage=lds_stage, lds_read=lds_read)


def fabric(mode):
    """fabric-side bytes per step from the committed PMC files (FETCH corrected, WRITE): per layer kernels x 8 + the three others"""
    for rnd in ("r04", "r03"):
        p = os.path.join("profiles", rnd, f"hbm_traffic_{mode}.json")
        if os.path.exists(p):
            k = json.load(open(p))["kernels"]
            per_layer = [n for n in k if n not in ("gemm_head_ddpm", "gemm_input_merge2", "gemm_input_merge0")]
            rd = sum(k[n]["fetch_bytes_corrected"] for n in per_layer) * (2 if "outproj_residual_ln" not in k and mode == "f32" else 1)
            wr = sum(k[n]["write_bytes"] for n in per_layer)
            # (the files carry one entry per kernel CLASS; in the split modes the residual-LN class runs once per layer, FFN2's)
            oth = [n for n in k if n not in per_layer]
            return L * rd + sum(k[n]["fetch_bytes_corrected"] for n in oth), L * wr + sum(k[n]["write_bytes"] for n in oth), p
    return None, None, None


def pmc_totals(mode):
    """measured instruction counts of one step (tools/pmc_step_totals.sh), newest round first; None when absent"""
    for rnd in ("r04",):
        p = os.path.join("profiles", rnd, f"pmc_step_totals_{mode}.json")
        if os.path.exists(p):
            return json.load(open(p))["per_step"], p
    return None, None


cap = 1400.0
print()
out = {"idle_watts": idle, "cap_watts": cap, "joules_per_unit_pJ": {k: v * 1e12 for k, v in jpu.items()}, "modes": {}}
for mode, mf_case in (("f16x3", "mfma_f16_hilo"), ("bf16x3", "mfma_bf16_hilo"), ("bf16", "mfma_bf16_random"), ("f32", "mfma_f32_random")):
    if "loop_" + mode not in by:
        continue
    c = counts(mode)
    rd, wr, src = fabric(mode)
    if rd is None:
        continue
    meas = by["loop_" + mode]
    t_meas = 1.0 / meas["rate"]
    e = {
        "mfma": c["mfma"] * jpu[mf_case],
        "fabric_read": rd * jpu["read_same_109MB"],
        "fabric_write": wr * jpu["write_same_109MB"],
        "l2_to_lds": c["lds_stage"] * jpu["dma_l2_to_lds"],
        "lds_fragment_reads": c["lds_read"] * jpu["lds_read_b128"],
    }
    pm, pm_src = pmc_totals(mode)
    if pm:  # non-MFMA vector instructions (wave level), counted by the hardware; priced with the fma loop's joules per instruction
        e["valu"] = max(0.0, pm["SQ_INSTS_VALU"] - pm["SQ_INSTS_MFMA"]) * jpu["valu_fma"]
        if "salu" in jpu and "SQ_INSTS_SALU" in pm:
            e["salu"] = pm["SQ_INSTS_SALU"] * jpu["salu"]
    if "mfma_f16_from_lds" in jpu:
        # MFMAs whose operands change with every instruction (fed from LDS as in a K loop: 20 KiB of fragment reads per 72 MFMAs) cost
        # more than the register-only loop, which re-uses 8 operand registers: the factor measured on f16 is applied to every mode
        ref = jpu["mfma_f16_hilo"] + 20 * 1024 / 72.0 * jpu["lds_read_b128"]
        e["mfma_operand_refresh"] = e["mfma"] * (jpu["mfma_f16_from_lds"] / ref - 1.0)
    dyn = sum(e.values())
    e_meas = meas["watts"] * t_meas
    t_pred_cap = dyn / (cap - idle)                       # if the step ran at the cap the whole time
    t_pred_same_power = dyn / (meas["watts"] - idle)       # at the power the loop actually drew
    out["modes"][mode] = {"measured_ms": t_meas * 1e3, "measured_watts": meas["watts"], "measured_sclk_mhz": meas["mhz"],
                          "measured_joules_per_step": e_meas, "joules_idle": idle * t_meas, "joules_dynamic_measured": e_meas - idle * t_meas,
                          "joules_model": e, "joules_dynamic_model": dyn, "predicted_ms_at_cap": t_pred_cap * 1e3,
                          "predicted_ms_at_measured_power": t_pred_same_power * 1e3, "counts": c, "fabric_read_bytes": rd, "fabric_write_bytes": wr, "fabric_source": src, "pmc_source": pm_src,
                          "pmc_mfma_instructions": pm["SQ_INSTS_MFMA"] if pm else None}
    print(f"{mode}: measured {t_meas * 1e3:.3f} ms/step at {meas['watts']:.0f} W ({meas['mhz']:.0f} MHz) = {e_meas:.3f} J/step, of which idle {idle * t_meas:.3f} J, dynamic {e_meas - idle * t_meas:.3f} J")
    print("   model: " + "  ".join(f"{k} {v:.3f} J" for k, v in e.items()) + f"  = {dyn:.3f} J dynamic")
    print(f"   predicted step: {t_pred_same_power * 1e3:.3f} ms at the loop's own power, {t_pred_cap * 1e3:.3f} ms at the {cap:.0f} W cap   (measured {t_meas * 1e3:.3f} ms; model / measured dynamic energy = {dyn / (e_meas - idle * t_meas):.2f})")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
