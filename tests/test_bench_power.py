"""CPU: bench.py's power summary (which card is this job's, means over the timed window) and the committed energy model it reprices."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_power_summary_picks_the_card_whose_power_rises():
    # three cards: card 0 busy all the time (another job), card 1 idle, card 2 = this job (idle, then 1300 W during the timed loops)
    rows = []
    for i in range(100):
        t = 1000.0 + 0.25 * i
        mine = 1300.0 if 1010.0 <= t <= 1020.0 else 280.0
        rows.append([t, 1350.0, 1950.0, 270.0, 100.0, mine, 2000.0 if mine > 1000 else 150.0])
    p = bench.PowerTrace.summarise(rows, 1000.0, 1010.0, 1020.0)
    assert p["card_column"] == 2 and p["cards_sampled"] == 3
    assert abs(p["watts"] - 1300.0) < 1e-6 and abs(p["sclk_mhz"] - 2000.0) < 1e-6 and p["samples"] > 30
    assert abs(p["watts_before_the_run"] - 280.0) < 1e-6
    assert p["card_matched_by"].startswith("power rise")
    # with the device's PCI address matched to a column, the busy neighbour cannot be picked even if IT rises more
    q = bench.PowerTrace.summarise(rows, 1000.0, 1010.0, 1020.0, column=1)
    assert q["card_column"] == 1 and abs(q["watts"] - 270.0) < 1e-6 and q["card_matched_by"] == "pci address"
    assert bench.PowerTrace.summarise([], 0, 1, 2) is None


def test_power_model_reprices_the_committed_ledger():
    m = bench.power_model("f16x3", {"watts": 1300.0}, 2.5)
    assert m is not None and m["source"].endswith("energy_model.json")
    with open(os.path.join(ROOT, m["source"])) as f:
        led = json.load(f)["modes"]["f16x3"]
    assert abs(m["joules_dynamic_modelled"] - led["joules_dynamic_model"]) < 1e-12
    # MFMA + bytes + other vector instructions = the modelled dynamic joules
    assert abs(m["joules_mfma"] + m["joules_bytes"] + (m["joules_valu"] or 0.0) - m["joules_dynamic_modelled"]) < 1e-9
    assert abs(m["predicted_ms"] - m["joules_dynamic_modelled"] / (1300.0 - m["idle_watts"]) * 1e3) < 1e-9
    assert m["measured_joules"] == 1300.0 * 2.5e-3
    assert bench.power_model("no-such-dtype", None, 1.0) is None


def test_csrc_digest_matches_the_committed_traffic_files():
    """hbm_traffic_*.json of the newest round carry the digest of the kernel sources they were measured with; bench.py flags a file
    taken with other sources (roofline.traffic_stale).  The digest itself must be stable and cover every kernel source."""
    d = bench.csrc_digest()
    assert len(d) == 16 and d == bench.csrc_digest()
    t, src, stale = bench.hbm_traffic("f16x3", "gemm_ffn1_gelu", 64, 196)
    assert t and src and stale in (True, False)
