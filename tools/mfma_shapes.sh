#!/bin/bash
# GPU: joules per multiply-accumulate of the two fp16 / bf16 matrix-instruction shapes (tools/micro/mfma_shapes.hip) -> $1 (default gpurun_out/r05/mfma_shapes.txt)
out=${1:-gpurun_out/r05/mfma_shapes.txt}; secs=${2:-6}
mkdir -p $(dirname $out)
rm -f /tmp/tamf_stop_sampler
python3 tools/power_sampler.py /tmp/mfma_shapes_power.txt /tmp/tamf_stop_sampler &
spid=$!
sleep 2
tools/micro/mfma_shapes $secs > /tmp/mfma_shapes_cases.txt 2>&1
touch /tmp/tamf_stop_sampler
wait $spid
python3 - > $out <<'PY'
rows = [list(map(float, l.split())) for l in open("/tmp/mfma_shapes_power.txt") if l[0] != "#" and l.strip()]
cases = [l.split() for l in open("/tmp/mfma_shapes_cases.txt") if l.startswith("CASE")]
ncard = (len(rows[0]) - 1) // 2
def mean(k, t0, t1, col):
    v = [r[1 + 2 * k + col] for r in rows if t0 + 1.0 <= r[0] <= t1]
    return sum(v) / max(1, len(v))
idle = next(c for c in cases if c[1] == "idle")
busy = next(c for c in cases if c[1] == "f16_16x16x32")
card = max(range(ncard), key=lambda k: mean(k, float(busy[2]), float(busy[3]), 0) - mean(k, float(idle[2]), float(idle[3]), 0))
p_idle = mean(card, float(idle[2]), float(idle[3]), 0)
print(f"card column {card} of {ncard}; idle {p_idle:.0f} W")
print("case              MAC/s        dense TFLOP/s   watts   sclk MHz   pJ per MAC (above idle)")
for c in cases:
    if c[1] == "idle": continue
    t0, t1, rate = float(c[2]), float(c[3]), float(c[4])
    w, f = mean(card, t0, t1, 0), mean(card, t0, t1, 1)
    print(f"{c[1]:15s} {rate:.4e}   {2 * rate / 1e12:8.0f}      {w:7.0f}   {f:7.0f}    {(w - p_idle) / rate * 1e12:6.3f}")
PY
cat $out
