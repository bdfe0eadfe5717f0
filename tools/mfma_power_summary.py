"""Summarise gpurun_out/mfma_power.log (tools/mfma_power.sh): per case TFLOP/s with the rocm-smi samples taken while it ran."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_power.log"
smi, cases = [], []
for l in open(path):
    m = re.search(r"\[t=(\d+)\].*sclk clock level: \d: \((\d+)Mhz\).*Power \(W\): ([\d.]+)", l)
    if m:
        smi.append((int(m.group(1)), int(m.group(2)), float(m.group(3))))
        continue
    m = re.search(r"\[t=(\d+)\] (.*?)\s+([\d.]+) TFLOP/s.*launches in (\d+) ms", l)
    if m:
        cases.append((int(m.group(1)), m.group(2).strip(), float(m.group(3)), int(m.group(4))))
for t_end, name, tf, ms in cases:
    t0 = t_end - ms / 1000.0
    s = [(c, w) for (t, c, w) in smi if t0 + 0.9 <= t <= t_end - 0.9]
    if s:
        print(f"{name:34s} {tf:7.1f} TFLOP/s   sclk {sum(c for c, _ in s) / len(s):5.0f} MHz   package power {sum(w for _, w in s) / len(s):5.0f} W   ({len(s)} samples)")
    else:
        print(f"{name:34s} {tf:7.1f} TFLOP/s   (no rocm-smi sample inside)")
