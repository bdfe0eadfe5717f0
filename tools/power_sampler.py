"""GPU box: sample package power and shader clock about 4 x per second until the file given as argv[2] appears:
   python tools/power_sampler.py <out.txt> <stop-file>
Lines: "<epoch seconds> <watts card0> <sclk MHz card0> <watts card1> <sclk card1> ..." - EVERY amdgpu card of the box (hwmon
power1_average / power1_input, pp_dpm_sclk): the sysfs card order is not the HIP device order and other cards may belong to other jobs,
so tools/energy_model.py picks the card whose power follows this job's cases.  Falls back to parsing rocm-smi (about 1 Hz, GPU[0]).
Touches no GPU API (no HIP initialisation in this process)."""
import glob
import os
import re
import subprocess
import sys
import time

out, stop = sys.argv[1], sys.argv[2]


def sysfs_cards():
    found = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        pw = glob.glob(card + "/hwmon/hwmon*/power1_average") + glob.glob(card + "/hwmon/hwmon*/power1_input")
        sc = card + "/pp_dpm_sclk"
        if pw and os.path.exists(sc):
            try:
                int(open(pw[0]).read())
                open(sc).read()
                found.append((pw[0], sc))
            except (OSError, ValueError):
                continue
    return found


cards = sysfs_cards()
with open(out, "w") as f:
    f.write(f"# source: {'sysfs: ' + ' '.join(c[0] for c in cards) if cards else 'rocm-smi GPU[0]'}\n")
    # PCI address of every sampled card, in column order: bench.py matches its own device (torch's pci_domain/bus/device ids) against it
    f.write("# pci: " + " ".join(os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(os.path.dirname(c[0]))))) for c in cards) + "\n")
    t_end = time.time() + 1800.0  # (never outlives its parent by more than this)
    while not os.path.exists(stop) and time.time() < t_end:
        t = time.time()
        vals = []
        try:
            if cards:
                for pw, sc in cards:
                    watts = int(open(pw).read()) / 1e6
                    m = re.search(r"(\d+)Mhz \*", open(sc).read())
                    vals += [f"{watts:.1f}", f"{float(m.group(1)) if m else float('nan'):.0f}"]
                time.sleep(0.2)
            else:
                txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
                mp = re.search(r"GPU\[0\].*?Power.*?:\s*([\d.]+)", txt)
                ms = re.search(r"GPU\[0\].*?sclk.*?\((\d+)Mhz\)", txt)
                vals = [f"{float(mp.group(1)) if mp else float('nan'):.1f}", f"{float(ms.group(1)) if ms else float('nan'):.0f}"]
        except Exception as e:  # noqa: BLE001
            f.write(f"# {e}\n")
            time.sleep(0.5)
            continue
        f.write(f"{t:.3f} " + " ".join(vals) + "\n")
        f.flush()
