#!/bin/bash
# GPU: everything the energy model of DESIGN.md section 6 is built from -> gpurun_out/r04/energy_*.txt
#   tools/energy_model.sh [seconds per case]
secs=${1:-5}
out=gpurun_out/r04; mkdir -p $out
rm -f /tmp/tamf_stop_sampler
python3 tools/power_sampler.py $out/energy_power_samples.txt /tmp/tamf_stop_sampler &
spid=$!
sleep 2
tools/micro/energy $secs > $out/energy_cases.txt 2>&1
python3 tools/energy_loops.py 8 64 196 >> $out/energy_cases.txt 2>&1
touch /tmp/tamf_stop_sampler
wait $spid
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" > $out/energy_powercap.txt
grep CASE $out/energy_cases.txt | cut -c1-120
head -3 $out/energy_power_samples.txt; wc -l $out/energy_power_samples.txt
