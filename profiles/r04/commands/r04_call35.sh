#!/bin/bash
# the default bench line against the traffic files of this commit (traffic_stale: false)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python3 bench.py > gpurun_out/r04/bench_default_final.log 2>&1
tail -n 1 gpurun_out/r04/bench_default_final.log | cut -c1-300
