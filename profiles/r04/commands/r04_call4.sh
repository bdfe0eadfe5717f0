#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/energy_model.sh 5
python tools/two_streams.py bf16 64 200 2>&1 | grep "ms/step" > gpurun_out/r04/two_streams_bf16.txt
python tools/two_streams.py bf16 128 200 2>&1 | grep "ms/step" >> gpurun_out/r04/two_streams_bf16.txt
cat gpurun_out/r04/two_streams_bf16.txt
