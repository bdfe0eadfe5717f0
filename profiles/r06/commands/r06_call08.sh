#!/bin/bash
# config 3's shard (32 clips per GPU) as two concurrent half batches of 16 on two streams (tools/two_streams.py): do they beat one batch of 32?
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
{
for p in f16x3 bf16; do
  timeout 300 python tools/two_streams.py $p 32 200 2>&1 | grep "ms/step"
  timeout 300 python tools/two_streams.py $p 32 200 0x400fffff 2>&1 | grep "ms/step"
  timeout 300 python tools/two_streams.py $p 64 200 2>&1 | grep "ms/step"
done
} > gpurun_out/r06/two_streams_b32_c08.txt 2>&1
cat gpurun_out/r06/two_streams_b32_c08.txt
