#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
tools/micro/issue_overlap > gpurun_out/r04/issue_overlap_c31.txt 2>&1
cat gpurun_out/r04/issue_overlap_c31.txt
