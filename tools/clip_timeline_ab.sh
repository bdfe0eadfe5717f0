#!/bin/bash
# GPU: per-wave K-tile interval stamps of the clip GEMMs, standalone (GEMM hook) and as they run inside the hipGraph step.
# Needs two debug builds next to the product library (build them HERE, on the CPU box, before gpurun - they travel with the tree):
#   L=oakink2-tamf_amd/oakink2_tamf_amd/lib
#   TAMF_HIPCC_FLAGS="-DTAMF_TIMELINE -DTAMF_TIMELINE_NI=2" tools/ab_build.sh HEAD && mv $L/libtamf_hip_A.so $L/libtamf_hip_T2.so
#   TAMF_HIPCC_FLAGS="-DTAMF_TIMELINE -DTAMF_TIMELINE_NI=4" tools/ab_build.sh HEAD && mv $L/libtamf_hip_A.so $L/libtamf_hip_T4.so
# (TAMF_TIMELINE_NI = only the launches with that many column tiles per wave write stamps, so that inside the step the last
#  stamping launch is the FFN2 (2) or the FFN1 (4) of the last layer)
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
echo "=== standalone FFN2"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_T2.so python tools/clip_timeline.py f16x3 3 512 2048 -1 2>&1 | grep -v amdgpu
echo "=== in situ FFN2";   TAMF_LIB_OVERRIDE=$L/libtamf_hip_T2.so python tools/clip_timeline_insitu.py f16x3 64 2>&1 | grep -v amdgpu
echo "=== standalone FFN1"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_T4.so python tools/clip_timeline.py f16x3 0 2048 512 -1 2>&1 | grep -v amdgpu
echo "=== in situ FFN1";   TAMF_LIB_OVERRIDE=$L/libtamf_hip_T4.so python tools/clip_timeline_insitu.py f16x3 64 2>&1 | grep -v amdgpu
