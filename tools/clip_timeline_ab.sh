L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
echo "=== standalone FFN2"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_T2.so python tools/clip_timeline.py f16x3 3 512 2048 -1 2>&1 | grep -v amdgpu
echo "=== in situ FFN2";   TAMF_LIB_OVERRIDE=$L/libtamf_hip_T2.so python tools/clip_timeline_insitu.py f16x3 64 2>&1 | grep -v amdgpu
echo "=== standalone FFN1"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_T4.so python tools/clip_timeline.py f16x3 0 2048 512 -1 2>&1 | grep -v amdgpu
echo "=== in situ FFN1";   TAMF_LIB_OVERRIDE=$L/libtamf_hip_T4.so python tools/clip_timeline_insitu.py f16x3 64 2>&1 | grep -v amdgpu
