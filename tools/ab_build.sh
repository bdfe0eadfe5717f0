#!/bin/bash
# Build the csrc of another git revision into lib/libtamf_hip_A.so so that two builds can be timed on the same GPU box:
#   tools/ab_build.sh HEAD        then on the box:  TAMF_LIB_OVERRIDE=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so python tools/kbench.py ...
set -e
ref=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" oakink2-tamf_amd/csrc include | tar -x -C "$tmp"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -o "$root/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so" "$tmp/oakink2-tamf_amd/csrc/tamf_hip.hip" $TAMF_HIPCC_FLAGS
rm -rf "$tmp"
echo "built libtamf_hip_A.so from $ref"
