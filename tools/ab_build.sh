#!/bin/bash
# Build another copy of the library so that two builds can be timed on the same GPU box:
#   tools/ab_build.sh <git ref> [name]      csrc of that revision            -> lib/libtamf_hip_<name>.so   (name default: A)
#   tools/ab_build.sh WORKTREE [name]       the working tree's csrc (e.g. with TAMF_HIPCC_FLAGS=-DTAMF_BENCH for the ablation bits)
# then on the box:  TAMF_LIB_OVERRIDE=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_<name>.so python tools/kbench.py ...
set -e
ref=${1:-HEAD}
name=${2:-A}
root=$(cd "$(dirname "$0")/.." && pwd)
out="$root/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_$name.so"
if [ "$ref" = WORKTREE ]; then
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-function -DTAMF_TEST_HOOKS -o "$out" "$root/oakink2-tamf_amd/csrc/tamf_hip.hip" $TAMF_HIPCC_FLAGS
else
  tmp=$(mktemp -d)
  git -C "$root" archive "$ref" oakink2-tamf_amd/csrc include | tar -x -C "$tmp"
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-function -DTAMF_TEST_HOOKS -o "$out" "$tmp/oakink2-tamf_amd/csrc/tamf_hip.hip" $TAMF_HIPCC_FLAGS
  rm -rf "$tmp"
fi
echo "built $(basename "$out") from $ref ${TAMF_HIPCC_FLAGS}"
