"""Build-time check of the counted waits of the clip-tile GEMM (csrc/tamf_gemm_clip.h).

The loader waves of clip_gemm_kernel guard LDS reuse with `s_waitcnt vmcnt(PH + SX)`: SX is the number of global-store
instructions the epilogue of a tile issues behind the LDS-DMA requests of the next K tile.  That number is a compile-time formula
(MSUBX * NCHUNK * CHUNK_STORES, or the V^T form), so the kernel is only correct while hipcc emits exactly ONE store instruction per
source-level 16-byte store - no merging, splitting or elision.  This script disassembles the library for gfx950 and checks, for every
clip_gemm_kernel instantiation:
  * the static number of global_store instructions equals what the source's formula implies for all its code variants
    (three activation variants x the X and Y wave roles; V^T form: one variant per role);
  * every counted wait immediate as large as the kernel's own (s_waitcnt vmcnt(N), N >= min(PH - 1, SX); the compiler's small
    counted waits for the column constants lie below) is one of {PH, PH - 1, PH + SX, PH - 1 + SX, SX};
  * no buffer_store / flat_store / scratch instruction appears (a spill or another store flavour would not be counted by the formula).
Run:  python tools/check_clip_stores.py [existing .s file]      (tests/test_isa_clip_waits.py runs it on CPU; needs hipcc only)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oakink2-tamf_amd"))
SRC = os.path.join(ROOT, "oakink2-tamf_amd", "csrc", "tamf_hip.hip")


def disassemble():
    hipcc = "/opt/rocm/bin/hipcc"
    out = os.path.join(tempfile.mkdtemp(prefix="tamf_isa_"), "tamf.s")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, SRC],
                   check=True, capture_output=True)
    return out


def main():
    from oakink2_tamf_amd import _isa_check  # (the logic lives in the package: _lib.build runs it on every build)

    path = sys.argv[1] if len(sys.argv) > 1 else disassemble()
    try:
        n = _isa_check.check(path, log=print)
    except _isa_check.IsaMismatch as e:
        raise SystemExit(str(e))
    print(f"all {n} clip_gemm_kernel instantiations match")
    try:
        nd = _isa_check.check_deep(path, log=print)
    except _isa_check.IsaMismatch as e:
        raise SystemExit(str(e))
    print(f"all {nd} gemm_deep_kernel instantiations match")
    try:
        k = _isa_check.check_scratch(path, log=print)
    except _isa_check.IsaMismatch as e:
        raise SystemExit(str(e))
    print(f"no register spilling beyond {_isa_check.SCRATCH_LIMIT} bytes per lane in {k} attention / clip-GEMM kernels")


if __name__ == "__main__":
    main()
