"""CPU (needs hipcc only): the counted s_waitcnt of the clip-tile GEMM's loader waves assumes that hipcc emits exactly one
global-store instruction per source-level 16-byte store of the epilogue (csrc/tamf_gemm_clip.h, SX).  tools/check_clip_stores.py
disassembles the library for gfx950 and checks the store count of every clip_gemm_kernel instantiation against the source's
formula, the immediates of the counted waits, and that no other store flavour or spill appears."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_clip_gemm_store_count_matches_the_counted_waits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_clip_stores.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "clip_gemm_kernel instantiations match" in r.stdout
    assert "gemm_deep_kernel instantiations match" in r.stdout  # (csrc/tamf_gemm_deep.h: one LDS-DMA instruction per counted request)
    assert "no register spilling beyond" in r.stdout  # (attention + clip-GEMM kernels: private segment <= 32 bytes per lane)
