"""tools only: bind this process to the library build named by TAMF_LIB_OVERRIDE (A/B runs of two builds on one box,
-DTAMF_TIMELINE debug builds; tools/ab_build.sh).  Import it before anything touches oakink2_tamf_amd.hip_backend.  The product
loader (oakink2_tamf_amd/_lib.py) reads no environment variable; this helper passes the path to it explicitly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oakink2-tamf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

_path = os.environ.get("TAMF_LIB_OVERRIDE")
if _path:
    from oakink2_tamf_amd import _lib

    _lib.load_from(_path)  # (stands for the product AND the hooks library: tools/ab_build.sh compiles with -DTAMF_TEST_HOOKS)

# measurement tools work on libtamf_hip_hooks.so (tamf_set_gemm_tuning, tamf_bench_*, tamf_debug_timeline live there only)
from oakink2_tamf_amd import hip_backend as _hb  # noqa: E402

_hb.use_test_hooks(True)
