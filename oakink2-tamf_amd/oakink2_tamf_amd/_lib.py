"""Locate / build / load libtamf_hip.so (the C-ABI of include/tamf_hip.h)."""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(_HERE, "..", "..", "include"))
LIB_PATH = os.path.join(_HERE, "lib", "libtamf_hip.so")
SOURCES = ["tamf_hip.hip", "tamf_device.h", "tamf_gemm.h", "tamf_gemm_clip.h", "tamf_attn.h", "tamf_misc.h", "tamf_geom.h"]

_lock = threading.Lock()
_lib = None


class TamfBuildError(RuntimeError):
    pass


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(INCLUDE, "tamf_hip.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared; cross-compiles without a GPU.  Returns the .so path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise TamfBuildError("hipcc not found: cannot build libtamf_hip.so")
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    # one builder at a time (the ranks of a multi-GPU launch all import this module): the others wait and then find it fresh
    import fcntl

    lock = open(LIB_PATH + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():
            return LIB_PATH
        return _build_locked(hipcc, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc: str, verbose: bool) -> str:
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", tmp, os.path.join(CSRC, "tamf_hip.hip")] + os.environ.get("TAMF_HIPCC_FLAGS", "").split()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise TamfBuildError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library (building it first if the in-tree .so is missing or older than its sources).
    Raises - never falls back to a CPU path."""
    global _lib
    with _lock:
        if _lib is None:
            build()
            # torch ships its own libamdhip64; import it first so that the library binds to the HIP runtime
            # instance torch uses (one runtime per process: shared device memory, streams, contexts).
            import torch  # noqa: F401

            _lib = ctypes.CDLL(LIB_PATH)
        return _lib


def load_from(path: str) -> ctypes.CDLL:
    """Bind the process to another build of the library, given explicitly by the caller, before the first load().  For the
    measurement scripts under tools/ (two builds alternating on one box, debug builds with timeline stamps): the product
    path never calls this and reads no environment variable."""
    global _lib
    with _lock:
        if _lib is not None:
            raise RuntimeError("libtamf_hip is already loaded in this process")
        import torch  # noqa: F401

        _lib = ctypes.CDLL(path)
        return _lib


EXPORTS = [
    "tamf_ctx_create", "tamf_ctx_destroy", "tamf_last_error", "tamf_load_weight", "tamf_finalize_weights",
    "tamf_set_schedule", "tamf_set_cond", "tamf_denoise", "tamf_ddpm_step", "tamf_sample_loop", "tamf_refine",
    "tamf_pose_decode", "tamf_h2o_dist", "tamf_contact_min_dist", "tamf_mesh_contains", "tamf_transform_points", "tamf_vertex_normals", "tamf_get_status_flags", "tamf_step_kernel_count", "tamf_loop_stats", "tamf_step_profile", "tamf_test_gemm", "tamf_test_gemm_ln", "tamf_test_attention", "tamf_test_philox", "tamf_bench_gemm", "tamf_bench_attention", "tamf_bench_mfma_rate", "tamf_set_gemm_tuning",
]
