"""Locate / build / load libtamf_hip.so (the C-ABI of include/tamf_hip.h) and, for tests/ and tools/ only, libtamf_hip_hooks.so
(the same sources with -DTAMF_TEST_HOOKS: + the entry points of include/tamf_hip_test.h)."""
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
HOOKS_PATH = os.path.join(_HERE, "lib", "libtamf_hip_hooks.so")  # test / measurement build (never loaded by the product path)
HEADERS = ("tamf_hip.h", "tamf_hip_test.h")
SOURCES = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] if os.path.isdir(CSRC) else []

_lock = threading.Lock()
_lib = None
_hooks = None


class TamfBuildError(RuntimeError):
    pass


STAMP_PATH = LIB_PATH + ".src.sha256"


def source_digest() -> str:
    """sha256 over the kernel sources and the C header (names + contents, sorted): what the built library is stamped with"""
    import hashlib

    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(INCLUDE, h_) for h_ in HEADERS]:
        if os.path.exists(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    """The in-tree library is current when the digest of the sources it was built from (written beside it by build()) equals the
    digest of the sources in the tree.  Content, not mtimes: copying the tree to a GPU box resets every mtime, and eight ranks of a
    multi-GPU launch must not queue behind a needless 80-second rebuild inside somebody's timed window."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HOOKS_PATH):
        return True
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip() != source_digest()
    except OSError:
        # a library without a stamp (built by an older tree): fall back to the mtime rule once; build() writes the stamp
        t = os.path.getmtime(LIB_PATH)
        deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(INCLUDE, h_) for h_ in HEADERS]
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


def _compile(hipcc: str, workdir: str, extra, name: str = "libtamf_hip.so"):
    """One hipcc run in `workdir` with -save-temps=obj: the library AND the device assembly of the same compile.
    The product build takes no flags from the environment (tools/ab_build.sh builds the -DTAMF_BENCH / -DTAMF_TIMELINE copies
    for measurements under other file names)."""
    os.makedirs(workdir, exist_ok=True)
    out = os.path.join(workdir, name)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-unused-function", "-save-temps=obj",
           "-o", out, os.path.join(CSRC, "tamf_hip.hip")] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=workdir)
    if res.returncode != 0:
        raise TamfBuildError("hipcc failed:\n" + res.stdout + res.stderr)
    asm = [f for f in os.listdir(workdir) if f.endswith("gfx950.s")]
    return out, (os.path.join(workdir, asm[0]) if asm else None)


def _build_locked(hipcc: str, verbose: bool) -> str:
    """Both libraries from the same sources, compiled side by side (two hipcc processes): libtamf_hip.so - the drop-in surface only -
    and libtamf_hip_hooks.so (-DTAMF_TEST_HOOKS: + include/tamf_hip_test.h).  The ISA checks run on the product's assembly; the hooks
    build instantiates the same kernel templates."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    from . import _isa_check

    with tempfile.TemporaryDirectory(prefix="tamf_build_", dir=os.path.dirname(LIB_PATH)) as wd:
        wp, wh = os.path.join(wd, "product"), os.path.join(wd, "hooks")
        with ThreadPoolExecutor(max_workers=2) as ex:
            fh = ex.submit(_compile, hipcc, wh, ["-DTAMF_TEST_HOOKS"], "libtamf_hip_hooks.so")
            out, asm = _compile(hipcc, wp, [])
            out_h, asm_h = fh.result()
        # the clip-tile GEMM's counted waits assume one global_store per source-level store: verified on the assembly of THIS
        # compile (whatever hipcc the site has); on a mismatch both libraries are rebuilt with every counted wait as vmcnt(0)
        safe = False
        try:
            n = nd = 0
            for a in (asm, asm_h):
                if a is None:
                    raise _isa_check.IsaMismatch("hipcc left no device assembly to check")
                n = _isa_check.check(a)
                nd = _isa_check.check_deep(a)
            if verbose:
                print(f"ISA check: all {n} clip_gemm_kernel and {nd} gemm_deep_kernel instantiations match the counted waits (both builds)")
            try:  # register spilling in a hot kernel is a performance bug, not a correctness one: reported, never fatal
                _isa_check.check_scratch(asm)
            except _isa_check.IsaMismatch as e:
                import warnings

                warnings.warn(f"libtamf_hip: {e}")
        except (_isa_check.IsaMismatch, IndexError, KeyError, ValueError) as e:  # (a newer hipcc may also break the checker's parsing)
            import shutil as _sh
            import warnings

            warnings.warn(f"libtamf_hip: {e}\nrebuilding with -DTAMF_CLIP_SAFE_WAIT (counted waits -> vmcnt(0))")
            _sh.rmtree(wp, ignore_errors=True)
            _sh.rmtree(wh, ignore_errors=True)
            with ThreadPoolExecutor(max_workers=2) as ex:
                fh = ex.submit(_compile, hipcc, wh, ["-DTAMF_TEST_HOOKS", "-DTAMF_CLIP_SAFE_WAIT"], "libtamf_hip_hooks.so")
                out, _ = _compile(hipcc, wp, ["-DTAMF_CLIP_SAFE_WAIT"])
                out_h, _ = fh.result()
            safe = True
        digest = source_digest()  # (of the sources as they are now: an edit during the compile makes the stamp differ next time)
        os.replace(out_h, HOOKS_PATH)
        os.replace(out, LIB_PATH)
        with open(STAMP_PATH + ".tmp", "w") as f:
            f.write(digest + "\n")
        os.replace(STAMP_PATH + ".tmp", STAMP_PATH)
    if verbose:
        print("built", LIB_PATH, "and", os.path.basename(HOOKS_PATH), "(safe waits)" if safe else "")
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


def load_hooks() -> ctypes.CDLL:
    """The -DTAMF_TEST_HOOKS build (include/tamf_hip_test.h): tests/, tools/ and bench.py's register-only MFMA probe.  A separate
    library object with its own process-global state (guard-band mode, kernel-selection word): contexts created through it are
    independent of contexts of libtamf_hip.so."""
    global _hooks
    with _lock:
        if _hooks is None:
            build()
            import torch  # noqa: F401

            _hooks = ctypes.CDLL(HOOKS_PATH)
        return _hooks


def load_from(path: str) -> ctypes.CDLL:
    """Bind the process to another build of the library, given explicitly by the caller, before the first load().  For the
    measurement scripts under tools/ (two builds alternating on one box, debug builds with timeline stamps; tools/ab_build.sh
    compiles them with -DTAMF_TEST_HOOKS): it stands for BOTH libraries.  The product path never calls this and reads no
    environment variable."""
    global _lib, _hooks
    with _lock:
        if _lib is not None or _hooks is not None:
            raise RuntimeError("libtamf_hip is already loaded in this process")
        import torch  # noqa: F401

        _lib = _hooks = ctypes.CDLL(path)
        return _lib


EXPORTS = [  # include/tamf_hip.h: what libtamf_hip.so exports, nothing else
    "tamf_ctx_create", "tamf_ctx_resize", "tamf_ctx_destroy", "tamf_last_error", "tamf_load_weight", "tamf_finalize_weights",
    "tamf_set_schedule", "tamf_set_timestep_map", "tamf_set_cond", "tamf_set_cond_ragged", "tamf_denoise", "tamf_ddpm_step", "tamf_sample_loop", "tamf_refine",
    "tamf_pose_decode", "tamf_h2o_dist", "tamf_contact_min_dist", "tamf_mesh_contains", "tamf_transform_points", "tamf_vertex_normals",
    "tamf_get_status_flags", "tamf_step_kernel_count", "tamf_loop_stats", "tamf_step_profile", "tamf_refine_profile",
]
HOOK_EXPORTS = [  # include/tamf_hip_test.h: additionally in libtamf_hip_hooks.so
    "tamf_test_gemm", "tamf_test_gemm_resid", "tamf_test_attention", "tamf_test_philox", "tamf_test_set_guard_bytes", "tamf_test_check_guards",
    "tamf_test_poke", "tamf_test_fail_alloc_after", "tamf_bench_gemm", "tamf_bench_attention", "tamf_bench_mfma_rate", "tamf_set_gemm_tuning",
]
