"""ctypes binding of include/tamf_hip.h.  PyTorch is used only for device memory and streams."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_uint8, c_uint64, c_void_p
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

PRECISIONS = {"f32": 0, "bf16": 1, "bf16x3": 2, "f16x3": 3}
KINDS = {"G": 0, "R": 1}


class TamfError(RuntimeError):
    pass


class TamfRangeError(TamfError):
    """A non-finite weight (at load) or an activation beyond +-65504 (status flag) does not fit the split-fp16 operand format of "f16x3"."""


STATUS_F16_RANGE = 1
STATUS_F16_WEIGHT_RANGE = 2
DEFAULT_PRECISION = "f16x3"  # fp32-equivalent at the stated 1e-5 tolerance (22 significand bits for values >= 2^-3 of fp16's normal range - weights
# are pre-scaled into it - and an absolute 3e-8 operand error below |v| = 0.12 for activations), range-guarded; DESIGN.md section 2


class _Arch(ctypes.Structure):
    _fields_ = [
        ("input_dim", c_int32),
        ("obj_input_dim", c_int32),
        ("hand_shape_dim", c_int32),
        ("obj_embed_dim", c_int32),
        ("latent_dim", c_int32),
        ("ff_size", c_int32),
        ("num_layers", c_int32),
        ("num_heads", c_int32),
        ("clip_dim", c_int32),
        ("h2o_dim", c_int32),
        ("kind", c_int32),
    ]


_bound = set()
_use_hooks = False


def _bind(L: ctypes.CDLL) -> ctypes.CDLL:
    """argtypes / restypes of whatever the library object exports (once per object)"""
    if id(L) in _bound:
        return L
    L.tamf_last_error.restype = c_char_p
    L.tamf_last_error.argtypes = [c_void_p]
    L.tamf_ctx_create.argtypes = [POINTER(_Arch), c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]
    L.tamf_ctx_destroy.argtypes = [c_void_p]
    L.tamf_ctx_destroy.restype = None
    L.tamf_load_weight.argtypes = [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int32]
    L.tamf_finalize_weights.argtypes = [c_void_p, c_int32, c_void_p]
    L.tamf_set_schedule.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    if hasattr(L, "tamf_set_timestep_map"):  # (round 6; absent from older A/B builds)
        L.tamf_set_timestep_map.argtypes = [c_void_p, c_int32, c_void_p]
    L.tamf_set_cond.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    if hasattr(L, "tamf_set_cond_ragged"):  # (absent only from older A/B builds loaded by tools/ through _lib.load_from)
        L.tamf_ctx_resize.argtypes = [c_void_p, c_int32, c_int32]
        L.tamf_set_cond_ragged.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.tamf_denoise.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.tamf_ddpm_step.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p]
    L.tamf_sample_loop.argtypes = [c_void_p, c_void_p, c_uint64, c_int64, c_void_p, c_void_p, c_int32, c_void_p]
    L.tamf_refine.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    if hasattr(L, "tamf_get_status_flags"):  # (absent only from older A/B builds loaded by tools/ through _lib.load_from)
        L.tamf_get_status_flags.argtypes = [c_void_p, POINTER(ctypes.c_uint32), c_int32, c_void_p]
    L.tamf_step_kernel_count.argtypes = [c_void_p]
    L.tamf_loop_stats.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32)]
    L.tamf_step_profile.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    if hasattr(L, "tamf_refine_profile"):
        L.tamf_refine_profile.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    # include/tamf_hip_test.h: only libtamf_hip_hooks.so (and the A/B builds of tools/ab_build.sh) has these
    if hasattr(L, "tamf_test_gemm"):
        L.tamf_test_gemm.argtypes = [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
        L.tamf_test_attention.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
        L.tamf_test_philox.argtypes = [c_uint64, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]
        L.tamf_set_gemm_tuning.argtypes = [c_int32]
    if hasattr(L, "tamf_test_gemm_resid"):
        L.tamf_test_gemm_resid.argtypes = [c_int32, c_int32, c_int32, c_int32] + [c_void_p] * 8
    if hasattr(L, "tamf_test_set_guard_bytes"):
        L.tamf_test_set_guard_bytes.argtypes = [c_int64]
        L.tamf_test_check_guards.argtypes = [c_void_p, POINTER(c_int32)]
        L.tamf_test_poke.argtypes = [c_void_p, c_int32, c_int64, c_int32]
    if hasattr(L, "tamf_test_fail_alloc_after"):
        L.tamf_test_fail_alloc_after.argtypes = [c_int32]
    if hasattr(L, "tamf_bench_mfma_rate"):
        L.tamf_bench_mfma_rate.argtypes = [c_int32, c_int32, POINTER(ctypes.c_float), POINTER(ctypes.c_float), c_void_p]
    _bound.add(id(L))
    return L


def hooks() -> ctypes.CDLL:
    """libtamf_hip_hooks.so: the -DTAMF_TEST_HOOKS build of the same sources (include/tamf_hip_test.h) - tests/, tools/, bench.py's
    register-only MFMA probe.  Never the product path."""
    return _bind(_lib.load_hooks())


def use_test_hooks(on: bool = True) -> None:
    """tests/ and tools/ only: contexts created from now on (and `lib()`) go through libtamf_hip_hooks.so, whose process-global
    switches - guard bands, allocation-failure injection, the kernel-selection word of tamf_set_gemm_tuning - then apply to them.
    A context keeps the library it was created with."""
    global _use_hooks
    _use_hooks = bool(on)


def lib() -> ctypes.CDLL:
    """the library new contexts are created with: libtamf_hip.so - the drop-in surface, nothing else - unless a test or tool has
    switched to the hooks build (use_test_hooks)"""
    return hooks() if _use_hooks else _bind(_lib.load())


def set_guard_bytes(n: int) -> None:
    """Test hook: contexts created from now on THROUGH THE HOOKS LIBRARY (use_test_hooks) pad every device allocation with n guard
    bytes at both ends (0 = off)."""
    _check(hooks().tamf_test_set_guard_bytes(int(n)), None, hooks())


def hand_side_code(hs) -> int:
    """'rh' -> 0, 'lh' -> 1 in any of the encodings a batch may carry: the reference's strings (also as bytes or numpy
    str_ from an .npz), or the 0 / 1 flags of the synthetic conditioning.  Anything else raises ValueError exactly as
    HandsideProcess does (interaction_segment_mdm.py:284)."""
    if isinstance(hs, bytes):
        hs = hs.decode()
    if isinstance(hs, str):
        if hs in ("rh", "lh"):
            return 0 if hs == "rh" else 1
        raise ValueError(f"unexpected hand_side: {hs}")
    try:
        v = int(hs)
    except (TypeError, ValueError):
        raise ValueError(f"unexpected hand_side: {hs}") from None
    if v in (0, 1):
        return v
    raise ValueError(f"unexpected hand_side: {hs}")


def _stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _check(rc: int, ctx=None, L=None):
    if rc != 0:
        msg = (L or lib()).tamf_last_error(ctx)
        raise (TamfRangeError if rc == -6 else TamfError)(f"libtamf_hip error {rc}: {msg.decode() if msg else '?'}")


def _dev_f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise TamfError("no MI355X/HIP device visible: the MF-MDM HIP path has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise TamfError(f"HIP path needs a cuda (ROCm) device, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class TamfContext:
    """One library context = one (model, device, precision, max batch, max frames)."""

    def __init__(self, arch: Mapping[str, int], max_batch: int, max_frames: int, precision: str = DEFAULT_PRECISION,
                 device=None, kind: str = "G", range_check: bool = False):
        """range_check=True: `denoise`, `sample_loop` and `refine` read this context's status word after the call (one stream
        synchronisation) and raise TamfRangeError when an activation left the fp16 range (f16x3 only).  The drop-in modules do
        their own check - with an f32 fallback - and leave this off; a raw context has no fallback."""
        self.device = require_gpu(device)
        self._L = lib()  # the library this context lives in: libtamf_hip.so unless a test / tool switched to the hooks build
        self.precision = precision
        self.range_check = bool(range_check) and precision == "f16x3"
        self.kind = kind
        a = _Arch(
            input_dim=int(arch.get("input_dim", 99)),
            obj_input_dim=int(arch.get("obj_input_dim", 9)),
            hand_shape_dim=int(arch.get("hand_shape_dim", 10)),
            obj_embed_dim=int(arch.get("obj_embed_dim", 768)),
            latent_dim=int(arch.get("latent_dim", 256)),
            ff_size=int(arch.get("ff_size", 1024)),
            num_layers=int(arch.get("num_layers", 8)),
            num_heads=int(arch.get("num_heads", 4)),
            clip_dim=int(arch.get("clip_dim", 512)),
            h2o_dim=int(arch.get("h2o_dim", 778)),
            kind=KINDS[kind],
        )
        self.input_dim = a.input_dim
        self.h2o_dim = a.h2o_dim
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)
        self._h = c_void_p()
        self.B = self.T = 0
        self.n_steps = 0
        self.max_timesteps = 0
        self._keep = []
        with torch.cuda.device(self.device):
            _check(self._L.tamf_ctx_create(ctypes.byref(a), max_batch, max_frames, PRECISIONS[precision],
                                           self.device.index, ctypes.byref(self._h)), None, self._L)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.tamf_ctx_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- weights ------------------------------------------------------------------------------
    def load_state_dict(self, sd: Mapping[str, torch.Tensor], max_timesteps: int = 5000, strict_weight_range: bool = False):
        """max_timesteps = rows of the timestep-embedding table; the default covers every t the reference's
        `pe[timesteps]` lookup accepts (sequence_pos_encoder.pe has 5000 rows).  strict_weight_range (f16x3): raise
        TamfRangeError when the library reports STATUS_F16_WEIGHT_RANGE - a tensor whose per-tensor scale is dominated by an
        outlier, so that its ordinary weights keep fewer than 22 significand bits (the modules then fall back to f32)."""
        L = self._L
        self.max_timesteps = int(max_timesteps)
        for name, t in sd.items():
            if not isinstance(t, torch.Tensor) or name.startswith("clip_model."):
                continue
            h = t.detach().to("cpu", torch.float32).contiguous()
            shape = (c_int64 * max(h.dim(), 1))(*(list(h.shape) or [1]))
            _check(L.tamf_load_weight(self._h, name.encode(), c_void_p(h.data_ptr()), shape, max(h.dim(), 1)), self._h, self._L)
        with torch.cuda.device(self.device):
            _check(L.tamf_finalize_weights(self._h, int(max_timesteps), c_void_p(_stream_ptr(self.device))), self._h, self._L)
        if strict_weight_range and self.precision == "f16x3" and (self.status_flags(clear=False) & STATUS_F16_WEIGHT_RANGE):
            note = L.tamf_last_error(self._h)
            raise TamfRangeError(note.decode() if note else "f16x3: a weight tensor's dynamic range exceeds the split-fp16 format")

    def set_schedule(self, coef1: np.ndarray, coef2: np.ndarray, log_variance_clipped: np.ndarray, timestep_map=None):
        """The float64 posterior tables of the sampler; `timestep_map` (respaced samplers, respace.py:60-119): step i of the loop
        evaluates the denoiser at timestep_map[i] of the base process (None = identity)."""
        c1 = np.ascontiguousarray(coef1, dtype=np.float64)
        c2 = np.ascontiguousarray(coef2, dtype=np.float64)
        lv = np.ascontiguousarray(log_variance_clipped, dtype=np.float64)
        assert c1.shape == c2.shape == lv.shape and c1.ndim == 1
        self.n_steps = int(c1.shape[0])
        _check(self._L.tamf_set_schedule(self._h, self.n_steps, c1.ctypes.data_as(c_void_p), c2.ctypes.data_as(c_void_p),
                                       lv.ctypes.data_as(c_void_p)), self._h)
        if timestep_map is not None:
            tm = np.ascontiguousarray(timestep_map, dtype=np.int32)
            assert tm.shape == (self.n_steps,), (tm.shape, self.n_steps)
            if not np.array_equal(tm, np.arange(self.n_steps, dtype=np.int32)):
                with torch.cuda.device(self.device):
                    _check(self._L.tamf_set_timestep_map(self._h, self.n_steps, tm.ctypes.data_as(c_void_p)), self._h, self._L)

    # -- conditioning -------------------------------------------------------------------------
    def set_cond(self, text_embedding: Optional[torch.Tensor], hand_side: Sequence, shape: torch.Tensor,
                 obj_embedding: torch.Tensor, obj_traj: torch.Tensor, obj_num: Optional[Sequence[int]] = None):
        """obj_num: per-clip object counts (each in [1, nobj]) - the object means then run over each clip's own objects, as when
        the reference's launchers call the model one clip at a time; None: over all nobj rows of the (zero-padded) batch, which is
        what the reference's forward computes on the batch it is handed."""
        side = [hand_side_code(hs) for hs in hand_side]
        B, nobj, T, _ = obj_traj.shape
        dev = self.device
        te = _dev_f32(text_embedding, dev) if text_embedding is not None else None
        sh, oe, ot = _dev_f32(shape, dev), _dev_f32(obj_embedding, dev), _dev_f32(obj_traj, dev)
        assert sh.shape[0] == B and sh.shape[1] == T and oe.shape[0] == B and oe.shape[1] == nobj
        side_np = np.asarray(side, dtype=np.uint8)
        assert side_np.shape[0] == B
        num_np = None
        if obj_num is not None:
            num_np = np.ascontiguousarray(torch.as_tensor(obj_num).cpu().numpy() if isinstance(obj_num, torch.Tensor) else obj_num, dtype=np.int32)
            if num_np.shape != (B,):
                raise ValueError(f"obj_num must hold one count per clip: shape {num_np.shape} for B = {B}")
        if num_np is None and not hasattr(self._L, "tamf_set_cond_ragged"):  # (an older A/B build loaded through _lib.load_from)
            with torch.cuda.device(dev):
                _check(self._L.tamf_set_cond(self._h, B, T, nobj, c_void_p(te.data_ptr() if te is not None else 0), side_np.ctypes.data_as(c_void_p),
                                           c_void_p(sh.data_ptr()), c_void_p(oe.data_ptr()), c_void_p(ot.data_ptr()), c_void_p(_stream_ptr(dev))), self._h)
            self._keep = [te, sh, oe, ot]
            self.B, self.T = int(B), int(T)
            return
        with torch.cuda.device(dev):
            _check(self._L.tamf_set_cond_ragged(self._h, B, T, nobj, num_np.ctypes.data_as(c_void_p) if num_np is not None else c_void_p(0),
                                              c_void_p(te.data_ptr() if te is not None else 0), side_np.ctypes.data_as(c_void_p),
                                              c_void_p(sh.data_ptr()), c_void_p(oe.data_ptr()), c_void_p(ot.data_ptr()),
                                              c_void_p(_stream_ptr(dev))), self._h)
        self._keep = [te, sh, oe, ot]
        self.B, self.T = int(B), int(T)

    # -- compute ------------------------------------------------------------------------------
    def _need_cond(self):
        if self.B <= 0:
            raise TamfError("conditioning not set (call set_cond first)")

    def denoise(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        self._need_cond()
        dev = self.device
        xd = _dev_f32(x, dev)
        assert tuple(xd.shape) == (self.B, self.input_dim, 1, self.T), (tuple(xd.shape), self.B, self.T)
        td = t.detach().to(device=dev, dtype=torch.int64).contiguous()
        assert td.shape == (self.B,), (tuple(td.shape), self.B)
        # the reference indexes pe[timesteps] and raises on an index outside the table; so does this path (the device
        # kernel additionally clamps, so an unchecked caller of the C-ABI cannot read out of bounds)
        lo, hi = int(td.min()), int(td.max())
        if lo < 0 or hi >= self.max_timesteps:
            raise IndexError(f"timestep {lo if lo < 0 else hi} outside the timestep-embedding table [0, {self.max_timesteps})")
        out = torch.empty_like(xd)
        with torch.cuda.device(dev):
            _check(self._L.tamf_denoise(self._h, c_void_p(xd.data_ptr()), c_void_p(td.data_ptr()), c_void_p(out.data_ptr()),
                                      c_void_p(_stream_ptr(dev))), self._h)
        self._raise_on_range()
        return out

    def ddpm_step(self, x_t: torch.Tensor, x0: torch.Tensor, t: int, noise: Optional[torch.Tensor]) -> torch.Tensor:
        dev = self.device
        xt, x0d = _dev_f32(x_t, dev), _dev_f32(x0, dev)
        nz = _dev_f32(noise, dev) if noise is not None else None
        out = torch.empty_like(xt)
        with torch.cuda.device(dev):
            _check(self._L.tamf_ddpm_step(self._h, c_void_p(xt.data_ptr()), c_void_p(x0d.data_ptr()), int(t),
                                        c_void_p(nz.data_ptr() if nz is not None else 0), c_void_p(out.data_ptr()),
                                        xt.numel(), c_void_p(_stream_ptr(dev))), self._h)
        return out

    def sample_loop(self, noise: Optional[torch.Tensor] = None, seed: int = 0, clip_id_base: int = 0,
                    dump: bool = False, use_graph: bool = True, out: Optional[torch.Tensor] = None):
        """noise: (n_steps+1, B, F, 1, T) draws in reference call order, or None for device Philox."""
        self._need_cond()
        dev = self.device
        shape = (self.B, self.input_dim, 1, self.T)
        nz = None
        if noise is not None:
            nz = _dev_f32(noise, dev)
            assert tuple(nz.shape) == (self.n_steps + 1,) + shape, (tuple(nz.shape), shape)
        if out is None:
            out = torch.empty(shape, device=dev, dtype=torch.float32)
        dmp = torch.empty((self.n_steps,) + shape, device=dev, dtype=torch.float32) if dump else None
        with torch.cuda.device(dev):
            _check(self._L.tamf_sample_loop(self._h, c_void_p(nz.data_ptr() if nz is not None else 0), int(seed) & (2**64 - 1),
                                          int(clip_id_base), c_void_p(out.data_ptr()),
                                          c_void_p(dmp.data_ptr() if dmp is not None else 0), 1 if use_graph else 0,
                                          c_void_p(_stream_ptr(dev))), self._h)
        self._keep_loop = [nz, dmp]
        self._raise_on_range()
        return (out, dmp) if dump else out

    def _raise_on_range(self):
        if self.range_check and (self.status_flags(clear=True) & STATUS_F16_RANGE):
            raise TamfRangeError("f16x3: an activation beyond +-65504 was stored as a split-fp16 operand by this context; the result may "
                                 "differ from the reference's fp32 arithmetic (use precision='f32' or 'bf16x3')")

    def status_flags(self, clear: bool = True) -> int:
        """Sticky status bits of THIS context (STATUS_F16_RANGE: one of its launches stored an activation beyond +-65504 as a
        split-fp16 operand since its last clear; other contexts on the device have their own words).  Synchronises the
        current stream."""
        v = ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _check(self._L.tamf_get_status_flags(self._h, ctypes.byref(v), 1 if clear else 0, c_void_p(_stream_ptr(self.device))),
                   self._h)
        return int(v.value)

    def resize(self, max_batch: int, max_frames: int) -> None:
        """Re-dimension the workspaces for (max_batch, max_frames); the uploaded weights and the schedule stay.  Conditioning must be
        set again."""
        with torch.cuda.device(self.device):
            _check(self._L.tamf_ctx_resize(self._h, int(max_batch), int(max_frames)), self._h, self._L)
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)
        self.B = self.T = 0
        self._keep = []

    def check_guards(self) -> int:
        """Test hook: verify the guard bands around every device allocation of this context (set_guard_bytes() before it was
        created).  Raises TamfError naming the allocations a kernel wrote outside of; returns the number of guarded allocations."""
        n = c_int32(0)
        _check(self._L.tamf_test_check_guards(self._h, ctypes.byref(n)), self._h, self._L)
        return int(n.value)

    def refine(self, sample_pose_repr: torch.Tensor, h2o_dist: torch.Tensor) -> torch.Tensor:
        dev = self.device
        xin, h2o = _dev_f32(sample_pose_repr, dev), _dev_f32(h2o_dist, dev)
        assert tuple(xin.shape) == (self.B, self.T, self.input_dim)
        assert tuple(h2o.shape) == (self.B, self.T, self.h2o_dim)
        out = torch.empty_like(xin)
        with torch.cuda.device(dev):
            _check(self._L.tamf_refine(self._h, c_void_p(xin.data_ptr()), c_void_p(h2o.data_ptr()), c_void_p(out.data_ptr()),
                                     c_void_p(_stream_ptr(dev))), self._h)
        self._raise_on_range()
        return out

    def loop_stats(self):
        """(graph captures so far, graph launches of the last sample_loop call)"""
        a, b = c_int32(), c_int32()
        _check(self._L.tamf_loop_stats(self._h, ctypes.byref(a), ctypes.byref(b)), self._h, self._L)
        return int(a.value), int(b.value)

    @property
    def step_kernel_count(self) -> int:
        return int(self._L.tamf_step_kernel_count(self._h))

    def step_profile(self, max_n: int = 256):
        """[(name, ms, algorithmic_flops)] of one denoiser step, measured with HIP events on the launch stream."""
        self._need_cond()
        ms = (c_float * max_n)()
        fl = (c_double * max_n)()
        names = ctypes.create_string_buffer(max_n * 48)
        with torch.cuda.device(self.device):
            n = self._L.tamf_step_profile(self._h, max_n, ms, fl, names, c_void_p(_stream_ptr(self.device)))
        if n < 0:
            _check(n, self._h, self._L)
        out = []
        for i in range(n):
            nm = names.raw[i * 48:(i + 1) * 48].split(b"\0", 1)[0].decode()
            out.append((nm, float(ms[i]), float(fl[i])))
        return out


    def refine_profile(self, sample_pose_repr: torch.Tensor, h2o_dist: torch.Tensor, max_n: int = 256):
        """[(name, ms, algorithmic_flops)] of one refine() call of an R context, HIP events on the launch stream."""
        dev = self.device
        xin, h2o = _dev_f32(sample_pose_repr, dev), _dev_f32(h2o_dist, dev)
        assert tuple(xin.shape) == (self.B, self.T, self.input_dim) and tuple(h2o.shape) == (self.B, self.T, self.h2o_dim)
        out = torch.empty_like(xin)
        ms, fl, names = (c_float * max_n)(), (c_double * max_n)(), ctypes.create_string_buffer(max_n * 48)
        with torch.cuda.device(dev):
            n = self._L.tamf_refine_profile(self._h, c_void_p(xin.data_ptr()), c_void_p(h2o.data_ptr()), c_void_p(out.data_ptr()), max_n,
                                          ms, fl, names, c_void_p(_stream_ptr(dev)))
        if n < 0:
            _check(n, self._h, self._L)
        return [(names.raw[i * 48:(i + 1) * 48].split(b"\0", 1)[0].decode(), float(ms[i]), float(fl[i])) for i in range(n)]


# ---- kernel-level test hooks ---------------------------------------------------------------------


def test_gemm(precision: str, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0) -> torch.Tensor:
    dev = require_gpu(a.device)
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), device=dev, dtype=torch.float32)
    a, w = _dev_f32(a, dev), _dev_f32(w, dev)
    b = _dev_f32(bias, dev) if bias is not None else None
    _check(hooks().tamf_test_gemm(PRECISIONS[precision], M, N, K, c_void_p(a.data_ptr()), c_void_p(w.data_ptr()),
                                c_void_p(b.data_ptr() if b is not None else 0), act, c_void_p(c.data_ptr()),
                                c_void_p(_stream_ptr(dev))), None, hooks())
    return c


def block_stats(x: torch.Tensor) -> torch.Tensor:
    """(S_b, Q_b) of every 32-column block of every row of x (M, N): sum, and sum of squares about the block's own mean -> (M, N / 32, 2)"""
    M, N = x.shape
    b = x.double().reshape(M, N // 32, 32)
    S = b.sum(-1)
    Q = ((b - S[..., None] / 32.0) ** 2).sum(-1)
    return torch.stack([S, Q], -1).float()


def test_gemm_resid(precision: str, a, w, bb, gamma, x, stats_in=None):
    """tamf_test_gemm_resid: returns (x_next (M, N), stats_out (M, N / 32, 2)); x is not modified"""
    dev = require_gpu(a.device)
    M, K = a.shape
    N = w.shape[0]
    a, w, bb, gamma = [_dev_f32(t, dev) for t in (a, w, bb, gamma)]
    y = _dev_f32(x, dev).clone()
    st_in = _dev_f32(stats_in, dev) if stats_in is not None else None
    st_out = torch.empty((M, N // 32, 2), device=dev, dtype=torch.float32)
    _check(hooks().tamf_test_gemm_resid(PRECISIONS[precision], M, N, K, c_void_p(a.data_ptr()), c_void_p(w.data_ptr()),
                                      c_void_p(bb.data_ptr()), c_void_p(gamma.data_ptr()),
                                      c_void_p(st_in.data_ptr() if st_in is not None else 0), c_void_p(y.data_ptr()),
                                      c_void_p(st_out.data_ptr()), c_void_p(_stream_ptr(dev))), None, hooks())
    return y, st_out


def test_attention(precision: str, qkv: torch.Tensor, H: int) -> torch.Tensor:
    dev = require_gpu(qkv.device)
    B, S, D3 = qkv.shape
    d = D3 // 3
    out = torch.empty((B, S, d), device=dev, dtype=torch.float32)
    q = _dev_f32(qkv, dev)
    _check(hooks().tamf_test_attention(PRECISIONS[precision], B, S, H, d // H, c_void_p(q.data_ptr()), c_void_p(out.data_ptr()),
                                     c_void_p(_stream_ptr(dev))), None, hooks())
    return out


def test_philox(seed: int, clip_id_base: int, draw: int, B: int, F: int, T: int, device=None) -> torch.Tensor:
    dev = require_gpu(device)
    out = torch.empty((B, F, 1, T), device=dev, dtype=torch.float32)
    _check(hooks().tamf_test_philox(seed, clip_id_base, draw, B, F, T, c_void_p(out.data_ptr()), c_void_p(_stream_ptr(dev))), None, hooks())
    torch.cuda.synchronize(dev)
    return out


def mfma_sustained_rate(precision: str, millis: int = 1500, device=None) -> Tuple[float, float]:
    """(dense TFLOP/s, implied shader MHz) of register-only MFMA loops in the mode's instruction on every SIMD of the device for
    about `millis` ms: what the matrix pipe sustains on this board with real operand bits (include/tamf_hip.h tamf_bench_mfma_rate)."""
    dev = require_gpu(device)
    tf, mhz = ctypes.c_float(), ctypes.c_float()
    with torch.cuda.device(dev):
        _check(hooks().tamf_bench_mfma_rate(PRECISIONS[precision], int(millis), ctypes.byref(tf), ctypes.byref(mhz), c_void_p(_stream_ptr(dev))), None, hooks())
    return float(tf.value), float(mhz.value)
