"""InterationSegmentMDM - the MF-MDM "G" denoiser behind the reference's module interface
(reference model/interaction_segment_mdm.py:12-174), computed by the gfx950 HIP library.

Same constructor keywords, same parameter / buffer names (so the reference's checkpoints load with
load_state_dict, launch/sample.py:190-192) and the same call contract `model(x, timesteps, batch=...)`
(x: (B, input_dim, 1, T) float32, timesteps: (B,) int64, returns x0_hat of x's shape).  The module is only a
parameter container plus a thin dispatcher: forward() hands the tensors to libtamf_hip (hip_backend.TamfContext);
there is no PyTorch compute path and no CPU fallback.

Differences from the reference, all deliberate:
  * the frozen CLIP text tower is not re-run inside every forward (the reference does, :145); pass its output as
    batch["text_embedding"] (B, clip_dim) float32 - or construct with load_clip=True (needs the `clip` package)
    and pass batch["text"], in which case it is encoded once per distinct batch dict;
  * step-invariant conditioning (prefix tokens 1..4, object half of input_merge.0) is computed once per batch dict;
  * eval only (dropout is identity in the reference's eval mode; autograd is not supported);
  * arithmetic: `precision` (default "f16x3", hip_backend.DEFAULT_PRECISION - the same default in the CLIs and bench.py)
    selects the MFMA operand format.  "f16x3" (split fp16, fp32-equivalent at the stated 1e-5 tolerance) cannot hold ACTIVATIONS
    beyond +-65504 (weights are pre-scaled per tensor by a power of two and fit whatever their magnitude; only a non-finite
    weight is refused); `range_check` says what happens when an activation leaves that range or a weight is non-finite:
    "fallback" (default) - the call is repeated in "f32" (the reference's own arithmetic) on a new library context and the
    module stays there; "raise" - hip_backend.TamfRangeError; "off" - no check (no stream synchronisation per call);
  * train()/eval() return self (the reference's override returns None, :176-178).
"""
from __future__ import annotations

import logging
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

_logger = logging.getLogger(__name__)


def _default_precision() -> str:
    from ..hip_backend import DEFAULT_PRECISION

    return DEFAULT_PRECISION


class PositionalEncoding(nn.Module):
    """sin/cos table buffer `pe` (max_len, 1, d) - reference :181-198 (only the buffer is used here)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1).contiguous())


class TimestepEmbedder(nn.Module):
    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))


class _Linear(nn.Module):
    def __init__(self, attr, in_f, out_f):
        super().__init__()
        setattr(self, attr, nn.Linear(in_f, out_f))


class HandsideProcess(nn.Module):
    def __init__(self, latent_dim):
        super().__init__()
        self.register_buffer("rh_embed", torch.zeros(latent_dim))
        lh = torch.zeros(latent_dim)
        lh[0] = 1.0
        self.register_buffer("lh_embed", lh)


class _HipDenoiserBase(nn.Module):
    """Shared HIP plumbing of the G and R modules."""

    kind = "G"
    supports_fused_loop = False

    def _init_hip(self, arch: Dict[str, int], precision: str, max_batch: Optional[int], max_frames: Optional[int],
                  range_check: str = "fallback", per_clip_object_mean: bool = False):
        if range_check not in ("fallback", "raise", "off"):
            raise ValueError(f"range_check must be 'fallback', 'raise' or 'off', got {range_check!r}")
        # False: the object means run over all rows of the (zero-padded) batch - the reference's forward on the batch it is given.
        # True: over each clip's own batch["obj_num"] objects - what the reference's launchers get by calling the model one clip at a
        # time (launch/sample.py:206, launch/sample_refine.py:228); the batched launchers of this package set it.
        self.per_clip_object_mean = bool(per_clip_object_mean)
        self._arch = dict(arch)
        self.precision = precision          # what the caller asked for
        self.active_precision = precision   # what the library context runs ("f32" after a range fallback)
        self.range_check = range_check
        self._max_batch, self._max_frames = max_batch, max_frames
        self._ctx = None
        self._ctx_dirty = True
        self._cond_key = None
        self._sched_key = None
        self._max_timesteps = 5000  # rows of sequence_pos_encoder.pe: every t the reference accepts

    # weights changed -> re-upload lazily
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._ctx_dirty = True
        self.active_precision = self.precision  # new weights: try the requested arithmetic again
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._ctx_dirty = True
        return out

    def refresh_hip_weights(self):
        """Call after modifying parameters in place."""
        self._ctx_dirty = True

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _context(self, B: int, T: int):
        from ..hip_backend import TamfContext, require_gpu

        dev = require_gpu(self._device())
        need_new = self._ctx is None or self._ctx_dirty or self._ctx.device != dev
        if not need_new and self._ctx.precision == self.active_precision and (B > self._ctx.max_batch or T > self._ctx.max_frames):
            # a larger batch / longer clips than the workspaces hold: re-dimension them, the uploaded weights stay (tamf_ctx_resize)
            self._ctx.resize(max(B, self._ctx.max_batch), max(T, self._ctx.max_frames))
            self._cond_key = None
        if need_new or self._ctx.precision != self.active_precision:
            from ..hip_backend import TamfRangeError

            if self._ctx is not None:
                self._ctx.close()
                self._ctx = None
            mb = max(B, self._max_batch or 0)
            mf = max(T, self._max_frames or 0)
            while True:
                ctx = TamfContext(self._arch, mb, mf, precision=self.active_precision, device=dev, kind=self.kind)
                try:
                    ctx.load_state_dict(self.state_dict(), max_timesteps=self._max_timesteps, strict_weight_range=self.range_check != "off")
                    break
                except TamfRangeError as e:  # a weight beyond the fp16 range
                    ctx.close()
                    if self.range_check != "fallback" or self.active_precision == "f32":
                        raise
                    _logger.warning("%s - falling back to f32 arithmetic", e)
                    self.active_precision = "f32"
            self._ctx = ctx  # (its status word is its own and starts clear: nothing to reset, nobody else's evidence to erase)
            self._ctx_dirty = False
            self._cond_key = None
            self._sched_key = None
        return self._ctx

    def _guarded(self) -> bool:
        return self.active_precision == "f16x3" and self.range_check != "off"

    def _range_tripped(self, ctx) -> bool:
        """After a call in f16x3: True when an activation left the fp16 range and the call must be repeated in f32
        (range_check="fallback"); raises for range_check="raise"."""
        if not self._guarded():
            return False
        from ..hip_backend import STATUS_F16_RANGE, TamfRangeError

        if not (ctx.status_flags(clear=True) & STATUS_F16_RANGE):
            return False
        msg = ("f16x3: an activation beyond +-65504 was stored as a split-fp16 operand; the result may differ from the "
               "reference's fp32 arithmetic")
        if self.range_check == "raise":
            raise TamfRangeError(msg + " (construct the module with precision='f32' or 'bf16x3')")
        _logger.warning("%s - repeating the call in f32 and staying there", msg)
        self.active_precision = "f32"
        return True

    @staticmethod
    def _tensor_key(t):
        return None if t is None else (t.data_ptr(), tuple(t.shape), t._version, str(t.device))

    def _set_cond(self, ctx, batch, text_embedding):
        """Step-invariant conditioning is precomputed once per batch: the cache key is object identity + tensor
        address / shape / version, and the keyed OBJECTS are kept alive next to the key - a freed batch dict or tensor
        can therefore never be mistaken for a new one that the allocator placed at the same address."""
        tensors = (text_embedding, batch["shape"], batch["obj_embedding"], batch["obj_traj"])
        obj_num = None
        if self.per_clip_object_mean:
            if "obj_num" not in batch:
                raise KeyError("per_clip_object_mean=True needs batch['obj_num'] (the collate carries it, dataset/collate.py)")
            n = batch["obj_num"]
            obj_num = tuple(int(v) for v in (n.tolist() if hasattr(n, "tolist") else n))
        key = (id(batch), tuple(batch["hand_side"]), obj_num) + tuple(self._tensor_key(t) for t in tensors)
        if key != self._cond_key:
            ctx.set_cond(text_embedding, batch["hand_side"], batch["shape"], batch["obj_embedding"], batch["obj_traj"], obj_num=obj_num)
            self._cond_key = key
            self._cond_refs = (batch,) + tensors

    def train(self, mode: bool = True):
        if mode:
            _logger.warning("the MI355X HIP denoiser is inference-only; train(True) only flips the flag")
        return super().train(mode)


class InterationSegmentMDM(_HipDenoiserBase):
    kind = "G"
    supports_fused_loop = True

    def __init__(self, input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768, latent_dim=256,
                 ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, activation="gelu", clip_dim=512,
                 clip_version="ViT-B/32", precision: Optional[str] = None, load_clip: bool = False,
                 max_batch: Optional[int] = None, max_frames: Optional[int] = None, range_check: str = "fallback",
                 per_clip_object_mean: bool = False, **kargs):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("the HIP FFN kernel fuses the exact erf-GELU (activation='gelu') only")
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.activation, self.clip_dim = dropout, activation, clip_dim
        self.input_feats, self.obj_input_feats = input_dim, obj_input_dim
        self.hand_shape_feats, self.obj_embed_feats = hand_shape_dim, obj_embed_dim
        self.cond_mask_prob = kargs.get("cond_mask_prob", 0.0)
        if self.cond_mask_prob:
            raise NotImplementedError("cond_mask_prob > 0 is a training feature (never set by the launchers)")

        self.hand_side_process = HandsideProcess(latent_dim)
        self.hand_shape_process = _Linear("shape_embed", hand_shape_dim, latent_dim)
        self.obj_embed_process = _Linear("embedding", obj_embed_dim, latent_dim)
        self.input_process = _Linear("poseEmbedding", input_dim, latent_dim)
        self.obj_input_process = _Linear("poseEmbedding", obj_input_dim, latent_dim)
        self.input_merge = nn.Sequential(nn.Linear(latent_dim * 2, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size, dropout=dropout,
                                           activation=activation)
        self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.embed_text = nn.Linear(clip_dim, latent_dim)
        self.output_process = _Linear("poseFinal", latent_dim, input_dim)
        self.clip_version = clip_version
        self.clip_model = self.load_and_freeze_clip(clip_version) if load_clip else None
        self._text_cache = (None, None)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()
        self._init_hip(dict(input_dim=input_dim, obj_input_dim=obj_input_dim, hand_shape_dim=hand_shape_dim,
                            obj_embed_dim=obj_embed_dim, latent_dim=latent_dim, ff_size=ff_size, num_layers=num_layers,
                            num_heads=num_heads, clip_dim=clip_dim), precision or _default_precision(), max_batch, max_frames,
                       range_check, per_clip_object_mean)

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith("clip_model.")]

    def load_and_freeze_clip(self, clip_version):
        try:
            import clip  # the reference's dependency (thirdparty/CLIP); optional here
        except ImportError as e:
            raise ImportError("load_clip=True needs the `clip` package; otherwise pass batch['text_embedding']") from e
        clip_model, _ = clip.load(clip_version, device="cpu", jit=False)
        clip_model.eval()
        for p in clip_model.parameters():
            p.requires_grad = False
        return clip_model

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        # checkpoints never contain the CLIP tower (util/state_util.py:32-34)
        return type(sd)((k, v) for k, v in sd.items() if not k.startswith("clip_model."))

    def encode_text(self, raw_text):
        """CLIP text features (B, clip_dim) float32 - reference :111-132 (context 20+2, zero-padded to 77)."""
        if self.clip_model is None:
            raise KeyError("batch['text_embedding'] missing and no CLIP tower loaded (construct with load_clip=True)")
        import clip

        dev = self._device()
        texts = clip.tokenize(raw_text, context_length=22, truncate=True).to(dev)
        texts = torch.cat([texts, torch.zeros([texts.shape[0], 77 - 22], dtype=texts.dtype, device=dev)], dim=1)
        return self.clip_model.encode_text(texts).float()

    def _text_embedding(self, batch):
        if "text_embedding" in batch and batch["text_embedding"] is not None:
            return batch["text_embedding"]
        key = (id(batch), tuple(batch["text"]))
        if self._text_cache[0] != key:
            with torch.no_grad():
                self._text_cache = (key, self.encode_text(batch["text"]))
        return self._text_cache[1]

    @torch.no_grad()
    def forward(self, x, timesteps, batch):
        """x: (B, input_dim, 1, T); timesteps: (B,) int; batch: dict with "text_embedding" (or "text"),
        "hand_side", "shape", "obj_embedding", "obj_traj"  ->  x0_hat (B, input_dim, 1, T)."""
        B, _, _, T = x.shape
        while True:
            ctx = self._context(B, T)
            self._set_cond(ctx, batch, self._text_embedding(batch))
            out = ctx.denoise(x, timesteps)
            if not self._range_tripped(ctx):
                return out

    @torch.no_grad()
    def fused_sample_loop(self, diffusion, shape, x_T=None, batch=None, dump_steps=None, noise_source="philox",
                          seed=None, clip_id_base=0, device=None):
        """The whole p_sample_loop as one library call (hipGraph replay); see GaussianDiffusion.p_sample_loop."""
        B, F, _, T = shape
        N = diffusion.num_timesteps
        # (a respaced sampler visits timesteps of its BASE process: the timestep table must reach them)
        n_table = max(N, int(getattr(diffusion, "original_num_steps", N)))
        if n_table > self._max_timesteps:
            self._max_timesteps = n_table
            self._ctx_dirty = True
        draws = None
        while True:
            res, draws = self._fused_loop_once(diffusion, shape, x_T, batch, dump_steps, noise_source, seed, clip_id_base, draws)
            if not self._range_tripped(self._ctx):
                return res

    def _fused_loop_once(self, diffusion, shape, x_T, batch, dump_steps, noise_source, seed, clip_id_base, draws):
        B, F, _, T = shape
        N = diffusion.num_timesteps
        ctx = self._context(B, T)
        # keyed on the coefficient values themselves (a new diffusion object at a recycled id must not hit)
        tmap = tuple(getattr(diffusion, "timestep_map", ())) if getattr(diffusion, "respaced", False) else None
        skey = (N, diffusion.posterior_mean_coef1.tobytes(), diffusion.posterior_mean_coef2.tobytes(),
                diffusion.posterior_log_variance_clipped.tobytes(), tmap)
        if self._sched_key != skey:
            ctx.set_schedule(diffusion.posterior_mean_coef1, diffusion.posterior_mean_coef2,
                             diffusion.posterior_log_variance_clipped, timestep_map=tmap)
            self._sched_key = skey
        self._set_cond(ctx, batch, self._text_embedding(batch))
        dev = ctx.device
        noise = draws  # (a repeat after a range fallback reuses the first attempt's draws: the torch generators have moved on)
        if noise is None and (noise_source in ("torch", "torch_cpu") or x_T is not None):
            gen_dev = torch.device("cpu") if noise_source == "torch_cpu" else dev
            n_bytes = (N + 1) * B * F * T * 4
            if n_bytes > 8 << 30:
                raise MemoryError(f"pre-drawn noise would need {n_bytes / 2**30:.1f} GiB; use noise_source='philox'")
            draws = torch.empty((N + 1, B, F, 1, T), dtype=torch.float32, device=dev)
            if noise_source == "philox":  # explicit x_T + device Philox for the step noise is not expressible: draw on device
                noise_source = "torch"
                gen_dev = dev
            draws[0] = x_T.to(dev) if x_T is not None else torch.randn(*shape, device=gen_dev).to(dev)
            for k in range(1, N + 1):  # th.randn_like(x) once per step, reference :448
                draws[k] = torch.randn(*shape, device=gen_dev).to(dev)
            noise = draws
        if seed is None:
            seed = torch.initial_seed()
        res = ctx.sample_loop(noise=noise, seed=seed, clip_id_base=clip_id_base, dump=dump_steps is not None)
        if dump_steps is not None:
            out, dump = res
            return [dump[i].clone() for i in range(N) if i in dump_steps], noise
        return res, noise
