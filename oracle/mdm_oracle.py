"""CPU restatement of the reference's MF-MDM denoiser (G), refiner trunk (R) and DDPM sampler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU tensor algebra, batch-first,
no nn.Module machinery; every function cites the reference lines it restates
(paths relative to /root/reference/src/oakink2_tamf/).

The dtype is selectable: float32 reproduces the reference's arithmetic type, float64 gives a
"truth" run used to calibrate tolerances.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import det

# --------------------------------------------------------------------------------------
# Architecture description (config/arch_mdm.yml, config/arch_mdm_l.yml, config/arch_refine.yml)
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class Arch:
    input_dim: int = 99
    obj_input_dim: int = 9
    hand_shape_dim: int = 10
    obj_embed_dim: int = 768
    latent_dim: int = 256
    ff_size: int = 1024
    num_layers: int = 8
    num_heads: int = 4
    clip_dim: int = 512
    h2o_dim: int = 778  # R only (model/segment_refine_model.py:267-279)
    kind: str = "G"  # "G" = InterationSegmentMDM, "R" = SegmentRefineModel trunk

    @property
    def prefix_len(self) -> int:
        return 5 if self.kind == "G" else 3


ARCH_MDM = Arch(latent_dim=256, ff_size=1024)
ARCH_MDM_L = Arch(latent_dim=512, ff_size=2048)
ARCH_REFINE = Arch(latent_dim=256, ff_size=1024, kind="R")
ARCH_TINY = Arch(latent_dim=128, ff_size=256, num_layers=2, num_heads=2)
ARCH_TINY_R = Arch(latent_dim=128, ff_size=256, num_layers=2, num_heads=2, kind="R")

# --------------------------------------------------------------------------------------
# Diffusion schedule tables (float64)
# --------------------------------------------------------------------------------------


def cosine_betas(n: int, max_beta: float = 0.999) -> np.ndarray:
    """model/diffusion/gaussian_diffusion.py:36-40,45-62 (cosine alpha_bar, clipped betas)."""

    def alpha_bar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    out = []
    for i in range(n):
        out.append(min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta))
    return np.array(out, dtype=np.float64)


def linear_betas(n: int) -> np.ndarray:
    """gaussian_diffusion.py:29-35 (scale_betas = 1)."""
    scale = 1000 / n
    return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


@dataclass
class DiffusionTables:
    """Float64 tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:116-161) as seen through
    SpacedDiffusion with use_timesteps = all (respace.py:69-83): the betas are re-derived from the
    base process' cumulative alphas before the tables are built."""

    num_timesteps: int
    betas: np.ndarray
    alphas_cumprod: np.ndarray
    alphas_cumprod_prev: np.ndarray
    posterior_variance: np.ndarray
    posterior_log_variance_clipped: np.ndarray
    posterior_mean_coef1: np.ndarray
    posterior_mean_coef2: np.ndarray
    sqrt_alphas_cumprod: np.ndarray
    sqrt_one_minus_alphas_cumprod: np.ndarray
    timestep_map: Optional[List[int]] = None  # base timestep of every step (make_tables; None = identity)


def _tables_from_betas(betas: np.ndarray) -> DiffusionTables:
    betas = np.array(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    plv = np.log(np.append(pv[1], pv[1:]))
    c1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
    c2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)
    return DiffusionTables(
        num_timesteps=int(betas.shape[0]),
        betas=betas,
        alphas_cumprod=ac,
        alphas_cumprod_prev=ac_prev,
        posterior_variance=pv,
        posterior_log_variance_clipped=plv,
        posterior_mean_coef1=c1,
        posterior_mean_coef2=c2,
        sqrt_alphas_cumprod=np.sqrt(ac),
        sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
    )


def space_timesteps(num_timesteps: int, section_counts) -> List[int]:
    """model/diffusion/respace.py:8-57: the kept timesteps of a respaced process, sorted ("N" / "a,b,c" section counts, or "ddimN")."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return list(range(0, num_timesteps, stride))
            raise ValueError("cannot create exactly that many steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, out = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            out.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(out))


def make_tables(steps: int, schedule: str = "cosine", use_timesteps: Optional[Sequence[int]] = None) -> DiffusionTables:
    """model/diffusion_util.py:5-31 -> SpacedDiffusion (respace.py:69-83): use_timesteps = None keeps every timestep (what the
    launchers run); a subset re-derives the betas of the kept steps from the base process' cumulative alphas.  The returned tables
    carry `timestep_map` (attribute set below): the base timestep of every kept step."""
    if schedule == "cosine":
        base_betas = cosine_betas(steps)
    elif schedule == "linear":
        base_betas = linear_betas(steps)
    else:
        raise NotImplementedError(f"unknown beta schedule: {schedule}")
    base = _tables_from_betas(base_betas)
    keep = set(range(steps)) if use_timesteps is None else set(int(t) for t in use_timesteps)
    last = 1.0
    new_betas, tmap = [], []
    for i, ac in enumerate(base.alphas_cumprod):
        if i in keep:
            new_betas.append(1 - ac / last)
            last = ac
            tmap.append(i)
    tab = _tables_from_betas(np.array(new_betas))
    tab.timestep_map = tmap
    return tab


# --------------------------------------------------------------------------------------
# Weights
# --------------------------------------------------------------------------------------


def positional_table(d: int, max_len: int = 5000) -> torch.Tensor:
    """model/interaction_segment_mdm.py:186-193 - float32 sin/cos table, shape (max_len, d)."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-np.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def state_dict_spec(arch: Arch) -> Dict[str, tuple]:
    """Key -> shape of the checkpoint format (util/state_util.py:22-39)."""
    d, ff = arch.latent_dim, arch.ff_size
    spec: Dict[str, tuple] = {}

    def lin(name, out_f, in_f):
        spec[f"{name}.weight"] = (out_f, in_f)
        spec[f"{name}.bias"] = (out_f,)

    spec["hand_side_process.rh_embed"] = (d,)
    spec["hand_side_process.lh_embed"] = (d,)
    lin("hand_shape_process.shape_embed", d, arch.hand_shape_dim)
    lin("obj_embed_process.embedding", d, arch.obj_embed_dim)
    lin("input_process.poseEmbedding", d, arch.input_dim)
    lin("obj_input_process.poseEmbedding", d, arch.obj_input_dim)
    if arch.kind == "R":
        lin("h2o_dist_input_process.poseEmbedding", d, arch.h2o_dim)
        lin("input_merge.0", d, 3 * d)
    else:
        lin("input_merge.0", d, 2 * d)
    lin("input_merge.2", d, d)
    spec["sequence_pos_encoder.pe"] = (5000, 1, d)
    for l in range(arch.num_layers):
        p = f"seqTransEncoder.layers.{l}"
        spec[f"{p}.self_attn.in_proj_weight"] = (3 * d, d)
        spec[f"{p}.self_attn.in_proj_bias"] = (3 * d,)
        lin(f"{p}.self_attn.out_proj", d, d)
        lin(f"{p}.linear1", ff, d)
        lin(f"{p}.linear2", d, ff)
        for n in ("norm1", "norm2"):
            spec[f"{p}.{n}.weight"] = (d,)
            spec[f"{p}.{n}.bias"] = (d,)
    if arch.kind == "G":
        spec["embed_timestep.sequence_pos_encoder.pe"] = (5000, 1, d)
        lin("embed_timestep.time_embed.0", d, d)
        lin("embed_timestep.time_embed.2", d, d)
        lin("embed_text", d, arch.clip_dim)
    lin("output_process.poseFinal", arch.input_dim, d)
    return spec


def det_state_dict(arch: Arch, tag: str = "w0") -> Dict[str, torch.Tensor]:
    """Seed-free weights with PyTorch-default-like magnitudes (see oracle/det.py)."""
    d = arch.latent_dim
    sd: Dict[str, torch.Tensor] = {}
    pe = positional_table(d).unsqueeze(1).contiguous()  # (5000, 1, d)
    for name, shape in state_dict_spec(arch).items():
        if name.endswith(".pe"):
            sd[name] = pe
        elif name == "hand_side_process.rh_embed":
            sd[name] = torch.zeros(d)
        elif name == "hand_side_process.lh_embed":
            v = torch.zeros(d)
            v[0] = 1.0
            sd[name] = v
        elif ".norm" in name and name.endswith(".weight"):
            sd[name] = torch.from_numpy(1.0 + det.det_uniform(f"{tag}/{name}", shape, 0.1))
        elif ".norm" in name and name.endswith(".bias"):
            sd[name] = torch.from_numpy(det.det_uniform(f"{tag}/{name}", shape, 0.05))
        elif name.endswith("in_proj_weight"):
            sd[name] = torch.from_numpy(det.det_uniform(f"{tag}/{name}", shape, math.sqrt(6.0 / (4 * d))))
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            sd[name] = torch.from_numpy(det.det_uniform(f"{tag}/{name}", shape, 0.02))
        elif name.endswith(".weight"):
            sd[name] = torch.from_numpy(det.det_uniform(f"{tag}/{name}", shape, 1.0 / math.sqrt(shape[1])))
        elif name.endswith(".bias"):
            wshape = state_dict_spec(arch)[name[: -len("bias")] + "weight"]
            sd[name] = torch.from_numpy(det.det_uniform(f"{tag}/{name}", shape, 1.0 / math.sqrt(wshape[1])))
        else:  # pragma: no cover
            raise KeyError(name)
    return sd


def det_cond(B: int, T: int, nobj: int = 2, tag: str = "c0", arch: Arch = ARCH_MDM) -> Dict[str, object]:
    """Synthetic conditioning of SURVEY.md section 8(d): unit-normal embeddings, alternating hand side,
    per-clip constant betas."""
    shape = det.det_normal(f"{tag}/shape", (B, 1, arch.hand_shape_dim)).repeat(T, axis=1)
    return {
        "text_embedding": torch.from_numpy(det.det_normal(f"{tag}/text", (B, arch.clip_dim))),
        "hand_side": ["rh" if b % 2 == 0 else "lh" for b in range(B)],
        "shape": torch.from_numpy(np.ascontiguousarray(shape)),
        "obj_embedding": torch.from_numpy(det.det_normal(f"{tag}/obj_emb", (B, nobj, arch.obj_embed_dim))),
        "obj_traj": torch.from_numpy(det.det_normal(f"{tag}/obj_traj", (B, nobj, T, arch.obj_input_dim))),
    }


def det_state_dict_stress(arch: Arch, tag: str = "w0") -> Dict[str, torch.Tensor]:
    """det_state_dict with the dynamic range of a TRAINED network: LayerNorm gains
    log-uniform in [0.2, 5], and about 1 % of the rows of every layer's `linear1.weight` / `self_attn.in_proj_weight` scaled x30
    (outlier channels).  Seed-free like everything in oracle/det.py; used for the stress fixtures of the fp32-tolerance gate."""
    sd = det_state_dict(arch, tag)
    for name in list(sd):
        if ".norm" in name and name.endswith(".weight"):
            u = det.det_uniform(f"{tag}/stress-gamma/{name}", tuple(sd[name].shape), 1.0)  # in [-1, 1)
            sd[name] = torch.from_numpy(np.exp(u * math.log(5.0)).astype(np.float32))  # log-uniform in [0.2, 5)
        elif name.endswith("linear1.weight") or name.endswith("in_proj_weight"):
            rows = sd[name].shape[0]
            pick = det.det_uniform(f"{tag}/stress-rows/{name}", (rows,), 1.0) > 0.98  # ~1 % of the rows
            w = sd[name].clone()
            w[torch.from_numpy(pick)] *= 30.0
            sd[name] = w
    return sd


def det_state_dict_stress_dc(arch: Arch, tag: str = "w0") -> Dict[str, torch.Tensor]:
    """det_state_dict_stress + a DC component in every LayerNorm output: biases beta = 2 + U(-0.5, 0.5) for all features (round 5).
    The residual rows u = x + sublayer(x) then carry a row mean of about 2 beside a spread of 0.2 - 5 (the gains): |mean| / sigma up to
    ~10 - the case the deferred LayerNorm of the HIP path (the normalisation applied by the CONSUMER of u, the row mean cancelling inside
    the product against zero-sum weight rows) is most exposed to.  Fixtures: forward_stress_dc_t160.npz, loop_stress_dc_b2_t160_50.npz."""
    sd = det_state_dict_stress(arch, tag)
    for name in list(sd):
        if ".norm" in name and name.endswith(".bias"):
            u = det.det_uniform(f"{tag}/stress-beta/{name}", tuple(sd[name].shape), 0.5)
            sd[name] = torch.from_numpy((2.0 + u).astype(np.float32))
    return sd


def det_cond_stress(B: int, T: int, nobj: int = 2, tag: str = "c0", arch: Arch = ARCH_MDM) -> Dict[str, object]:
    """det_cond with the magnitudes of real conditioning: CLIP text features of norm 10, object trajectories as
    [translation in metres | rot6d of a unit rotation] (obj_input_dim = 9: dev_fn/transform/rotation.py rot6d = the first two
    columns of R, row-major), hand shape betas ~ N(0, 1)."""
    c = det_cond(B, T, nobj=nobj, tag=tag, arch=arch)
    te = c["text_embedding"]
    c["text_embedding"] = te / te.norm(dim=-1, keepdim=True) * 10.0
    raw = det.det_normal(f"{tag}/stress-traj", (B, nobj, T, 12))
    tsl = 0.3 * raw[..., :3]
    a1, a2 = raw[..., 3:6], raw[..., 6:9]
    b1 = a1 / np.linalg.norm(a1, axis=-1, keepdims=True)
    a2 = a2 - (b1 * a2).sum(-1, keepdims=True) * b1
    b2 = a2 / np.linalg.norm(a2, axis=-1, keepdims=True)
    rot6d = np.stack([b1, b2], axis=-1).reshape(B, nobj, T, 6)  # columns of R interleaved row-major: (r00, r01, r10, r11, r20, r21)
    c["obj_traj"] = torch.from_numpy(np.ascontiguousarray(np.concatenate([tsl, rot6d], axis=-1).astype(np.float32)))
    return c


# --------------------------------------------------------------------------------------
# Denoiser forward (G) and refiner trunk (R)
# --------------------------------------------------------------------------------------


def _lin(sd, name, x):
    w = sd[f"{name}.weight"].to(x.dtype)
    b = sd[f"{name}.bias"].to(x.dtype)
    return x @ w.t() + b


def _silu(x):
    return x * torch.sigmoid(x)


def _gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _hand_side_rows(sd, hand_side, dtype) -> torch.Tensor:
    """model/interaction_segment_mdm.py:266-288 ("rh" -> rh_embed, "lh" -> lh_embed, else ValueError)."""
    rows = []
    for hs in hand_side:
        if hs == "rh" or hs == 0:
            rows.append(sd["hand_side_process.rh_embed"])
        elif hs == "lh" or hs == 1:
            rows.append(sd["hand_side_process.lh_embed"])
        else:
            raise ValueError(f"unexpected hand_side: {hs}")
    return torch.stack(rows, dim=0).to(dtype)


def encoder_stack(sd, arch: Arch, seq: torch.Tensor) -> torch.Tensor:
    """8x post-LN nn.TransformerEncoderLayer (gelu, no mask, eval) - interaction_segment_mdm.py:63-70,171.
    seq: (B, S, d) batch-first."""
    B, S, d = seq.shape
    H = arch.num_heads
    hd = d // H
    dt = seq.dtype
    for l in range(arch.num_layers):
        p = f"seqTransEncoder.layers.{l}"
        w_in = sd[f"{p}.self_attn.in_proj_weight"].to(dt)
        b_in = sd[f"{p}.self_attn.in_proj_bias"].to(dt)
        qkv = seq @ w_in.t() + b_in
        q, k, v = qkv.split(d, dim=-1)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        a = (att @ v).transpose(1, 2).reshape(B, S, d)
        a = _lin(sd, f"{p}.self_attn.out_proj", a)
        seq = _layer_norm(seq + a, sd[f"{p}.norm1.weight"].to(dt), sd[f"{p}.norm1.bias"].to(dt))
        h = _gelu_erf(_lin(sd, f"{p}.linear1", seq))
        h = _lin(sd, f"{p}.linear2", h)
        seq = _layer_norm(seq + h, sd[f"{p}.norm2.weight"].to(dt), sd[f"{p}.norm2.bias"].to(dt))
    return seq


def denoiser_forward(
    sd: Dict[str, torch.Tensor],
    arch: Arch,
    x: torch.Tensor,
    t: torch.Tensor,
    cond: Dict[str, object],
    dtype: torch.dtype = torch.float32,
    encoder: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
) -> torch.Tensor:
    """InterationSegmentMDM.forward (model/interaction_segment_mdm.py:134-174), CLIP output supplied as
    cond["text_embedding"] (the value encode_text(...).float() would return, :132).
    x: (B, 99, 1, T); t: (B,) int; returns (B, 99, 1, T).
    `encoder` (batch-first (B, S, d) -> (B, S, d)) replaces encoder_stack: bench.py's same-box library yardstick passes torch's own
    nn.TransformerEncoder there (the module the reference instantiates, :63-70); every parity use leaves it None."""
    assert arch.kind == "G"
    B, F, _, T = x.shape
    dt = dtype
    pe = sd["sequence_pos_encoder.pe"][:, 0, :].to(dt)  # (5000, d)
    x = x.to(dt)

    # prefix tokens (:141-159)
    e_t = _lin(sd, "embed_timestep.time_embed.2", _silu(_lin(sd, "embed_timestep.time_embed.0", pe[t.long()])))
    e_txt = _lin(sd, "embed_text", cond["text_embedding"].to(dt))
    e_side = _hand_side_rows(sd, cond["hand_side"], dt)
    e_shp = _lin(sd, "hand_shape_process.shape_embed", cond["shape"].to(dt).mean(dim=1))
    e_obj = _lin(sd, "obj_embed_process.embedding", cond["obj_embedding"].to(dt).mean(dim=1))
    prefix = torch.nan_to_num(torch.stack([e_t, e_txt, e_side, e_shp, e_obj], dim=1))  # (B, 5, d)

    # per-frame tokens (:161-166)
    hand = _lin(sd, "input_process.poseEmbedding", x[:, :, 0, :].transpose(1, 2))  # (B, T, d)
    obj = _lin(sd, "obj_input_process.poseEmbedding", cond["obj_traj"].to(dt).permute(0, 2, 1, 3)).mean(dim=2)
    h = _lin(sd, "input_merge.2", _silu(_lin(sd, "input_merge.0", torch.cat([hand, obj], dim=-1))))
    h = torch.nan_to_num(h)

    seq = torch.cat([prefix, h], dim=1)  # (B, S, d)
    seq = seq + pe[: seq.shape[1]].unsqueeze(0)  # (:169-170, :195-198; dropout inactive)
    seq = (encoder(seq) if encoder is not None else encoder_stack(sd, arch, seq))[:, arch.prefix_len :, :]
    out = _lin(sd, "output_process.poseFinal", seq)  # (B, T, 99)
    out = out.transpose(1, 2).unsqueeze(2)  # (B, 99, 1, T) (:313-318)
    return torch.nan_to_num(out)


def refine_forward(
    sd: Dict[str, torch.Tensor],
    arch: Arch,
    sample_pose_repr: torch.Tensor,
    h2o_dist: torch.Tensor,
    cond: Dict[str, object],
    dtype: torch.dtype = torch.float32,
) -> torch.Tensor:
    """SegmentRefineModel.forward trunk (model/segment_refine_model.py:175-217) with the hand->object
    distance feature supplied.  sample_pose_repr: (B, T, 99); h2o_dist: (B, T, 778); returns (B, T, 99)."""
    assert arch.kind == "R"
    dt = dtype
    x_in = sample_pose_repr.to(dt)
    pe = sd["sequence_pos_encoder.pe"][:, 0, :].to(dt)
    e_side = _hand_side_rows(sd, cond["hand_side"], dt)
    e_shp = _lin(sd, "hand_shape_process.shape_embed", cond["shape"].to(dt).mean(dim=1))
    e_obj = _lin(sd, "obj_embed_process.embedding", cond["obj_embedding"].to(dt).mean(dim=1))
    prefix = torch.nan_to_num(torch.stack([e_side, e_shp, e_obj], dim=1))
    hand = _lin(sd, "input_process.poseEmbedding", x_in)
    obj = _lin(sd, "obj_input_process.poseEmbedding", cond["obj_traj"].to(dt).permute(0, 2, 1, 3)).mean(dim=2)
    dist = _lin(sd, "h2o_dist_input_process.poseEmbedding", h2o_dist.to(dt))
    h = _lin(sd, "input_merge.2", _silu(_lin(sd, "input_merge.0", torch.cat([hand, obj, dist], dim=-1))))
    h = torch.nan_to_num(h)
    seq = torch.cat([prefix, h], dim=1)
    seq = seq + pe[: seq.shape[1]].unsqueeze(0)
    seq = encoder_stack(sd, arch, seq)[:, arch.prefix_len :, :]
    out = x_in + _lin(sd, "output_process.poseFinal", seq)
    return torch.nan_to_num(out)


# --------------------------------------------------------------------------------------
# DDPM reverse process
# --------------------------------------------------------------------------------------


def ddpm_step(tab: DiffusionTables, x_t: torch.Tensor, x0_hat: torch.Tensor, i: int, noise: torch.Tensor,
              guidance: Optional[torch.Tensor] = None) -> torch.Tensor:
    """p_sample with START_X / FIXED_SMALL / clip_denoised=False
    (gaussian_diffusion.py:209-229,273-320,412-460; tables cast float64 -> float32 at :1275)."""
    dt = x_t.dtype
    # _extract_into_tensor casts the float64 table entry to float32 (.float()) before the arithmetic
    cast = (lambda v: torch.tensor(np.float32(v), dtype=torch.float32).to(dt)) if dt == torch.float32 else (
        lambda v: torch.tensor(np.float32(v), dtype=torch.float32).to(dt)
    )
    c1 = cast(tab.posterior_mean_coef1[i])
    c2 = cast(tab.posterior_mean_coef2[i])
    logvar = cast(tab.posterior_log_variance_clipped[i])
    mean = c1 * x0_hat + c2 * x_t
    if guidance is not None:  # condition_mean (gaussian_diffusion.py:346-357): mean + posterior_variance[t] * grad log p(y | x)
        mean = mean.float() + cast(tab.posterior_variance[i]) * guidance.float()
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * torch.exp(0.5 * logvar) * noise


def sample_loop(
    sd: Dict[str, torch.Tensor],
    arch: Arch,
    tab: DiffusionTables,
    cond: Dict[str, object],
    shape: Sequence[int],
    draw: Callable[[int], torch.Tensor],
    dtype: torch.dtype = torch.float32,
    n_steps: Optional[int] = None,
    dump: Optional[List[torch.Tensor]] = None,
    unhoisted: bool = True,
    cond_fn: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
) -> torch.Tensor:
    """p_sample_loop / p_sample_loop_progressive (gaussian_diffusion.py:506-640): x_T = draw(0), then for
    i = N-1 .. 0: x <- p_sample(x, i) with eps_i = draw(k), k = 1.. in call order (one draw per step,
    also at i = 0 where it is multiplied by zero).  `n_steps` < N runs only the first n_steps iterations
    (test helper, not a reference feature)."""
    B = shape[0]
    x = draw(0).to(dtype)
    N = tab.num_timesteps
    indices = list(range(N))[::-1]
    if n_steps is not None:
        indices = indices[:n_steps]
    tmap = getattr(tab, "timestep_map", None)
    for k, i in enumerate(indices):
        # (_WrappedModel, respace.py:114-119: the denoiser sees the BASE process' timestep of step i)
        t = torch.full((B,), i if tmap is None else tmap[i], dtype=torch.long)
        x0 = denoiser_forward(sd, arch, x, t, cond, dtype=dtype)
        # cond_fn(x_t, t) sees the same (mapped) timesteps as the model (respace.py:91-92)
        x = ddpm_step(tab, x, x0, i, draw(k + 1).to(dtype), guidance=None if cond_fn is None else cond_fn(x, t))
        if dump is not None:
            dump.append(x.clone())
    return x


# --------------------------------------------------------------------------------------
# Device noise generator restatement (Philox4x32-10 + Box-Muller), keyed (seed, clip, step, element)
# --------------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on uint32 numpy arrays; returns 4 uint32 arrays."""
    c0 = c0.astype(np.uint32).copy()
    c1 = c1.astype(np.uint32).copy()
    c2 = c2.astype(np.uint32).copy()
    c3 = c3.astype(np.uint32).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PHILOX_M0
            p1 = c2.astype(np.uint64) * _PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_normal(seed: int, clip_ids: np.ndarray, draw_index: int, n_feat: int, T: int) -> np.ndarray:
    """Noise tensor (len(clip_ids), n_feat, 1, T) float32 exactly as the HIP library generates it:
    element e = tau*128 + f (the frame-major, 128-padded order of the sampler state) of clip c at draw k uses counter
    (e >> 2, k, c_lo, c_hi), key (seed_lo, seed_hi); the four 32-bit outputs give two Box-Muller pairs; element e
    takes output (e & 3)."""
    n = n_feat * T
    out = np.empty((len(clip_ids), n), dtype=np.float32)
    ff, tt = np.meshgrid(np.arange(n_feat, dtype=np.uint64), np.arange(T, dtype=np.uint64), indexing="ij")
    e = (tt * np.uint64(128) + ff).reshape(-1)
    grp = (e >> np.uint64(2)).astype(np.uint32)
    lane = (e & np.uint64(3)).astype(np.int64)
    for r, c in enumerate(clip_ids):
        c = int(c)
        r0, r1, r2, r3 = philox4x32_10(
            grp,
            np.full(n, draw_index, dtype=np.uint32),
            np.full(n, c & 0xFFFFFFFF, dtype=np.uint32),
            np.full(n, (c >> 32) & 0xFFFFFFFF, dtype=np.uint32),
            seed & 0xFFFFFFFF,
            (seed >> 32) & 0xFFFFFFFF,
        )
        u = np.stack([r0, r1, r2, r3], axis=0).astype(np.float32)
        # uniform in (0,1]: (x + 1) * 2^-32 evaluated in float32 like the device code (x*2^-32 + 2^-33)
        uf = u * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)
        ra = np.sqrt(np.float32(-2.0) * np.log(uf[0]))
        rb = np.sqrt(np.float32(-2.0) * np.log(uf[2]))
        tw = np.float32(6.283185307179586)
        z = np.stack(
            [ra * np.cos(tw * uf[1]), ra * np.sin(tw * uf[1]), rb * np.cos(tw * uf[3]), rb * np.sin(tw * uf[3])], axis=0
        ).astype(np.float32)
        out[r] = z[lane, np.arange(n)]
    return out.reshape(len(clip_ids), n_feat, 1, T)
