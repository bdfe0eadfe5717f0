"""Timestep respacing (reference model/diffusion/respace.py:8-119).  The MF-MDM launchers always keep every
timestep, so there the map is the identity and the model wrapper is a no-op; the betas are nevertheless re-derived
from the base process' cumulative alphas as the reference does (the float64 round-trip is part of parity).
Subsets of the timesteps (`space_timesteps(1000, "100")`, "ddim50" strides ...) are supported since round 6."""
from __future__ import annotations

import numpy as np

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """Evenly strided subsets per section; "ddimN" = fixed stride with exactly N steps (reference :8-57)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps a subset of the base process' timesteps (reference :60-104): the betas of the kept steps are
    re-derived from the base process' cumulative alphas, `timestep_map[i]` is the base timestep of step i, and the model is evaluated at
    the MAPPED timestep (`_WrappedModel`, :107-119).  With every step kept - what `create_gaussian_diffusion` builds and the launchers
    use - the map is the identity.  Round 6: subsets are supported too (the fused hipGraph loop takes the map, `tamf_set_timestep_map`;
    the per-step path wraps the model like the reference)."""

    def __init__(self, use_timesteps, betas, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        base = GaussianDiffusion(betas=betas, **kwargs)
        last, new_betas = 1.0, []
        self.timestep_map = []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        if not new_betas:
            raise ValueError("use_timesteps keeps no timestep of the base process")
        super().__init__(betas=np.array(new_betas), **kwargs)

    @property
    def respaced(self) -> bool:
        return self.timestep_map != list(range(self.original_num_steps))

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel) or not self.respaced:
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def condition_mean(self, cond_fn, *args, **kwargs):  # (the guidance function sees the base process' timesteps too, :91-92)
        return super().condition_mean(self._wrap_model(cond_fn), *args, **kwargs)


class _WrappedModel:
    """model(x, ts) -> model(x, timestep_map[ts]) (reference :107-119; rescale_timesteps is never set by the factory)"""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = list(timestep_map)
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        if rescale_timesteps:
            raise NotImplementedError("rescale_timesteps is never set by create_gaussian_diffusion")

    def parameters(self):
        return self.model.parameters()


    def __call__(self, x, ts, **kwargs):
        import torch as th

        map_tensor = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        return self.model(x, map_tensor[ts], **kwargs)
