"""Timestep respacing (reference model/diffusion/respace.py:8-119).  The MF-MDM launchers always keep every
timestep, so the map is the identity and the model wrapper is a no-op; the betas are nevertheless re-derived
from the base process' cumulative alphas as the reference does (the float64 round-trip is part of parity)."""
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
    def __init__(self, use_timesteps, betas, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        base = GaussianDiffusion(betas=betas, **kwargs)
        if self.use_timesteps != set(range(self.original_num_steps)):
            raise NotImplementedError("timestep subsetting is not used by the MF-MDM launchers (timestep_respacing = [steps])")
        last, new_betas = 1.0, []
        self.timestep_map = []
        for i, ac in enumerate(base.alphas_cumprod):
            new_betas.append(1 - ac / last)
            last = ac
            self.timestep_map.append(i)
        super().__init__(betas=np.array(new_betas), **kwargs)
