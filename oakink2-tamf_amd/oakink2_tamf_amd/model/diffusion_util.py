"""create_gaussian_diffusion - the factory the launchers call (reference model/diffusion_util.py:5-31):
x0 prediction, fixed-small sigma, no respacing, no timestep rescaling."""
from .diffusion.gaussian_diffusion import get_named_beta_schedule
from .diffusion.respace import SpacedDiffusion, space_timesteps


def create_gaussian_diffusion(diffusion_steps, noise_schedule, sigma_small=True, timestep_respacing=""):
    """`timestep_respacing` is this build's one addition to the reference's signature: the reference hard-codes "" (every step,
    diffusion_util.py:10) although its SpacedDiffusion takes subsets; "" keeps its behaviour, "100" / "10,20,30" / "ddim50" are the
    section counts of `space_timesteps` (respace.py:8-57) - ancestral sampling over a strided subset of the trained timesteps."""
    if not sigma_small:
        raise NotImplementedError("fixed-large sigma is never selected by the reference launchers")
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps, 1.0)
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing or [diffusion_steps]), betas=betas,
                           rescale_timesteps=False)
