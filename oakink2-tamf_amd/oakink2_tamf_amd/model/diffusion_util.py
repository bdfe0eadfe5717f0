"""create_gaussian_diffusion - the factory the launchers call (reference model/diffusion_util.py:5-31):
x0 prediction, fixed-small sigma, no respacing, no timestep rescaling."""
from .diffusion.gaussian_diffusion import get_named_beta_schedule
from .diffusion.respace import SpacedDiffusion, space_timesteps


def create_gaussian_diffusion(diffusion_steps, noise_schedule, sigma_small=True):
    if not sigma_small:
        raise NotImplementedError("fixed-large sigma is never selected by the reference launchers")
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps, 1.0)
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, [diffusion_steps]), betas=betas,
                           rescale_timesteps=False)
