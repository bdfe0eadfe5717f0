"""Gaussian diffusion schedule + DDPM ancestral sampler for the MF-MDM denoiser (x0-prediction, fixed-small
variance), host side.

Mirrors the call surface of the reference's model/diffusion/gaussian_diffusion.py for the branch sample.sh
reaches (START_X / FIXED_SMALL / p_sample*, reference lines 116-161, 209-229, 231-320, 412-460, 506-640);
DDIM / PLMS / VLB / training losses are out of scope.  The schedule stays float64
numpy on the host exactly like the reference.  The launchers' call (clip_denoised=False, no denoised_fn / init_image /
skip) takes the fused path: the whole loop is one library call, the per-step update fused into the output-head GEMM inside
the hipGraph loop.  Any other combination takes the generic per-step path below: model.forward on the HIP library,
q_posterior / p_sample in torch on the device, line for line as the reference (the standalone tamf_ddpm_step entry point
of the C-ABI is the same update for callers without torch).
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Callable, Optional

import numpy as np
import torch as th


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int, scale_betas: float = 1.0) -> np.ndarray:
    """"linear" (Ho et al.) and "cosine" (Nichol & Dhariwal) schedules - reference :20-42."""
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps: int, alpha_bar: Callable[[float], float], max_beta: float = 0.999) -> np.ndarray:
    """beta_i = min(1 - abar((i+1)/N) / abar(i/N), max_beta) - reference :45-62."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def _extract(arr: np.ndarray, t: th.Tensor, shape) -> th.Tensor:
    """float64 table -> per-sample float32 coefficients broadcast to `shape` (reference :1265-1278)."""
    res = th.from_numpy(arr).to(device=t.device)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


class GaussianDiffusion:
    """x0-predicting, fixed-small-variance diffusion (the configuration model/diffusion_util.py fixes)."""

    def __init__(self, *, betas, rescale_timesteps: bool = False):
        if rescale_timesteps:
            raise NotImplementedError("rescale_timesteps is never enabled by the reference launchers")
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        self.rescale_timesteps = False
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    # ---- forward process (only what init_image / skip_timesteps need) ----------------------------------
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = th.randn_like(x_start)
        return (_extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        mean = (_extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, _extract(self.posterior_variance, t, x_t.shape), _extract(self.posterior_log_variance_clipped, t, x_t.shape)

    # ---- reverse process, generic (per-step) path ---------------------------------------------------------
    def _scale_timesteps(self, t):
        return t

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        model_kwargs = model_kwargs or {}
        assert t.shape == (x.shape[0],)
        pred_xstart = model(x, self._scale_timesteps(t), **model_kwargs)
        if denoised_fn is not None:
            pred_xstart = denoised_fn(pred_xstart)
        if clip_denoised:
            pred_xstart = pred_xstart.clamp(-1, 1)
        mean, var, logvar = self.q_posterior_mean_variance(pred_xstart, x, t)
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": pred_xstart}

    def condition_mean(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """mean of the previous step under a guidance term: cond_fn(x, t, **model_kwargs) = grad log p(y | x) (reference :346-357,
        Sohl-Dickstein et al. 2015): new mean = mean + variance * gradient.  Never passed by the launchers; a per-step Python hook,
        so the loop runs on the per-step path (HIP forward + torch update)."""
        gradient = cond_fn(x, self._scale_timesteps(t), **(model_kwargs or {}))
        return p_mean_var["mean"].float() + p_mean_var["variance"] * gradient.float()

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, const_noise=False):
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = th.randn_like(x)
        if const_noise:
            noise = noise[[0]].repeat(x.shape[0], 1, 1, 1)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        if cond_fn is not None:
            out["mean"] = self.condition_mean(cond_fn, out, x, t, model_kwargs=model_kwargs)
        sample = out["mean"] + nonzero_mask * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False, const_noise=False):
        if cond_fn_with_grad:
            raise NotImplementedError("cond_fn_with_grad is not used by the MF-MDM sampling path")
        if randomize_class and model_kwargs and "y" in model_kwargs:
            raise NotImplementedError("class-conditional sampling is not part of MF-MDM")
        if device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else th.randn(*shape, device=device)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            my_t = th.ones([shape[0]], device=device, dtype=th.long) * indices[0]
            img = self.q_sample(init_image, my_t, img)
        if progress:
            from tqdm.auto import tqdm

            indices = tqdm(indices)
        for i in indices:
            t = th.full((shape[0],), i, device=device, dtype=th.long)
            with th.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                    model_kwargs=model_kwargs, const_noise=const_noise)
                yield out
                img = out["sample"]

    # ---- reverse process, entry point ----------------------------------------------------------------------
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                      randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                      noise_source: str = "philox", seed: Optional[int] = None, clip_id_base: int = 0):
        """Same signature and return value as the reference (:506-571), plus three keyword-only extras:

        noise_source  "philox"    (default) device Philox stream keyed (seed, global clip id, draw, element);
                                  seed defaults to torch.initial_seed(), so torch.manual_seed() controls it
                      "torch"     th.randn / th.randn_like on `device` in the reference's call order
                      "torch_cpu" the same draws from the torch CPU generator (bit-identical to a reference CPU run
                                  with the same seed), uploaded
        When the model is this package's HIP InterationSegmentMDM and no per-step Python hook is requested
        (clip_denoised=False, no denoised_fn / cond_fn / init_image / skip / const_noise) the whole loop runs
        as one hipGraph-replayed library call; otherwise it falls back to the per-step generic path above."""
        fused_ok = (getattr(model, "supports_fused_loop", False) and not clip_denoised and denoised_fn is None
                    and cond_fn is None and not skip_timesteps and init_image is None and not const_noise
                    and not randomize_class and not cond_fn_with_grad)
        if fused_ok:
            return model.fused_sample_loop(self, tuple(shape), x_T=noise, batch=(model_kwargs or {}).get("batch"),
                                           dump_steps=dump_steps, noise_source=noise_source, seed=seed,
                                           clip_id_base=clip_id_base, device=device)
        final = None
        dump = [] if dump_steps is not None else None
        for i, sample in enumerate(self.p_sample_loop_progressive(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, device=device, progress=progress, skip_timesteps=skip_timesteps,
                init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad,
                const_noise=const_noise)):
            if dump is not None and i in dump_steps:
                dump.append(deepcopy(sample["sample"]))
            final = sample
        return dump if dump is not None else final["sample"]

    # everything below is deliberately absent from the MI355X path
    def ddim_sample_loop(self, *a, **k):
        raise NotImplementedError("DDIM sampling is never reached by sample.sh (SURVEY.md section 2, row 2)")

    def plms_sample_loop(self, *a, **k):
        raise NotImplementedError("PLMS sampling is never reached by sample.sh (SURVEY.md section 2, row 2)")

    def training_losses(self, *a, **k):
        raise NotImplementedError("training is out of scope of the MI355X sampling path")
