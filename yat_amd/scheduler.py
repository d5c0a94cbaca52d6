"""Noise schedules and timestep sampling for the training recipes (host side, tiny): flow matching (SANA), DDPM (PixArt).

Restates what train_sana.py:41,185-204 takes from diffusers:
* FlowMatchEulerDiscreteScheduler tables [RECALL]: s_i = (1000 - i)/1000, sigma_i = shift*s_i/(1+(shift-1)*s_i),
  timesteps_i = 1000*sigma_i (fp32);
* compute_density_for_timestep_sampling('logit_normal', B, 0, 1) [RECALL]: u = sigmoid(N(0,1)) drawn on
  the CPU generator the trainer passes (common/trainer.py:325);
* get_sigmas (train_sana.py:195-204): the reference finds each timestep's index in the table with B
  device->host syncs; the index is already known (it was used to pick the timestep), so no lookup
  and no sync is needed here.
"""
from __future__ import annotations

import torch


class FlowMatchSchedule:
    def __init__(self, num_train_timesteps: int = 1000, shift: float = 3.0):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        s = torch.linspace(1.0, float(num_train_timesteps), num_train_timesteps, dtype=torch.float32).flip(0) \
            / num_train_timesteps
        self.sigmas = shift * s / (1 + (shift - 1) * s)
        self.timesteps = self.sigmas * num_train_timesteps
        self.config = type("Cfg", (), {"num_train_timesteps": num_train_timesteps, "shift": shift})()

    def sample(self, batch_size: int, generator: torch.Generator | None):
        """-> (indices int64 [B], timesteps f32 [B], sigmas bf16 [B]) on the CPU, reference draw order."""
        u = torch.sigmoid(torch.normal(mean=0.0, std=1.0, size=(batch_size,), device="cpu", generator=generator))
        idx = (u * self.num_train_timesteps).long()
        return idx, self.timesteps[idx], self.sigmas.to(torch.bfloat16)[idx]


class DDPMSchedule:
    """[RECALL diffusers DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, 'linear')] as
    train_pixart_sigma.py:37 loads it; used at :173-176: ``timesteps = scheduler.timesteps[indices]`` (int64, = 999 - index)
    and ``add_noise`` = sqrt(acp_t) x + sqrt(1 - acp_t) n with the table cast to the sample dtype (bf16) first."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear"):
        self.num_train_timesteps = num_train_timesteps
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":           # [RECALL] the SD1.5 scheduler config (train_sd15.py:30-31)
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}")
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        acp = self.alphas_cumprod.to(torch.bfloat16)
        self.sqrt_alpha_prod = acp ** 0.5                      # bf16 [1000], each op rounded as in add_noise
        self.sqrt_one_minus_alpha_prod = (1 - acp) ** 0.5
        self.config = type("Cfg", (), {"num_train_timesteps": num_train_timesteps, "beta_start": beta_start,
                                       "beta_end": beta_end, "beta_schedule": beta_schedule})()

    def sample(self, batch_size: int, generator: torch.Generator | None = None):
        """-> (timesteps int64 [B], sqrt_alpha_prod bf16 [B], sqrt_one_minus_alpha_prod bf16 [B]) on the CPU; the draw is
        compute_density_for_timestep_sampling('logit_normal', B, 0, 1) (:172), from the global RNG unless told otherwise."""
        u = torch.sigmoid(torch.normal(mean=0.0, std=1.0, size=(batch_size,), device="cpu", generator=generator))
        t = self.timesteps[(u * self.num_train_timesteps).long()]
        return t, self.sqrt_alpha_prod[t], self.sqrt_one_minus_alpha_prod[t]


class DPMSolverPP2M:
    """The scheduler behind the reference's PixArt-Sigma validation (train_pixart_sigma.py:117-129 calls ``self.pipe(...,
    guidance_scale=5.0, num_inference_steps=20, output_type='latent')``; the pipe's own scheduler is what steps): for the
    PixArt-Sigma checkpoints a ``DPMSolverMultistepScheduler`` [RECALL: scheduler_config.json -- algorithm_type
    'dpmsolver++', solver_order 2, solver_type 'midpoint', lower_order_final, linear betas 1e-4 .. 0.02 over 1000 steps,
    epsilon prediction, timestep_spacing 'linspace', final_sigmas_type 'zero', no Karras sigmas, no thresholding].  Restated
    [RECALL diffusers DPMSolverMultistepScheduler]:

    * ``set_timesteps(n)``: timesteps = round(linspace(0, 999, n + 1))[::-1][:-1] (int64); sigmas = sqrt((1 - acp) / acp)
      interpolated at them, a trailing 0;
    * per step: x0 = (x - sigma_t eps) / alpha_t with (alpha_t, sigma_t) = (1, sigma) / sqrt(sigma^2 + 1), evaluated in the
      model dtype (the sample is still bf16 there); then, on the fp32 sample, the first-order update
      ``x <- (sigma_t / sigma_s) x - alpha_t (exp(-h) - 1) x0`` for the first step and -- final sigma zero -- the last one,
      and the second-order multistep (midpoint) update with D1 = (x0 - x0_prev) / r0, r0 = h_prev / h, in between;
      h = lambda_t - lambda_s, lambda = log(alpha) - log(sigma); the result goes back to the model dtype."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02):
        import numpy as np
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        acp = torch.cumprod(1.0 - betas, dim=0)
        self._sigmas_train = (((1 - acp) / acp) ** 0.5).numpy()
        self._np = np
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        np = self._np
        ts = np.linspace(0, self.num_train_timesteps - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = np.interp(ts, np.arange(0, len(self._sigmas_train)), self._sigmas_train)
        self.timesteps = torch.from_numpy(ts)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self._x0_prev, self._i = None, 0
        return self.timesteps

    @staticmethod
    def _alpha_sigma(sigma: torch.Tensor):
        alpha_t = 1.0 / ((sigma ** 2 + 1.0) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step(self, eps: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        """One scheduler step: ``eps`` = the (guided, learned-sigma-stripped) model output, ``sample`` = the current latents,
        both in the model dtype.  The scalars stay 0-dim fp32 tensors, as in the scheduler: a dimensioned bf16 tensor times a
        0-dim fp32 tensor is a bf16 tensor (torch's promotion rule), the fp32 sample times one an fp32 tensor."""
        i, n = self._i, len(self.timesteps)
        a_s, s_s = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - s_s * eps) / a_s                                        # convert_model_output (before the upcast)
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        lam_t, lam_s = torch.log(a_t) - torch.log(s_t), torch.log(a_s) - torch.log(s_s)
        h = lam_t - lam_s
        x = sample.to(torch.float32)
        if self._x0_prev is None or i == n - 1:                                # first step; final sigma zero -> last step
            out = (s_t / s_s) * x - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            a_p, s_p = self._alpha_sigma(self.sigmas[i - 1])
            r0 = (lam_s - (torch.log(a_p) - torch.log(s_p))) / h
            d1 = (1.0 / r0) * (x0 - self._x0_prev)
            out = (s_t / s_s) * x - (a_t * (torch.exp(-h) - 1.0)) * x0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * d1
        self._x0_prev, self._i = x0, i + 1
        return out.to(eps.dtype)
