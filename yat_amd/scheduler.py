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
