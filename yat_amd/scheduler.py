"""Flow-matching schedule and timestep sampling for the SANA recipe (host side, tiny).

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
