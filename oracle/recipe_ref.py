"""Torch-CPU restatement of the SANA training recipe and optimizer step (oracle, test-only).

Follows:
* SanaModel.optimize                 /root/reference/train_sana.py:163-219
* step loop (clip, AdamW, zero_grad) /root/reference/common/trainer.py:246-248,337-356
* flow-match scheduler tables        [RECALL] diffusers FlowMatchEulerDiscreteScheduler.__init__
                                     (used at train_sana.py:41,192-197)
* timestep density                   [RECALL] diffusers compute_density_for_timestep_sampling
                                     ('logit_normal'; used at train_sana.py:185-191)
* noise                              [RECALL] diffusers randn_tensor: CPU generator -> draw on CPU
                                     in the requested dtype, then move (train_sana.py:183)

``torch.optim.AdamW`` and ``torch.nn.utils.clip_grad_norm_`` are the reference's own
dependency (stock torch) run on CPU, so that part of the oracle is the reference code itself.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class FlowMatchSchedule:
    """[RECALL] FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift) tables."""

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 3.0):
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        ts = torch.from_numpy(ts).to(torch.float32)
        sigmas = ts / num_train_timesteps
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.sigmas = sigmas                      # [1000] f32, sigma_0 = 1.0
        self.timesteps = sigmas * num_train_timesteps


def logit_normal_u(batch_size: int, generator: torch.Generator | None) -> torch.Tensor:
    """[RECALL] compute_density_for_timestep_sampling('logit_normal', B, 0, 1.0) (mode_scale unused)."""
    u = torch.normal(mean=0.0, std=1.0, size=(batch_size,), device="cpu", generator=generator)
    return torch.sigmoid(u)


def pad_embeddings(embeddings, pad_to: int = 512):
    """train_sana.py:168-180: zero-pad each [L_i, C] to pad_to rows, int64 mask with ones on the first L_i."""
    padded, masks = [], []
    for emb in embeddings:
        padded.append(F.pad(emb, pad=(0, 0, 0, pad_to - emb.shape[0]), mode="constant", value=0))
        m = torch.zeros(pad_to, dtype=torch.long)
        m[: emb.shape[0]] = 1
        masks.append(m)
    return torch.stack(padded), torch.stack(masks)


def draw_recipe_randoms(shape, batch_size, sched: FlowMatchSchedule, generator, dtype=torch.bfloat16):
    """The random draws of one optimize() call in the reference's order: noise first
    (train_sana.py:183), then u (:185-191) -> indices (:192) -> timesteps (:193) -> sigmas (:195-204)."""
    noise = torch.randn(shape, generator=generator, device="cpu", dtype=dtype)
    u = logit_normal_u(batch_size, generator)
    indices = (u * sched.num_train_timesteps).long()
    timesteps = sched.timesteps[indices]
    # get_sigmas: index of each t in the table by equality, then sigmas[idx] cast to the latent dtype
    step_indices = [(sched.timesteps == t).nonzero().item() for t in timesteps]
    sigmas = sched.sigmas.to(dtype=dtype)[step_indices].flatten()
    return noise, indices, timesteps, sigmas


def optimize_ref(model, sched: FlowMatchSchedule, latents, embeddings, generator, pad_to: int = 512,
                 dtype=torch.bfloat16, taps=None):
    """train_sana.py:163-219 -> (loss, noise_pred, target).  ``model`` is a SanaTransformerRef in ``dtype``.
    With dtype=float32 the same draws (made in bf16, as the reference does) are up-cast, giving the
    fp32 ground truth for identical inputs."""
    B = latents.shape[0]
    enc, mask = pad_embeddings(embeddings, pad_to)
    enc = enc.to(dtype)
    latents = latents.to(torch.bfloat16)
    noise, _, timesteps, sigmas = draw_recipe_randoms(latents.shape, B, sched, generator, torch.bfloat16)
    latents, noise, sigmas = latents.to(dtype), noise.to(dtype), sigmas.to(dtype)
    while sigmas.ndim < latents.ndim:
        sigmas = sigmas.unsqueeze(-1)
    noisy = (1.0 - sigmas) * latents + sigmas * noise                       # :207
    pred = model(noisy, enc, timesteps, encoder_attention_mask=mask, taps=taps)   # :210-215
    target = noise - latents                                               # :217
    loss = F.mse_loss(pred.float(), target.float())                        # :218
    return loss, pred, target


def clip_and_adamw_step(params, optimizer: torch.optim.AdamW, max_norm: float = 1.0):
    """trainer.py:347-348,356: clip_grad_norm_(1.0) -> optimizer.step() -> zero_grad()."""
    total = torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], max_norm=max_norm)
    optimizer.step()
    optimizer.zero_grad()
    return total


def inference_schedule_ref(sched: FlowMatchSchedule, num_inference_steps: int):
    """[RECALL] FlowMatchEulerDiscreteScheduler.set_timesteps(n) without dynamic shifting: linspace over
    t(sigma_max)..t(sigma_min) of the training table, /1000, static shift applied again, *1000; sigmas + [0]."""
    n = sched.num_train_timesteps
    ts = np.linspace(float(sched.sigmas[0]) * n, float(sched.sigmas[-1]) * n, num_inference_steps, dtype=np.float32)
    sig = torch.from_numpy(ts) / n
    sig = sched.shift * sig / (1 + (sched.shift - 1) * sig)
    return sig * n, torch.cat([sig, torch.zeros(1)])


@torch.no_grad()
def sample_latents_ref(model, sched: FlowMatchSchedule, latents, prompt_embeds, prompt_mask, negative_embeds, negative_mask,
                       num_inference_steps=20, guidance_scale=5.0, dtype=torch.bfloat16):
    """[RECALL] SanaPipeline.__call__ denoising loop (train_sana.py:135-147 calls it with guidance 5.0, 20 steps,
    output_type='latent'): CFG batch (uncond | cond), Euler flow-match step.  ``model`` is a SanaTransformerRef in
    ``dtype``; dtype=float32 gives the ground truth for the same initial latents."""
    timesteps, sigmas = inference_schedule_ref(sched, num_inference_steps)
    x = latents.to(dtype)
    enc = torch.cat([negative_embeds, prompt_embeds]).to(dtype)
    mask = torch.cat([negative_mask, prompt_mask])
    for i in range(num_inference_steps):
        x_in = torch.cat([x, x])
        t = timesteps[i].expand(x_in.shape[0])
        v = model(x_in, enc, t, encoder_attention_mask=mask)
        v_u, v_c = v.float().chunk(2)
        v = (v_u + guidance_scale * (v_c - v_u)).to(dtype)
        x = (x.float() + (float(sigmas[i + 1]) - float(sigmas[i])) * v.float()).to(dtype)
    return x


class EMAModelRef:
    """[RECALL] diffusers.training_utils.EMAModel(parameters, decay=0.999) as the reference constructs and steps it
    (common/trainer.py:266-268,350-351): defaults min_decay=0, update_after_step=0, use_ema_warmup=False.  Shadows are
    clones of the parameters (bf16 for a bf16 model), updated in place in that dtype."""

    def __init__(self, parameters, decay: float = 0.999, min_decay: float = 0.0, update_after_step: int = 0):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.optimization_step = 0
        self.cur_decay_value = None

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = (1 + step) / (10 + step)                      # use_ema_warmup=False branch
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters):
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        one_minus_decay = 1 - decay
        for s_param, param in zip(self.shadow_params, list(parameters)):
            if param.requires_grad:
                s_param.sub_(one_minus_decay * (s_param - param))
            else:
                s_param.copy_(param)
