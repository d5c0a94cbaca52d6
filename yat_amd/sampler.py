"""Latent sampler for the validation pass (train_sana.py:99-161 calls ``self.pipe(..., guidance_scale=5.0,
num_inference_steps=20, output_type='latent')``): classifier-free guidance over the HIP transformer with a
flow-match Euler integrator.

What it restates [RECALL, diffusers SanaPipeline.__call__ / FlowMatchEulerDiscreteScheduler]:
* ``set_timesteps(n)``: n timesteps linearly spaced from t(sigma_max) to t(sigma_min) of the TRAINING table, divided by
  1000, pushed through the static shift once more, times 1000; sigmas get a trailing 0;
* initial latents: N(0,1) in the model dtype from the given generator (init_noise_sigma = 1);
* per step: the latents are duplicated (unconditional | conditional), one transformer call with the timestep expanded
  to the batch, ``v = v_u + g (v_c - v_u)``, Euler update ``x <- x + (sigma_next - sigma) v`` in fp32, cast back.
Text encoding (Gemma) and VAE decoding are outside the hot-path scope: the caller passes prompt embeddings and gets
latents back, exactly the middle third of the reference's validate().
"""
from __future__ import annotations

import torch

from .scheduler import FlowMatchSchedule

BF16 = torch.bfloat16


def inference_schedule(sched: FlowMatchSchedule, num_inference_steps: int):
    """-> (timesteps f32 [n], sigmas f32 [n+1])."""
    n_train = sched.num_train_timesteps
    smax, smin = float(sched.sigmas[0]), float(sched.sigmas[-1])
    ts = torch.linspace(smax * n_train, smin * n_train, num_inference_steps, dtype=torch.float32)
    sig = ts / n_train
    sig = sched.shift * sig / (1 + (sched.shift - 1) * sig)
    return sig * n_train, torch.cat([sig, torch.zeros(1)])


@torch.no_grad()
def sample_latents(model, prompt_embeds, prompt_mask, negative_embeds, negative_mask, height, width, *,
                   num_inference_steps=20, guidance_scale=5.0, generator=None, schedule: FlowMatchSchedule | None = None,
                   latents=None, callback_on_step_end=None):
    """prompt_embeds / negative_embeds: [B, T, C] bf16, masks [B, T] (1 keep); height, width in latent pixels.
    Returns latents [B, C_in, height, width] (bf16, on the model's device).  ``callback_on_step_end(pipe, step, timestep,
    callback_kwargs)`` is the diffusers pipelines' hook of the same name, called after every step (the trainer's
    ``validation_step_callback``, common/trainer.py:270-281; of the reference's entry points only train_sdxl.py:107 passes it)."""
    sched = schedule or FlowMatchSchedule()
    dev = model.device
    B = prompt_embeds.shape[0]
    cin = model.cfg.in_channels
    if latents is None:
        if generator is not None and generator.device.type == "cuda":
            latents = torch.randn(B, cin, height, width, generator=generator, device=dev, dtype=BF16)
        else:
            latents = torch.randn(B, cin, height, width, generator=generator, dtype=BF16).to(dev)
    latents = latents.to(device=dev, dtype=BF16)
    enc = torch.cat([negative_embeds, prompt_embeds]).to(device=dev, dtype=BF16)
    mask = torch.cat([negative_mask, prompt_mask]).to(dev)
    timesteps, sigmas = inference_schedule(sched, num_inference_steps)
    do_cfg = guidance_scale > 1.0
    for i in range(num_inference_steps):
        x_in = torch.cat([latents, latents]) if do_cfg else latents
        t = timesteps[i].expand(x_in.shape[0]).to(dev)
        v = model(x_in, encoder_hidden_states=enc if do_cfg else enc[B:], timestep=t,
                  encoder_attention_mask=mask if do_cfg else mask[B:]).sample
        if do_cfg:
            v_u, v_c = v.float().chunk(2)
            v = (v_u + guidance_scale * (v_c - v_u)).to(BF16)       # the pipeline combines in the model dtype
        latents = (latents.float() + (float(sigmas[i + 1]) - float(sigmas[i])) * v.float()).to(BF16)
        if callback_on_step_end is not None:
            callback_on_step_end(None, i, timesteps[i], {})
    return latents


@torch.no_grad()
def sample_latents_sd3(model, prompt_embeds, pooled, negative_embeds, negative_pooled, height, width, *,
                       num_inference_steps=20, guidance_scale=5.0, generator=None, schedule: FlowMatchSchedule | None = None,
                       latents=None):
    """The middle third of the reference's SD3.5 ``validate()`` (train_sd35.py:129-142: ``self.pipe(prompt_embeds=...,
    negative_prompt_embeds=..., pooled_prompt_embeds=..., negative_pooled_prompt_embeds=..., guidance_scale=5.0,
    num_inference_steps=20, generator=generator, output_type='latent')``) over the HIP MMDiT [RECALL,
    StableDiffusion3Pipeline.__call__]: the same classifier-free-guidance / flow-match Euler loop as ``sample_latents``, the
    condition being (token embeddings [B, T, C], pooled projection [B, P]) pairs and no mask; the scheduler's static shift
    (3.0 for SD3.5) comes with ``schedule``.  Returns latents [B, C_in, height, width] (bf16): ``output_type='latent'``
    returns them as the loop leaves them (the VAE's scaling / shift belong to the decode)."""
    sched = schedule or FlowMatchSchedule(shift=3.0)
    dev = model.device
    B = prompt_embeds.shape[0]
    cin = model.cfg.in_channels
    if latents is None:
        if generator is not None and generator.device.type == "cuda":
            latents = torch.randn(B, cin, height, width, generator=generator, device=dev, dtype=BF16)
        else:
            latents = torch.randn(B, cin, height, width, generator=generator, dtype=BF16).to(dev)
    latents = latents.to(device=dev, dtype=BF16)
    do_cfg = guidance_scale > 1.0
    enc = (torch.cat([negative_embeds, prompt_embeds]) if do_cfg else prompt_embeds).to(device=dev, dtype=BF16)
    pool = (torch.cat([negative_pooled, pooled]) if do_cfg else pooled).to(device=dev, dtype=BF16)
    timesteps, sigmas = inference_schedule(sched, num_inference_steps)
    for i in range(num_inference_steps):
        x_in = torch.cat([latents, latents]) if do_cfg else latents
        t = timesteps[i].expand(x_in.shape[0]).to(dev)
        v = model(x_in, encoder_hidden_states=enc, pooled_projections=pool, timestep=t).sample
        if do_cfg:
            v_u, v_c = v.float().chunk(2)
            v = (v_u + guidance_scale * (v_c - v_u)).to(BF16)       # the pipeline combines in the model dtype
        latents = (latents.float() + (float(sigmas[i + 1]) - float(sigmas[i])) * v.float()).to(BF16)
    return latents


@torch.no_grad()
def sample_latents_pixart(model, prompt_embeds, prompt_mask, negative_embeds, negative_mask, height, width, *,
                          num_inference_steps=20, guidance_scale=5.0, generator=None, solver=None, latents=None):
    """The middle third of the reference's PixArt-Sigma ``validate()`` (train_pixart_sigma.py:117-129) over the HIP transformer:
    the denoising loop of the pipeline as the reference vendors it (utils/patch_pixart_sigma_pipeline.py:158-208 -- CFG batch
    (unconditional | conditional), ``scale_model_input`` = identity, integer timestep expanded to the batch, no micro-conditions,
    ``noise_pred_uncond + g (noise_pred_text - noise_pred_uncond)``, the learned-sigma half dropped with ``.chunk(2, dim=1)[0]``,
    ``scheduler.step``) with the pipe's DPM-Solver++ (2M) scheduler (``yat_amd.scheduler.DPMSolverPP2M`` [RECALL]).
    ``pag_scale=2.0`` in the reference's call lands in the plain pipeline's ``**kwargs`` and is ignored (:67), so there is no
    perturbed-attention pass.  height, width in latent pixels; returns latents [B, C_in, height, width] (bf16)."""
    from .scheduler import DPMSolverPP2M
    solver = solver or DPMSolverPP2M()
    dev = model.device
    B = prompt_embeds.shape[0]
    cin = model.cfg.in_channels
    if latents is None:
        if generator is not None and generator.device.type == "cuda":
            latents = torch.randn(B, cin, height, width, generator=generator, device=dev, dtype=BF16)
        else:
            latents = torch.randn(B, cin, height, width, generator=generator, dtype=BF16).to(dev)
    latents = latents.to(device=dev, dtype=BF16) * solver.init_noise_sigma
    do_cfg = guidance_scale > 1.0
    enc = (torch.cat([negative_embeds, prompt_embeds]) if do_cfg else prompt_embeds).to(device=dev, dtype=BF16)
    mask = (torch.cat([negative_mask, prompt_mask]) if do_cfg else prompt_mask).to(dev)
    timesteps = solver.set_timesteps(num_inference_steps)
    solver.sigmas = solver.sigmas.to(dev)
    learned_sigma = model.cfg.out_channels // 2 == cin
    for i in range(num_inference_steps):
        x_in = torch.cat([latents, latents]) if do_cfg else latents
        t = timesteps[i].expand(x_in.shape[0]).to(dev)
        eps = model(x_in, encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask).sample
        if do_cfg:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)              # in the model dtype, as the pipeline combines
        if learned_sigma:
            eps = eps.chunk(2, dim=1)[0]
        latents = solver.step(eps.contiguous(), latents)
    return latents
