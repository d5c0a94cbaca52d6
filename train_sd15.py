#!/usr/bin/env python3
"""SD1.5 trainer entry point (`train_sd15.py --config config.yaml`, reference train_sd15.py:167-182) -- BASELINE config 1, the
reference's CPU plumbing run: config -> shards -> bucket sampler -> ``optimize`` (DDPM epsilon-prediction, :140-165) ->
backward -> clip(1.0) -> AdamW, a few steps at batch 1 on the host.

Scope: *plumbing only*.  The UNet is diffusers' ``UNet2DConditionModel``, which is
neither vendored in the reference nor installed here, and the reference defines no GPU work for this config -- so there is no
HIP UNet.  The entry point loads the UNet from a LOCAL diffusers-layout directory (``pretrained_model_path``, or
``<pretrained_pipe_path>/unet``: ``config.json`` + ``diffusion_pytorch_model.safetensors``, train_sd15.py:20-28) into
yat_amd/sd15.py's host module; ``SD15Model`` also accepts any ``nn.Module`` with the call contract ``model(noisy, timestep,
encoder_hidden_states)`` -> sample.  The step runs through the SAME trainer loop, sampler, shard format and YAML reader as the
HIP recipes; only the arithmetic is stock torch on the host (``torch.optim.AdamW`` + ``clip_grad_norm_``, which is literally
what the reference calls, common/trainer.py:246-248,347-348).  On a GPU device the model refuses to run: a CPU arithmetic path
next to the HIP kernels is not something this build ships, so ``python train_sd15.py --config ...`` pins the run to the host.

Reference quirks handled: ``optimize(self, ratio, latents, embeddings)`` (:140) has three of the five arguments
``Model.run`` passes (common/trainer.py:337) -- TypeError at HEAD; the 512 px aspect table is hard-coded (:36), BASELINE's
"256 px" is expressed through ``aspect_ratio: 256``-sized latents in the shards (the bucket table only keys the ratios).
"""
import argparse

import torch
import torch.nn.functional as F

from yat_amd.common.aspect_ratios import ASPECT_RATIO_512_BIN
from yat_amd.common.trainer import Model
from yat_amd.common.training_parameters_reader import TrainingParameters
from yat_amd.scheduler import DDPMSchedule


class _TorchClipAdamW:
    """clip_grad_norm_(1.0) -> AdamW.step() -> zero_grad() (common/trainer.py:347-356) for a plain nn.Module on the host."""

    def __init__(self, module, lr, weight_decay):
        self.model = module
        self.opt = torch.optim.AdamW(module.parameters(), lr=lr, weight_decay=weight_decay)
        self.param_groups = self.opt.param_groups
        for g in self.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.ema_shadow = None

    def step(self):
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=1.0)
        self.opt.step()
        self.opt.zero_grad()


class SD15Model(Model):
    def __init__(self, params: TrainingParameters, accelerator=None, unet: torch.nn.Module | None = None):
        super().__init__(params, accelerator)
        if self.accelerator.device.type != "cpu":
            raise NotImplementedError("train_sd15.py is BASELINE's CPU plumbing config: the SD1.5 UNet has no HIP path in this "
                                      "build (SURVEY.md 8 row a24); run it on the host")
        if unet is None:                                                       # :20-28
            import os
            from yat_amd.sd15 import UNet2DConditionCPU
            path = params.pretrained_model_path
            if path is None and params.pretrained_pipe_path is not None:
                path = os.path.join(params.pretrained_pipe_path, "unet")      # the pipeline directory's own UNet
            if path is None:
                raise NotImplementedError("no UNet: set pretrained_model_path (a local diffusers-layout UNet directory) or "
                                          "pretrained_pipe_path; single-file checkpoints and hub names are not supported offline")
            unet = UNet2DConditionCPU.from_pretrained(path)
        self.model = unet.to(torch.bfloat16)                                   # :39
        self.scheduler = DDPMSchedule(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")   # :30-31 [RECALL]
        self.aspect_ratios = ASPECT_RATIO_512_BIN                              # :36
        self.model.enable_gradient_checkpointing()                             # :44
        self.pipe = None

    def make_optimizer(self, trained):
        return _TorchClipAdamW(trained, self.params.learning_rate, self.params.weight_decay)

    def extract_latents(self, images):
        raise NotImplementedError("VAE encoding is outside the hot-path scope; train from cached-feature shards")

    def extract_embeddings(self, captions):
        raise NotImplementedError("text encoding is outside the hot-path scope; train from cached-feature shards")

    def validate(self):
        raise NotImplementedError("the SD1.5 validation pipeline (CLIP + VAE decode) is outside the hot-path scope")

    def save_model(self):
        import os
        if hasattr(self.model, "save_pretrained"):                             # common/trainer.py:295-296
            self.model.save_pretrained(f"models/{self.global_step}")
            return
        from safetensors.torch import save_file
        os.makedirs(f"models/{self.global_step}", exist_ok=True)
        save_file({k: v.detach().contiguous() for k, v in self.model.state_dict().items()},
                  f"models/{self.global_step}/diffusion_pytorch_model.safetensors")

    def _validate_and_save(self):
        if self.accelerator.is_main_process:
            self.save_model()

    def optimize(self, ratio, latents, embeddings, repa_tokens=None, generator: torch.Generator = None):
        """train_sd15.py:140-165; noise and timestep draws from the global RNGs as there (:149,151)."""
        dev = self.accelerator.device
        emb = torch.stack(embeddings).squeeze(1).to(device=dev, dtype=torch.bfloat16)        # :145
        latents = latents.to(device=dev, dtype=torch.bfloat16)                               # :146
        noise = torch.randn(latents.shape, device=dev, dtype=torch.bfloat16)                 # :149
        t, a, c = self.scheduler.sample(latents.shape[0], None)                              # :151-153
        noisy = a.view(-1, 1, 1, 1) * latents + c.view(-1, 1, 1, 1) * noise                  # :154 add_noise, bf16 op by op
        pred = self.model(noisy, t, emb)                                                     # :157-161
        return F.mse_loss(pred.float(), noise.float())                                       # :163-164


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", required=True, type=str)
    args = parser.parse_args()
    params = TrainingParameters()
    params.read_yaml(args.config)
    if params.extract_features:                                                # :175-181 needs the VAE and the text encoder
        raise SystemExit("extract_features: VAE / text encoding are outside the hot-path scope; train from cached-feature shards")
    from yat_amd.common.trainer import HipAccelerator
    # BASELINE config 1 is defined on the host ("10 steps on CPU ... no GPU"): the accelerator is pinned to the CPU
    trainer = SD15Model(params, accelerator=HipAccelerator(params.gradient_accumulation_steps, device="cpu"))
    trainer.run()
    print(f"train_sd15: {trainer.global_step} steps, last losses "
          f"{[round(float(l), 4) for l in list(trainer.loss_history)[-3:]]}")
