#!/usr/bin/env python3
"""SANA trainer entry point -- same CLI as the reference (`train_sana.py --config config.yaml`, README.md:45,
train_sana.py:221-237), driving the MI355X-native hot path.

    python train_sana.py --config config.yaml                      # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_sana.py --config config.yaml

``pretrained_model_path`` (or ``pretrained_pipe_path``/transformer) must be a LOCAL diffusers directory
(config.json + diffusion_pytorch_model.safetensors); with neither present the 1.6B architecture is random-initialised
(there is no network here).  Feature extraction (VAE / text encoder, ``extract_features`` / ``compute_features``) is outside the hot-path scope:
training consumes cached-feature shards; validation samples latents from cached prompt embeddings (no VAE decode).
"""
import argparse
import json
import os

import torch

from yat_amd.common.training_parameters_reader import TrainingParameters
from yat_amd.common.trainer import Model
from yat_amd.common.aspect_ratios import table_for_resolution
from yat_amd.recipe import SanaRecipe
from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
from yat_amd.scheduler import FlowMatchSchedule


class SanaModel(Model):
    def __init__(self, params: TrainingParameters, accelerator=None, config: SanaConfig | None = None):
        super().__init__(params, accelerator)
        dev = self.accelerator.device
        path = params.pretrained_model_path
        if path is None and params.pretrained_pipe_path and os.path.isdir(os.path.join(params.pretrained_pipe_path, "transformer")):
            path = os.path.join(params.pretrained_pipe_path, "transformer")
        if path is not None and os.path.isdir(path):
            self.model = SanaTransformer2DModelHIP.from_pretrained(path, device=dev)          # train_sana.py:20-23
        else:
            self.model = SanaTransformer2DModelHIP(config or SanaConfig(), device=dev).init_synthetic(0)
        shift = 3.0
        sched_cfg = os.path.join(params.pretrained_pipe_path or "", "scheduler", "scheduler_config.json")
        if os.path.isfile(sched_cfg):                                                          # :41
            with open(sched_cfg) as f:
                shift = float(json.load(f).get("shift", shift))
        self.scheduler = FlowMatchSchedule(shift=shift)
        vae_compression = 32                                                                   # :45-57
        self.aspect_ratios = table_for_resolution(self.model.config.sample_size * vae_compression)
        self.model.enable_gradient_checkpointing()                                             # :63 (no-op here)
        self.recipe = SanaRecipe(self.model, self.scheduler, pad_to=512, device=dev)
        self.pipe = None

    def extract_latents(self, images):
        raise NotImplementedError("VAE encoding is outside the hot-path scope; train from cached-feature shards")

    def extract_embeddings(self, captions):
        raise NotImplementedError("text encoding is outside the hot-path scope; train from cached-feature shards")

    def validate(self):
        """Middle third of train_sana.py:99-161: 20-step flow-match Euler sampling with CFG 5.0 over the HIP transformer,
        generator seeded 42 (:108).  The text encoder and the VAE are outside this build's scope, so the prompt embeddings
        come from a cached file (``validation_embeds.pt`` next to the shards or in the cwd: a list of
        (prompt_embeds [1,T,C], mask [1,T], negative_embeds, negative_mask) tuples as ``pipe.encode_prompt`` returns them)
        and the result is the latents (``output_type='latent'``), stored under models/<step>/."""
        from yat_amd.sampler import sample_latents
        cands = [os.path.join(os.path.dirname(p), "validation_embeds.pt") for p in (self.params.local_shard_paths or [])]
        path = next((c for c in cands + ["validation_embeds.pt"] if os.path.isfile(c)), None)
        if path is None:
            raise NotImplementedError("no cached validation embeddings (text encoding is outside the hot-path scope)")
        embeds = torch.load(path, map_location="cpu")
        gen = torch.Generator(device=self.accelerator.device).manual_seed(42)
        side = self.model.config.sample_size
        out = []
        for pe, pm, ne, nm in embeds:
            out.append(sample_latents(self.model, pe, pm, ne, nm, side, side, num_inference_steps=20, guidance_scale=5.0,
                                      generator=gen, schedule=self.scheduler).cpu())
        os.makedirs(f"models/{self.global_step}", exist_ok=True)
        torch.save(out, f"models/{self.global_step}/validation_latents.pt")
        if self.logger is not None:                # :157 logs the decoded image; without the VAE: a latent preview
            for idx, lat in enumerate(out):
                x = lat[0, :3].float()
                x = (x - x.amin()) / (x.amax() - x.amin()).clamp_min(1e-6)
                self.logger.add_image(f"validation_latents/{idx}", x, self.global_step)
        return out

    def optimize(self, ratio, latents, embeddings, repa_tokens, generator: torch.Generator = None):
        """train_sana.py:163-219 on the HIP path.  With gradients enabled (the training call, common/trainer.py:337) the step
        runs on the allocation-free device path -- one packed H2D copy, forward, loss and backward as straight-line launches
        (yat_amd.recipe.SanaRecipe.optimize_device) -- and the returned loss is marked so that ``accelerator.backward`` does
        not run a second backward; under ``no_grad`` (exploration trials, :326-336) it is the plain forward + loss."""
        if torch.is_grad_enabled() and not latents.is_cuda and os.environ.get("YAT_TRAINER_FAST", "1") != "0":
            loss = self.recipe.optimize_device(latents, embeddings, generator,
                                               gscale=1.0 / self.accelerator.gradient_accumulation_steps)
            loss.yat_backward_done = True
            return loss
        return self.recipe.optimize(latents, embeddings, generator)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", required=True, type=str)
    parser.add_argument("--max-steps", type=int, default=None)
    args = parser.parse_args()
    params = TrainingParameters()
    params.read_yaml(args.config)
    if params.extract_features:
        raise SystemExit("extract_features (VAE/text-encoder feature extraction) is outside this build's scope")
    trainer = SanaModel(params)
    trainer.run(max_steps=args.max_steps)
