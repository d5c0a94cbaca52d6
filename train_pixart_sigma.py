#!/usr/bin/env python3
"""PixArt-Sigma trainer entry point -- same CLI as the reference (`train_pixart_sigma.py --config config.yaml`,
train_pixart_sigma.py:187-198), driving the MI355X-native path (BASELINE config 3).

    python train_pixart_sigma.py --config config.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_pixart_sigma.py --config config.yaml

``pretrained_model_path`` (or ``pretrained_pipe_path``/transformer) must be a LOCAL diffusers directory; with neither the
PixArt-Sigma-XL-2 architecture is random-initialised (no network here).  VAE / T5 feature extraction and the validation
pipeline (PAG + DPM-Solver sampling + VAE decode, :78-149) are outside the hot-path scope: training consumes cached-feature
shards and ``validate()`` is a no-op that leaves the checkpoint cadence intact.

Reference quirk: ``PixartSigmaTrainer.optimize(self, latents, embeddings)`` (:151) still has the two-argument signature
while ``Model.run`` calls ``optimize(ratio, latents, embeddings, repa_features, generator)`` (common/trainer.py:337) -- at
HEAD the reference's PixArt entry point raises TypeError on the first step.  Here the recipe body is the one written at
:151-185 and the signature is the trainer's.
"""
import argparse
import json
import os

import torch

from yat_amd.common.training_parameters_reader import TrainingParameters
from yat_amd.common.trainer import Model
from yat_amd.common.aspect_ratios import table_for_resolution
from yat_amd.pixart import PixArtConfig, PixArtTransformer2DModelHIP
from yat_amd.recipe import PixArtRecipe
from yat_amd.scheduler import DDPMSchedule


class PixartSigmaTrainer(Model):
    def __init__(self, params: TrainingParameters, accelerator=None, config: PixArtConfig | None = None):
        super().__init__(params, accelerator)
        if getattr(params, "use_repa", False):
            # REPAPixArtTransformerModel (:26,31) only adds a projector whose output never reaches the loss
            # (common/trainer.py:340-341 is commented out): nothing to train there
            print("[Warning] use_repa: the REPA projector is not built (its loss term is disabled in the reference)")
        dev = self.accelerator.device
        path = params.pretrained_model_path
        if path is None and params.pretrained_pipe_path and os.path.isdir(os.path.join(params.pretrained_pipe_path, "transformer")):
            path = os.path.join(params.pretrained_pipe_path, "transformer")
        if path is not None and os.path.isdir(path):
            self.model = PixArtTransformer2DModelHIP.from_pretrained(path, device=dev)       # :24-33
        else:
            self.model = PixArtTransformer2DModelHIP(config or PixArtConfig(), device=dev).init_synthetic(0)
        self.model.enable_gradient_checkpointing()                                            # :34 (no-op here)
        kw = {}
        sched_cfg = os.path.join(params.pretrained_pipe_path or "", "scheduler", "scheduler_config.json")
        if os.path.isfile(sched_cfg):                                                         # :37
            with open(sched_cfg) as f:
                raw = json.load(f)
            if raw.get("beta_schedule", "linear") != "linear":
                raise NotImplementedError(f"beta_schedule {raw['beta_schedule']!r}")
            kw = {k: raw[k] for k in ("num_train_timesteps", "beta_start", "beta_end") if k in raw}
        self.scheduler = DDPMSchedule(**kw)
        vae_compression = 8                                                                   # :41-50
        self.aspect_ratios = table_for_resolution(self.model.config.sample_size * vae_compression)
        self.recipe = PixArtRecipe(self.model, self.scheduler, pad_to=300, device=dev)
        self.pipe = None

    def extract_latents(self, images):
        raise NotImplementedError("VAE encoding is outside the hot-path scope; train from cached-feature shards")

    def extract_embeddings(self, captions):
        raise NotImplementedError("text encoding is outside the hot-path scope; train from cached-feature shards")

    def validate(self):
        """Middle third of train_pixart_sigma.py:76-149: the 20-step DPM-Solver++ sampling with CFG 5.0 over the HIP transformer
        (:117-129; ``pag_scale`` lands in the plain pipeline's ``**kwargs`` and is ignored), generator seeded 42 on the device
        (:94).  The T5 encoder and the VAE are outside this build's scope, so the prompt embeddings come from a cached file
        (``validation_embeds.pt`` next to the shards or in the cwd: a list of (prompt_embeds [1,T,C], mask [1,T],
        negative_embeds, negative_mask) tuples as ``pipe.encode_prompt`` returns them, :100-108) and the result is the latents
        (``output_type='latent'``), stored under models/<step>/ with a three-channel preview for the logger (:143)."""
        from yat_amd.sampler import sample_latents_pixart
        cands = [os.path.join(os.path.dirname(p), "validation_embeds.pt") for p in (self.params.local_shard_paths or [])]
        path = next((c for c in cands + ["validation_embeds.pt"] if os.path.isfile(c)), None)
        if path is None:
            raise NotImplementedError("no cached validation embeddings (text encoding is outside the hot-path scope)")
        embeds = torch.load(path, map_location="cpu")
        gen = torch.Generator(device=self.accelerator.device).manual_seed(42)
        side = self.model.config.sample_size
        out = []
        for pe, pm, ne, nm in embeds:
            out.append(sample_latents_pixart(self.model, pe, pm, ne, nm, side, side, num_inference_steps=20, guidance_scale=5.0,
                                             generator=gen).cpu())
        os.makedirs(f"models/{self.global_step}", exist_ok=True)
        torch.save(out, f"models/{self.global_step}/validation_latents.pt")
        if self.logger is not None:
            for idx, lat in enumerate(out):
                x = lat[0, :3].float()
                x = (x - x.amin()) / (x.amax() - x.amin()).clamp_min(1e-6)
                self.logger.add_image(f"validation_latents/{idx}", x, self.global_step)
        return out

    def optimize(self, ratio, latents, embeddings, repa_tokens=None, generator: torch.Generator = None):
        """train_pixart_sigma.py:151-185 on the HIP path.  The reference draws noise and timesteps from the GLOBAL RNGs
        (:170,172) and ignores the trainer's per-step generator; so does this.  With gradients enabled (the training call,
        common/trainer.py:337) the step runs on the allocation-free device path (one packed H2D copy, launch plans:
        ``PixArtRecipe.optimize_device``) and the returned loss is marked so that ``accelerator.backward`` does not run a second
        backward; under ``no_grad`` (exploration trials) it is the plain forward + loss."""
        if torch.is_grad_enabled() and not latents.is_cuda and os.environ.get("YAT_TRAINER_FAST", "1") != "0":
            loss = self.recipe.optimize_device(latents, embeddings, None, gscale=1.0 / self.accelerator.gradient_accumulation_steps)
            loss.yat_backward_done = True
            return loss
        return self.recipe.optimize(latents, embeddings, None)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", required=True, type=str)
    parser.add_argument("--max-steps", type=int, default=None)
    args = parser.parse_args()
    params = TrainingParameters()
    params.read_yaml(args.config)
    if params.extract_features:
        raise SystemExit("extract_features (VAE/text-encoder feature extraction) is outside this build's scope")
    trainer = PixartSigmaTrainer(params)
    trainer.run(max_steps=args.max_steps)
