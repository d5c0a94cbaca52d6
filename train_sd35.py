#!/usr/bin/env python3
"""SD3.5 trainer entry point -- same CLI as the reference (`train_sd35.py --config config.yaml`, train_sd35.py:196-204),
driving the MI355X-native MMDiT path (BASELINE config 4).

    python train_sd35.py --config config.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_sd35.py --config config.yaml

``pretrained_model_path`` (or ``pretrained_pipe_path``/transformer) must be a LOCAL diffusers directory; with neither the
SD3.5-Medium architecture is random-initialised (no network here).  The three text encoders, the VAE and the validation
pipeline (:94-163) are outside the hot-path scope: training consumes cached-feature shards whose samples carry the prompt
embeddings (``emb.pt`` [333, 4096]) and the pooled projection (``pooled.pt`` [2048]).

Reference quirks: ``SD35Trainer.optimize(self, model, batch)`` (:165) has a pre-refactor signature with a
``(latents, embeddings, pooled_projections)`` batch no sampler produces any more, while ``Model.run`` calls
``optimize(ratio, latents, embeddings, repa_features, generator)`` (common/trainer.py:337) -- at HEAD the reference's SD3.5
entry raises TypeError on the first step; ``initialize`` (:160-163) refers to undefined names.  Here the recipe body is the
one written at :165-194 under the trainer's signature, with ``embeddings`` = the sampler's list of (prompt_embeds, pooled)
pairs.  Like the reference, noise and timestep draws come from the GLOBAL RNGs (:180,182), not the trainer's generator.
"""
import argparse
import json
import os

import torch

from yat_amd.common.training_parameters_reader import TrainingParameters
from yat_amd.common.trainer import Model
from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN
from yat_amd.recipe import SD3Recipe
from yat_amd.scheduler import FlowMatchSchedule
from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP


class SD35Trainer(Model):
    def __init__(self, params: TrainingParameters, accelerator=None, config: SD3Config | None = None):
        super().__init__(params, accelerator)
        dev = self.accelerator.device
        path = params.pretrained_model_path
        if path is None and params.pretrained_pipe_path and os.path.isdir(os.path.join(params.pretrained_pipe_path, "transformer")):
            path = os.path.join(params.pretrained_pipe_path, "transformer")
        if path is not None and os.path.isdir(path):
            self.model = SD3Transformer2DModelHIP.from_pretrained(path, device=dev)            # :28-43
        else:
            self.model = SD3Transformer2DModelHIP(config or SD3Config(), device=dev).init_synthetic(0)
        self.model.enable_gradient_checkpointing()                                              # :47 (no-op here)
        shift = 3.0
        sched_cfg = os.path.join(params.pretrained_pipe_path or "", "scheduler", "scheduler_config.json")
        if os.path.isfile(sched_cfg):                                                           # :49
            with open(sched_cfg) as f:
                shift = float(json.load(f).get("shift", shift))
        self.scheduler = FlowMatchSchedule(shift=shift)
        self.aspect_ratios = ASPECT_RATIO_1024_BIN                                              # :55
        self.recipe = SD3Recipe(self.model, self.scheduler, device=dev)
        self.pipe = None

    def extract_latents(self, images):
        raise NotImplementedError("VAE encoding is outside the hot-path scope; train from cached-feature shards")

    def extract_embeddings(self, captions):
        raise NotImplementedError("text encoding is outside the hot-path scope; train from cached-feature shards")

    def check_empty_embeddings(self, emb, path):
        """SD3.5's per-sample embedding is a ``(prompt_embeds [T, C], pooled [P])`` pair (``SD3Recipe.stack_embeddings``), so
        ``empty_embeds.pt`` holds that pair for the empty prompt -- as the pair itself or as a list with the pair (what
        ``extract_embeddings([''])`` returns, :76-92).  A bare tensor (the SANA / PixArt layout) is an error here: CFG
        dropout would slice it by rows."""
        if isinstance(emb, (list, tuple)) and len(emb) == 1 and isinstance(emb[0], (list, tuple)):
            emb = emb[0]
        ok = (isinstance(emb, (list, tuple)) and len(emb) == 2 and all(torch.is_tensor(t) for t in emb)
              and emb[0].ndim in (2, 3) and emb[1].numel() == emb[1].shape[-1])
        if not ok:
            raise ValueError(f"{path}: SD3.5 needs the empty prompt's (prompt_embeds [T, C], pooled [P]) pair, got "
                             f"{type(emb).__name__}" + (f" of shape {tuple(emb.shape)}" if torch.is_tensor(emb) else ""))
        prompt = emb[0] if emb[0].ndim == 2 else emb[0][0]
        return [(prompt, emb[1].reshape(-1))]

    def validate(self):
        """Middle third of train_sd35.py:94-162: 20-step flow-match Euler sampling with CFG 5.0 over the HIP MMDiT (:129-142),
        generator seeded 42 (:110).  The three text encoders and the VAE are outside this build's scope, so the prompt
        embeddings come from a cached file (``validation_embeds.pt`` next to the shards or in the cwd: a list of
        (prompt_embeds [1,T,C], negative_prompt_embeds, pooled_prompt_embeds [1,P], negative_pooled_prompt_embeds) tuples as
        ``pipe.encode_prompt`` returns them, :116-118) and the result is the latents (``output_type='latent'``), stored
        under models/<step>/ with a three-channel preview for the logger in place of the decoded image (:154)."""
        from yat_amd.sampler import sample_latents_sd3
        cands = [os.path.join(os.path.dirname(p), "validation_embeds.pt") for p in (self.params.local_shard_paths or [])]
        path = next((c for c in cands + ["validation_embeds.pt"] if os.path.isfile(c)), None)
        if path is None:
            raise NotImplementedError("no cached validation embeddings (text encoding is outside the hot-path scope)")
        embeds = torch.load(path, map_location="cpu")
        gen = torch.Generator().manual_seed(42)                                              # a CPU generator (:110)
        side = self.model.cfg.sample_size
        out = []
        for pe, ne, pp, npp in embeds:
            out.append(sample_latents_sd3(self.model, pe, pp, ne, npp, side, side, num_inference_steps=20, guidance_scale=5.0,
                                          generator=gen, schedule=self.scheduler).cpu())
        os.makedirs(f"models/{self.global_step}", exist_ok=True)
        torch.save(out, f"models/{self.global_step}/validation_latents.pt")
        if self.logger is not None:
            for idx, lat in enumerate(out):
                x = lat[0, :3].float()
                x = (x - x.amin()) / (x.amax() - x.amin()).clamp_min(1e-6)
                self.logger.add_image(f"validation_latents/{idx}", x, self.global_step)
        return out

    def optimize(self, ratio, latents, embeddings, repa_tokens=None, generator: torch.Generator = None):
        """train_sd35.py:165-194 on the HIP path (yat_amd.recipe.SD3Recipe); global RNG streams as there.  With gradients
        enabled (the training call, common/trainer.py:337) the step runs on the allocation-free device path (one packed H2D copy,
        launch plans: ``SD3Recipe.optimize_device``) and the loss is marked "backward done"."""
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
    trainer = SD35Trainer(params)
    trainer.run(max_steps=args.max_steps)
