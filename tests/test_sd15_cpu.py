"""BASELINE config 1 -- the reference's CPU plumbing run of train_sd15.py: a 16-sample local shard -> bucket sampler -> SD15
``optimize`` (DDPM epsilon-prediction, train_sd15.py:140-165) -> backward -> clip(1.0) -> AdamW, batch 1, 10 steps, all on the
host, through the same YAML reader / trainer loop / shard format as the HIP recipes.  The UNet is the oracle's restatement of
the SD1.5 layout at a tiny width (diffusers is absent offline)."""
import os
import sys

import torch

BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sd15_cpu_plumbing_10_steps(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    from oracle.sd15_ref import UNet2DConditionRef, sd15_optimize_ref
    from train_sd15 import SD15Model
    from yat_amd.common.shards import write_shard
    from yat_amd.common.trainer import HipAccelerator
    from yat_amd.common.training_parameters_reader import TrainingParameters
    g = torch.Generator().manual_seed(0)
    # 16 "256 px" samples: 4 x 32 x 32 latents (VAE /8), CLIP-shaped embeddings [1, 77, C] as extract_embeddings returns them
    samples = [dict(__key__=f"{i:07d}", ratio="1.0", latent=(torch.randn(4, 32, 32, generator=g) * 0.5).to(BF),
                    emb=torch.randn(1, 77, 32, generator=g).to(BF)) for i in range(16)]
    shard = str(tmp_path / "shard-000000.tar")
    write_shard(shard, samples)
    (tmp_path / "config.yaml").write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", f"  - {shard}", "num_shards: 1", "dataset_seed: 1", "batch_size: 1",
        "learning_rate: 1e-3", "steps: 10", "num_steps_per_validation: 5", "validation_prompts:", "  - x", "bfloat16: true",
        "aspect_ratio: 512", "warmup_steps: 3", ""]))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("YAT_TENSORBOARD", "0")
    params = TrainingParameters()
    params.read_yaml(str(tmp_path / "config.yaml"))
    torch.manual_seed(11)
    unet = UNet2DConditionRef()
    trainer = SD15Model(params, accelerator=HipAccelerator(params.gradient_accumulation_steps, device="cpu"), unet=unet)
    before = {k: v.clone() for k, v in trainer.model.state_dict().items()}
    # first-step loss == the oracle's recipe restatement on the same model, batch and RNG state
    trainer.initialize()
    batch = next(iter(trainer.sampler))
    state = torch.get_rng_state()
    l_ref = sd15_optimize_ref(trainer.model, trainer.scheduler, batch.vae_features, batch.embeddings)
    torch.set_rng_state(state)
    l_mine = trainer.optimize(batch.ratio, batch.vae_features, batch.embeddings, None, torch.Generator())
    assert torch.equal(l_ref, l_mine)
    trainer.sampler = None                       # a fresh sampler for the run (initialize() builds it)
    trainer.run()
    losses = [float(l) for l in trainer.loss_history]
    assert len(losses) == 10 and all(l == l and l < 1e3 for l in losses), losses
    assert any(not torch.equal(before[k], v) for k, v in trainer.model.state_dict().items())
    assert trainer.lr_scheduler.get_last_lr()[0] == 1e-3                     # 3 warm-up steps done
    assert sorted(os.listdir(tmp_path / "models")) == ["0", "5"]             # save cadence :371,398 (step 0 included)
    # on a GPU device the entry point refuses: no CPU arithmetic path beside the HIP kernels
    class _Acc(HipAccelerator):
        def __init__(self):
            super().__init__(1, device="cpu")
            self.device = torch.device("cuda", 0)
    import pytest
    with pytest.raises(NotImplementedError, match="CPU plumbing"):
        SD15Model(params, accelerator=_Acc(), unet=unet)
