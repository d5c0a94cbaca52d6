"""End-to-end trainer loop on the GPU: cached-feature shards -> bucket sampler -> SanaModel.optimize (HIP recipe) ->
backward -> clip + AdamW (+EMA, warm-up) -> checkpoint in the diffusers layout -> reload.  This is the reference's
`Model.run` (common/trainer.py:298-403) driven through the same YAML keys, on a tiny SANA configuration."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _write_shards(tmp, cfg, n_shards=2, per=16):
    from yat_amd.common.shards import write_shard
    from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN
    g = torch.Generator().manual_seed(0)
    paths = []
    for s in range(n_shards):
        samples = []
        for i in range(per):
            r = ["1.0", "0.5", "2.0"][(i + s) % 3]
            H, W = ASPECT_RATIO_1024_BIN[r]
            L = int(torch.randint(3, 40, (1,), generator=g))
            samples.append(dict(__key__=f"{s:03d}{i:05d}", ratio=r,
                                latent=(torch.randn(cfg.in_channels, int(H) // 128, int(W) // 128, generator=g) * 0.5).to(BF),
                                emb=torch.randn(L, cfg.caption_channels, generator=g).to(BF)))
        p = str(tmp / f"shard-{s:06d}.tar")
        write_shard(p, samples)
        paths.append(p)
    return paths


def test_trainer_runs_saves_and_reloads(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from train_sana import SanaModel
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    cfg = SanaConfig(num_layers=2, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=32)
    paths = _write_shards(tmp_path, cfg)
    yaml_path = tmp_path / "config.yaml"
    yaml_path.write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], "num_shards: 2", "dataset_seed: 7",
        "batch_size: 4", "learning_rate: 1e-3", "steps: 6", "num_steps_per_validation: 3", "validation_prompts:", "  - x",
        "bfloat16: true", "gradient_accumulation_steps: 2", "warmup_steps: 2", "weight_decay: 0.01", "aspect_ratio: 1024",
        "use_ema: true", "train_unconditional_prob: 0.0", ""]))
    g = torch.Generator().manual_seed(1)
    pe = torch.randn(1, 12, cfg.caption_channels, generator=g).to(BF)
    torch.save([(pe, torch.ones(1, 12, dtype=torch.long), torch.zeros(1, 12, cfg.caption_channels, dtype=BF),
                 torch.cat([torch.ones(1, 1, dtype=torch.long), torch.zeros(1, 11, dtype=torch.long)], 1))],
               tmp_path / "validation_embeds.pt")
    monkeypatch.chdir(tmp_path)                       # the trainer writes models/<step>/ relative to the cwd
    params = TrainingParameters()
    params.read_yaml(str(yaml_path))
    trainer = SanaModel(params, config=cfg)
    before = trainer.model.flat_param.clone()
    trainer.run()
    torch.cuda.synchronize()
    losses = [float(l) for l in trainer.loss_history]
    assert len(losses) == 6 and all(torch.isfinite(torch.tensor(losses))), losses
    assert not torch.equal(before, trainer.model.flat_param)
    # validation cadence 3 -> checkpoints at steps 0 and 3 in the diffusers layout; they reload into the same module
    saved = sorted(os.listdir(tmp_path / "models"))
    assert saved, "no checkpoint written"
    ck = tmp_path / "models" / saved[-1]
    assert (ck / "config.json").exists() and (ck / "diffusion_pytorch_model.safetensors").exists()
    assert json.loads((ck / "config.json").read_text())["_class_name"] == "SanaTransformer2DModel"
    lat = torch.load(ck / "validation_latents.pt")    # 20-step CFG sampling from the cached validation embeddings
    assert len(lat) == 1 and lat[0].shape == (1, cfg.in_channels, cfg.sample_size, cfg.sample_size)
    assert torch.isfinite(lat[0].float()).all()
    re = SanaTransformer2DModelHIP.from_pretrained(str(ck), device="cuda")
    assert re.flat_param.shape == trainer.model.flat_param.shape and torch.isfinite(re.flat_param.float()).all()
    # the TensorBoard event file (runs/<date>_<host>/events.out.tfevents.*) holds every step's loss and the latent preview
    from yat_amd.common.tb_writer import read_events
    ev = read_events(trainer.logger.path)
    logged = [(e["step"], e["value"]) for e in ev if e.get("tag") == "train/loss"]
    assert [s for s, _ in logged] == list(range(6))
    assert all(abs(v - l) <= 1e-6 * max(1.0, abs(l)) for (_, v), l in zip(logged, losses))
    assert any(e.get("tag") == "validation_latents/0" and "image" in e for e in ev)


def test_overfits_a_fixed_batch():
    """Learning sanity at real width (D=2240, 2 blocks, B=4, 16x16 latents): 40 optimizer steps on ONE fixed batch with a
    fixed (noise, timestep) draw must drive the flow-matching loss down by a large factor -- forward, backward, clip and
    AdamW all have to be right for that, on the four-stream schedule, with the overlapped per-bucket update."""
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.optim import FlatAdamW
    cfg = SanaConfig(num_layers=2)
    model = SanaTransformer2DModelHIP(cfg, device="cuda").init_synthetic(seed=0)
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.0, max_grad_norm=1.0, overlap_update=True)
    recipe = SanaRecipe(model, pad_to=512, device="cuda")
    g = torch.Generator().manual_seed(3)
    latents = (torch.randn(4, cfg.in_channels, 16, 16, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (30, 120, 64, 200)]
    losses = []
    for _ in range(40):
        loss = recipe.optimize(latents, embs, torch.Generator().manual_seed(9))      # same draw every step
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    model.join_pending_update()
    torch.cuda.synchronize()
    print("[train] overfit losses:", [round(x, 4) for x in losses[::5]], round(losses[-1], 4))
    assert all(l == l and l < 1e4 for l in losses)
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("algo", ["lokr", "lora", "loha", "dora"])
def test_trainer_lokr_config(tmp_path, monkeypatch, algo):
    """BASELINE config 5 plumbing end to end on a tiny model: lora_* YAML keys -> LoKr (or plain LoRA) adapters on the README
    target modules, frozen base, AdamW over the adapter set only, peft-layout adapter checkpoint."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from safetensors.torch import load_file
    from train_sana import SanaModel
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sana import SanaConfig
    cfg = SanaConfig(num_layers=2, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=32)
    paths = _write_shards(tmp_path, cfg)
    yaml_path = tmp_path / "config.yaml"
    targets = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]
    yaml_path.write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], "num_shards: 2", "dataset_seed: 7",
        "batch_size: 4", "learning_rate: 1e-3", "steps: 4", "num_steps_per_validation: 2", "validation_prompts:", "  - x",
        "bfloat16: true", "lora_rank: 2", "lora_alpha: 2", f"lora_algo: {'lora' if algo == 'dora' else algo}", *(["lora_use_dora: true"] if algo == "dora" else []),
        *(["lora_dropout: 0.05"] if algo in ("lokr", "loha") else []),
        "lora_target_modules:", *[f"  - {t}" for t in targets], "aspect_ratio: 1024", ""]))
    monkeypatch.chdir(tmp_path)
    params = TrainingParameters()
    params.read_yaml(str(yaml_path))
    trainer = SanaModel(params, config=cfg)
    base = trainer.model.flat_param.clone()
    trainer.run()
    torch.cuda.synchronize()
    assert trainer.adapters is not None and len(trainer.loss_history) == 4
    assert all(torch.isfinite(torch.tensor([float(l) for l in trainer.loss_history])))
    assert torch.equal(base, trainer.model.flat_param), "the frozen base moved"
    saved = sorted(os.listdir(tmp_path / "models"))
    ck = tmp_path / "models" / saved[-1]
    sd = load_file(str(ck / "adapter_model.safetensors"))
    conf = json.loads((ck / "adapter_config.json").read_text())
    assert conf["r"] == 2 and conf["target_modules"] == targets
    # lora_pretrained: a second trainer resumes from the saved adapter (PeftModel.from_pretrained, common/trainer.py:236)
    yaml_path.write_text(yaml_path.read_text() + f"lora_pretrained: {ck}\n")
    params2 = TrainingParameters()
    params2.read_yaml(str(yaml_path))
    resumed = SanaModel(params2, config=cfg)
    resumed.initialize()
    assert type(resumed.adapters) is type(trainer.adapters)
    for k, v in resumed.adapters.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    if algo == "dora":         # lora_use_dora is true when the key is present (common/training_parameters_reader.py:142,192)
        from yat_amd.dora import DoRAAdapters
        assert isinstance(trainer.adapters, DoRAAdapters) and conf["peft_type"] == "LORA" and conf["use_dora"] is True
        assert len(sd) == 3 * len(trainer.adapters.entries)
        assert sd["base_model.model.transformer_blocks.1.attn2.to_out.0.lora_magnitude_vector.weight"].shape == (128,)
        assert any(v.abs().max() > 0 for k, v in sd.items() if k.endswith("lora_B.weight")), "lora_B never left its zero init"
        return
    if algo == "lora":
        assert conf["peft_type"] == "LORA" and len(sd) == 2 * len(trainer.adapters.entries)
        assert sd["base_model.model.transformer_blocks.1.attn2.to_out.0.lora_B.weight"].shape == (128, 2)
        assert any(v.abs().max() > 0 for k, v in sd.items() if k.endswith("lora_B.weight")), "lora_B never left its zero init"
        return
    if algo == "loha":
        assert conf["peft_type"] == "LOHA" and len(sd) == 4 * len(trainer.adapters.entries)
        assert sd["base_model.model.transformer_blocks.1.attn2.to_out.0.hada_w1_a"].shape == (128, 2)
        assert any(v.abs().max() > 0 for k, v in sd.items() if k.endswith("hada_w2_a")), "hada_w2_a never left its zero init"
        return
    assert conf["peft_type"] == "LOKR"
    assert "base_model.model.transformer_blocks.1.attn2.to_out.0.lokr_w1" in sd
    assert "base_model.model.patch_embed.proj.lokr_w2_a" in sd and len(sd) == 3 * len(trainer.adapters.entries)
    assert any(v.abs().max() > 0 for k, v in sd.items() if k.endswith("lokr_w1")), "w1 never left its zero init"


def test_pixart_trainer_runs_and_saves(tmp_path, monkeypatch):
    """BASELINE config 3 through the same trainer loop: `train_pixart_sigma.py` entry class on a tiny PixArt-Sigma
    configuration (cached latents at VAE compression 8, DDPM epsilon-prediction recipe, bf16 loss), checkpoint in the
    diffusers layout, reload."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from train_pixart_sigma import PixartSigmaTrainer
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.common.shards import write_shard
    from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN
    from yat_amd.pixart import PixArtConfig, PixArtTransformer2DModelHIP
    cfg = PixArtConfig(num_layers=2, num_attention_heads=2, attention_head_dim=24, cross_attention_dim=48,
                       caption_channels=64, sample_size=128)
    g = torch.Generator().manual_seed(0)
    paths = []
    for s in range(2):
        samples = []
        for i in range(16):
            r = ["1.0", "0.5", "2.0"][(i + s) % 3]
            H, W = ASPECT_RATIO_1024_BIN[r]
            L = int(torch.randint(3, 40, (1,), generator=g))
            samples.append(dict(__key__=f"{s:03d}{i:05d}", ratio=r,          # 1024 px bucket / 32: small latents, same ratios
                                latent=(torch.randn(4, int(H) // 32, int(W) // 32, generator=g) * 0.5).to(BF),
                                emb=torch.randn(L, cfg.caption_channels, generator=g).to(BF)))
        p = str(tmp_path / f"shard-{s:06d}.tar")
        write_shard(p, samples)
        paths.append(p)
    yaml_path = tmp_path / "config.yaml"
    yaml_path.write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], "num_shards: 2", "dataset_seed: 7",
        "batch_size: 4", "learning_rate: 1e-3", "steps: 5", "num_steps_per_validation: 4", "validation_prompts:", "  - x",
        "bfloat16: true", "gradient_accumulation_steps: 1", "warmup_steps: 2", "weight_decay: 0.0", "aspect_ratio: 1024",
        "train_unconditional_prob: 0.0", ""]))
    monkeypatch.chdir(tmp_path)
    params = TrainingParameters()
    params.read_yaml(str(yaml_path))
    trainer = PixartSigmaTrainer(params, config=cfg)
    before = trainer.model.flat_param.clone()
    trainer.run()
    torch.cuda.synchronize()
    losses = [float(l) for l in trainer.loss_history]
    assert len(losses) == 5 and all(l == l and l < 1e3 for l in losses), losses
    assert not torch.equal(before, trainer.model.flat_param)
    saved = sorted(os.listdir(tmp_path / "models"), key=int)
    ck = tmp_path / "models" / saved[-1]
    assert json.loads((ck / "config.json").read_text())["_class_name"] == "PixArtTransformer2DModel"
    re = PixArtTransformer2DModelHIP.from_pretrained(str(ck), device="cuda")
    assert re.flat_param.shape == trainer.model.flat_param.shape and torch.isfinite(re.flat_param.float()).all()


def _tiny_trainer(tmp_path, monkeypatch, extra_yaml=(), steps=3):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from train_sana import SanaModel
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sana import SanaConfig
    cfg = SanaConfig(num_layers=1, num_attention_heads=2, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=32, cross_attention_dim=64, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=32)
    paths = _write_shards(tmp_path, cfg)
    yaml_path = tmp_path / "config.yaml"
    yaml_path.write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], "num_shards: 2", "dataset_seed: 7",
        "batch_size: 4", "learning_rate: 1e-3", f"steps: {steps}", "num_steps_per_validation: 100", "validation_prompts:",
        "  - x", "bfloat16: true", "aspect_ratio: 1024", *extra_yaml, ""]))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("YAT_TENSORBOARD", "0")
    params = TrainingParameters()
    params.read_yaml(str(yaml_path))
    return SanaModel(params, config=cfg), cfg


def test_cfg_dropout_uses_cached_empty_embedding(tmp_path, monkeypatch):
    """Whole-batch CFG dropout (common/trainer.py:306-308,319-323) with ``train_unconditional_prob: 1.0``: every step must
    train on the empty-prompt embedding, read from ``empty_embeds.pt`` beside the shards (the text encoder that produces it
    in the reference is outside the hot path).  Without the file the trainer names it in the error."""
    trainer, cfg = _tiny_trainer(tmp_path, monkeypatch, ["train_unconditional_prob: 1.0"])
    with pytest.raises(FileNotFoundError, match="empty_embeds.pt"):
        trainer.run()
    trainer, cfg = _tiny_trainer(tmp_path, monkeypatch, ["train_unconditional_prob: 1.0"])
    empty = torch.randn(2, cfg.caption_channels, generator=torch.Generator().manual_seed(5)).to(BF)
    torch.save([empty], tmp_path / "empty_embeds.pt")
    seen, inner = [], trainer.optimize

    def spy(ratio, latents, embeddings, repa, generator):
        seen.append([e.clone() for e in embeddings])
        return inner(ratio, latents, embeddings, repa, generator)
    trainer.optimize = spy
    trainer.run()
    torch.cuda.synchronize()
    assert len(seen) == 3
    for embs in seen:
        assert len(embs) == 4 and all(torch.equal(e.cpu(), empty) for e in embs)
    # the conditioning really reached the kernels: same batch, same draw, real captions -> a different loss
    losses = [float(l) for l in trainer.loss_history]
    assert all(l == l for l in losses)
    trainer2, _ = _tiny_trainer(tmp_path, monkeypatch, ["train_unconditional_prob: 0.0"])
    trainer2.run()
    torch.cuda.synchronize()
    assert [float(l) for l in trainer2.loss_history] != losses


def test_exploration_steps_keep_the_min_loss_draw(tmp_path, monkeypatch):
    """:326-336 -- k no-grad trial draws from the step's generator, then the training draw restarts from the RNG state of the
    trial with the smallest loss: the loss the step logs must equal the smallest trial loss (same state -> same noise and
    timesteps -> same arithmetic, bit for bit)."""
    trainer, cfg = _tiny_trainer(tmp_path, monkeypatch, ["exploration_steps: 3"], steps=2)
    calls, inner = [], trainer.optimize

    def spy(ratio, latents, embeddings, repa, generator):
        loss = inner(ratio, latents, embeddings, repa, generator)
        calls.append((torch.is_grad_enabled(), float(loss.detach())))
        return loss
    trainer.optimize = spy
    trainer.run()
    torch.cuda.synchronize()
    assert len(calls) == 2 * 4
    for s in range(2):
        trial = calls[4 * s: 4 * s + 3]
        train = calls[4 * s + 3]
        assert all(not g for g, _ in trial) and train[0]
        assert train[1] == min(l for _, l in trial), (trial, train)


def _trainer_world2_worker(rank, world, port, tmp, shard):
    """One rank of a two-rank trainer job on this one GPU (gloo: RCCL refuses two ranks on one device)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      YAT_DIST_BACKEND="gloo", YAT_SHARD_OPTIMIZER="1" if shard else "0", YAT_TENSORBOARD="0")
    os.chdir(tmp)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.cuda.set_device(0)
    from train_sana import SanaModel
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sana import SanaConfig
    cfg = SanaConfig(num_layers=2, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=32)
    params = TrainingParameters()
    params.read_yaml(os.path.join(tmp, "config.yaml"))
    torch.manual_seed(5)
    import random
    random.seed(5)
    trainer = SanaModel(params, config=cfg)
    trainer.run()
    torch.cuda.synchronize()
    ddp = trainer.accelerator.ddp
    assert ddp is not None and (ddp.shard is not None) == shard and ddp.world == 2
    opt = trainer.optimizer
    opt.gather_ema()
    torch.save(dict(param=trainer.model.flat_param.cpu(), ema=opt.ema_shadow.cpu(), losses=[float(x) for x in trainer.loss_history],
                    norm=opt.grad_norm.cpu()), os.path.join(tmp, f"trainer_rank{rank}_{'shard' if shard else 'rep'}.pt"))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_trainer_world2_sharded_optimizer_equals_replicated(tmp_path):
    """The whole trainer loop (common/trainer.py:298-403) as a two-rank job on one GPU, once with the replicated optimizer step
    and once with YAT_SHARD_OPTIMIZER=1: gradient accumulation 2, warm-up, EMA, validation + checkpoint (the EMA shadow is
    all-gathered instead of averaged), the logged loss through the reference's own gather.  Same shards, same draws -> the same
    parameters and EMA shadow on both ranks, bit for bit, and the same logged losses (to the carried loss's own 2e-3)."""
    import socket
    import torch.multiprocessing as mp
    from yat_amd.sana import SanaConfig
    cfg = SanaConfig(num_layers=2, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=32)
    paths = _write_shards(tmp_path, cfg)
    (tmp_path / "config.yaml").write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], "num_shards: 2", "dataset_seed: 7",
        "batch_size: 4", "learning_rate: 1e-3", "steps: 4", "num_steps_per_validation: 2", "validation_prompts:", "  - x",
        "bfloat16: true", "gradient_accumulation_steps: 2", "warmup_steps: 2", "weight_decay: 0.01", "aspect_ratio: 1024",
        "use_ema: true", "train_unconditional_prob: 0.0", ""]))
    g = torch.Generator().manual_seed(1)
    pe = torch.randn(1, 12, cfg.caption_channels, generator=g).to(BF)
    torch.save([(pe, torch.ones(1, 12, dtype=torch.long), torch.zeros(1, 12, cfg.caption_channels, dtype=BF),
                 torch.cat([torch.ones(1, 1, dtype=torch.long), torch.zeros(1, 11, dtype=torch.long)], 1))],
               tmp_path / "validation_embeds.pt")
    out = {}
    for shard in (False, True):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_trainer_world2_worker, args=(2, port, str(tmp_path), shard), nprocs=2, join=True)
        out[shard] = [torch.load(tmp_path / f"trainer_rank{r}_{'shard' if shard else 'rep'}.pt") for r in (0, 1)]
    rep, sh = out[False], out[True]
    assert torch.equal(rep[0]["param"], rep[1]["param"]) and torch.isfinite(rep[0]["param"].float()).all()
    for r in (0, 1):
        assert torch.equal(sh[r]["param"], rep[0]["param"]), f"rank {r}: the sharded trainer's parameters differ"
        assert torch.equal(sh[r]["ema"], rep[0]["ema"]) and torch.equal(sh[r]["norm"], rep[0]["norm"])
        assert len(sh[r]["losses"]) == 4
        for a, b in zip(sh[r]["losses"], rep[0]["losses"]):
            assert abs(a - b) <= 2e-3 * abs(b), (sh[r]["losses"], rep[0]["losses"])
    assert sorted(os.listdir(tmp_path / "models")), "no checkpoint written by the two-rank job"
