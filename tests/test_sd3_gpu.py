"""SD3.5 MMDiT (BASELINE config 4) on the GPU: kernels of csrc/mmdit_ops.hip against torch restatements, then the HIP model +
recipe (train_sd35.py:165-194) against the CPU oracle (oracle/sd3_ref.py) in bf16 and fp32 on fixed seeds.  Criteria are the
ones of tests/test_sana_gpu.py: as close to the fp32 truth as the oracle's own bf16 evaluation."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


# ------------------------------------------------------------------------------------------------ kernels
def _rms_ref(x, w, eps, dt):
    """diffusers RMSNorm [RECALL]: fp32 statistics and normalisation, cast to the weight dtype, then * weight."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    y = x.float() * torch.rsqrt(v + eps)
    if dt == BF:
        y = y.to(BF)
    return (y * w.to(dt)).to(dt)


@pytest.mark.parametrize("B,N,T,H,dh", [(2, 24, 10, 2, 64), (3, 40, 0, 3, 64), (2, 130, 33, 4, 32), (1, 17, 5, 2, 128)])
def test_qknorm_concat_fwd_bwd(B, N, T, H, dh):
    from yat_amd import ops
    D, L, eps = H * dh, N + T, 1e-6
    qkv_i, qkv_t = rnd(B * N, 3 * D, seed=1), (rnd(B * T, 3 * D, seed=2) if T else None)
    ws = [(1.0 + 0.2 * torch.randn(dh, generator=torch.Generator().manual_seed(10 + k))).to(BF).to(DEV) for k in range(4)]
    joint = torch.empty(B * L, 3 * D, dtype=BF, device=DEV)
    rstd = torch.empty(B * L, 2 * H, dtype=torch.float32, device=DEV)
    ops.qknorm_concat_fwd(qkv_i, qkv_t, B, N, T, H, dh, eps, ws[0], ws[1], ws[2] if T else None, ws[3] if T else None, joint, rstd)

    def ref(dt):
        xi = qkv_i.to(dt).cpu().view(B, N, 3, H, dh).requires_grad_(True)
        parts = [xi]
        w = [t.cpu().to(dt).requires_grad_(True) for t in ws]
        q = [_rms_ref(xi[:, :, 0], w[0], eps, dt)]
        k = [_rms_ref(xi[:, :, 1], w[1], eps, dt)]
        v = [xi[:, :, 2]]
        if T:
            xt = qkv_t.to(dt).cpu().view(B, T, 3, H, dh).requires_grad_(True)
            parts.append(xt)
            q.append(_rms_ref(xt[:, :, 0], w[2], eps, dt)); k.append(_rms_ref(xt[:, :, 1], w[3], eps, dt)); v.append(xt[:, :, 2])
        out = torch.stack([torch.cat(q, 1), torch.cat(k, 1), torch.cat(v, 1)], dim=2)      # [B, L, 3, H, dh]
        return out.reshape(B * L, 3 * D), parts, w
    o_bf, _, _ = ref(BF)
    assert torch.equal(joint.cpu(), o_bf.detach()), "forward is not bit-identical to the bf16 restatement"
    # backward against fp32 autograd of the same function
    dj = rnd(B * L, 3 * D, seed=5)
    o32, parts, w32 = ref(torch.float32)
    o32.backward(dj.float().cpu())
    dqi = torch.empty_like(qkv_i)
    dqt = torch.empty_like(qkv_t) if T else None
    dws = [torch.zeros(dh, dtype=BF, device=DEV) for _ in range(4)]
    wsb = torch.empty(ops.qknorm_concat_bwd_workspace_bytes(B, N, T, dh), dtype=torch.uint8, device=DEV)
    ops.qknorm_concat_bwd(qkv_i, qkv_t, B, N, T, H, dh, ws[0], ws[1], ws[2] if T else None, ws[3] if T else None, rstd, dj,
                          dqi, dqt, dws[0], dws[1], dws[2] if T else None, dws[3] if T else None, wsb)
    e = rel(dqi, parts[0].grad.reshape(B * N, 3 * D))
    print(f"[parity] qknorm_concat B={B} N={N} T={T} H={H} dh={dh}: fwd bit-exact; d_img rel={e:.3e}")
    assert e <= 4e-3
    if T:
        assert rel(dqt, parts[1].grad.reshape(B * T, 3 * D)) <= 4e-3
    for k in range(4 if T else 2):
        ek = rel(dws[k], w32[k].grad)
        assert ek <= 6e-3, (k, ek)
    # accumulate_dw adds to what is there
    before = [t.clone() for t in dws]
    ops.qknorm_concat_bwd(qkv_i, qkv_t, B, N, T, H, dh, ws[0], ws[1], ws[2] if T else None, ws[3] if T else None, rstd, dj,
                          dqi, dqt, dws[0], dws[1], dws[2] if T else None, dws[3] if T else None, wsb, accumulate_dw=True)
    assert rel(dws[0], 2 * before[0].float()) <= 8e-3


def test_joint_rows_roundtrip():
    from yat_amd import ops
    B, N, T, C = 3, 20, 7, 48
    img, txt = rnd(B * N, C, seed=1), rnd(B * T, C, seed=2)
    joint = torch.empty(B * (N + T), C, dtype=BF, device=DEV)
    ops.joint_rows(joint, img, txt, B, N, T, to_joint=True)
    expect = torch.cat([img.view(B, N, C), txt.view(B, T, C)], dim=1).reshape(-1, C)
    assert torch.equal(joint, expect)
    i2, t2 = torch.empty_like(img), torch.empty_like(txt)
    ops.joint_rows(joint, i2, t2, B, N, T, to_joint=False)
    assert torch.equal(i2, img) and torch.equal(t2, txt)
    ops.joint_rows(joint, img, None, B, N, T, to_joint=True)                 # no text side: zeros
    assert joint.view(B, N + T, C)[:, N:].abs().max().item() == 0 and torch.equal(joint.view(B, N + T, C)[:, :N], img.view(B, N, C))


# ------------------------------------------------------------------------------------------------ model + recipe
def _setup(cfg_kw, seed=0):
    from oracle.sd3_ref import SD3Config as RefCfg, SD3TransformerRef, init_like_pretrained
    from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP
    rcfg = RefCfg.tiny(**cfg_kw) if not cfg_kw.pop("_full", False) else RefCfg(**cfg_kw)
    ref = SD3TransformerRef(rcfg)
    init_like_pretrained(ref, seed)
    ref_bf = copy.deepcopy(ref).to(BF)
    ref_32 = copy.deepcopy(ref_bf).float()
    hip = SD3Transformer2DModelHIP(SD3Config(**{k: getattr(rcfg, k) for k in SD3Config.__dataclass_fields__}), device=DEV)
    hip.load_state_dict(ref_bf.state_dict())
    return rcfg, ref_bf, ref_32, hip


def _compare_step(rcfg, ref_bf, ref_32, hip, B, Hl, Wl, T, tag, seed=0, check_adamw=True):
    from oracle.sd3_ref import optimize_ref
    from oracle.recipe_ref import FlowMatchSchedule as RefSched
    from yat_amd.recipe import SD3Recipe
    from yat_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(100 + seed)
    latents = (torch.randn(B, rcfg.in_channels, Hl, Wl, generator=g) * 0.5).to(BF)
    prompt = torch.randn(B, T, rcfg.joint_attention_dim, generator=g).to(BF)
    pooled = torch.randn(B, rcfg.pooled_projection_dim, generator=g).to(BF)
    outs = {}
    for name, model, dt in (("bf16", ref_bf, BF), ("fp32", ref_32, torch.float32)):
        loss, pred, _ = optimize_ref(model, RefSched(), latents, prompt, pooled, torch.Generator().manual_seed(7), dt)
        loss.backward()
        outs[name] = (loss.item(), pred.detach())
    recipe = SD3Recipe(hip, device=DEV)
    loss, pred, _ = recipe.optimize(latents, (prompt, pooled), torch.Generator().manual_seed(7), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    l_h, l_b, l_t = loss.item(), outs["bf16"][0], outs["fp32"][0]
    print(f"[parity] sd3 {tag}: loss hip={l_h:.6f} oracle_bf16={l_b:.6f} oracle_fp32={l_t:.6f}")
    # the loss itself is a bf16 number in this recipe (MSELoss evaluated in bf16): one bf16 ulp of slack
    assert abs(l_h - l_t) <= 1.15 * abs(l_b - l_t) + 2 ** -7 * abs(l_t)
    e_h, e_b = rel(pred, outs["fp32"][1]), rel(outs["bf16"][1], outs["fp32"][1])
    print(f"[parity] sd3 {tag}: pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_b:.3e} "
          f"hip_vs_oracle_bf16={rel(pred, outs['bf16'][1]):.3e}")
    assert e_h <= 1.15 * e_b + 1e-3
    p32 = dict(ref_32.named_parameters())
    worst, num_h, num_b, den = [], 0.0, 0.0, 0.0
    for name, pb in ref_bf.named_parameters():
        gh, gb, gt = hip.G[name].float().cpu(), pb.grad.float(), p32[name].grad.float()
        assert torch.isfinite(gh).all(), name
        num_h += (gh - gt).pow(2).sum().item(); num_b += (gb - gt).pow(2).sum().item(); den += gt.pow(2).sum().item()
        worst.append((rel(gh, gt), rel(gb, gt), name))
    tot_h, tot_b = (num_h / den) ** 0.5, (num_b / den) ** 0.5
    print(f"[parity] sd3 {tag}: grads (all params) hip_vs_fp32={tot_h:.3e} oracle_bf16_vs_fp32={tot_b:.3e}")
    for eh, eb, name in sorted(worst, reverse=True)[:6]:
        print(f"[parity]   {name}: hip={eh:.3e} oracle_bf16={eb:.3e}")
    assert tot_h <= 1.15 * tot_b + 1e-3
    for eh, eb, name in worst:
        assert eh <= 2.0 * eb + 2e-2, (name, eh, eb)
    if check_adamw:
        opt_ref = torch.optim.AdamW(ref_bf.parameters(), lr=1e-3, weight_decay=0.01)
        torch.nn.utils.clip_grad_norm_(ref_bf.parameters(), max_norm=1.0)
        opt_ref.step()
        FlatAdamW(hip, lr=1e-3, weight_decay=0.01).step()
        torch.cuda.synchronize()
        n_bad = n_all = 0
        for name, pb in ref_bf.named_parameters():
            a, b = hip.P[name].float().cpu(), pb.data.float()
            n_bad += ((a - b).abs() > 2.0 ** -7 * b.abs().clamp_min(1e-30)).sum().item()
            n_all += b.numel()
        print(f"[parity] sd3 {tag}: AdamW {n_bad}/{n_all} parameters differ by more than 1 bf16 ulp from torch CPU")
        assert n_bad <= 0.01 * n_all


@pytest.mark.parametrize("B,Hl,Wl,T,kw", [
    (2, 12, 8, 10, {}),                                                  # 3 blocks: two dual, last context_pre_only
    (3, 8, 20, 33, dict(num_layers=4, dual_attention_layers=(1,))),      # non-square bucket, a plain block first
    (1, 16, 16, 7, dict(num_layers=1, dual_attention_layers=())),        # a single (context_pre_only) block
])
def test_sd3_step_matches_oracle(B, Hl, Wl, T, kw):
    rcfg, ref_bf, ref_32, hip = _setup(dict(kw))
    _compare_step(rcfg, ref_bf, ref_32, hip, B, Hl, Wl, T, f"tiny B={B} {Hl}x{Wl} T={T} L={rcfg.num_layers}")


def test_sd3_validation_sampler_matches_oracle():
    """CFG + flow-match Euler latent sampler over the MMDiT (the middle third of the reference's SD3.5 validate(),
    train_sd35.py:129-142): HIP model vs the oracle in bf16 and fp32 from the same initial latents, a few steps on a tiny
    configuration with the SD3.5 scheduler shift; same bar as the SANA sampler."""
    from oracle.recipe_ref import FlowMatchSchedule as RefSched
    from oracle.sd3_ref import sample_latents_sd3_ref
    from yat_amd.sampler import sample_latents_sd3
    from yat_amd.scheduler import FlowMatchSchedule
    rcfg, ref_bf, ref_32, hip = _setup({})
    g = torch.Generator().manual_seed(21)
    B, T, Hl, Wl = 2, 10, 12, 8
    x0 = torch.randn(B, rcfg.in_channels, Hl, Wl, generator=g).to(BF)
    enc, neg = (torch.randn(B, T, rcfg.joint_attention_dim, generator=g).to(BF) for _ in range(2))
    pool, npool = (torch.randn(B, rcfg.pooled_projection_dim, generator=g).to(BF) for _ in range(2))
    steps = 4
    out = sample_latents_sd3(hip, enc, pool, neg, npool, Hl, Wl, num_inference_steps=steps, guidance_scale=5.0, latents=x0,
                             schedule=FlowMatchSchedule(shift=3.0))
    o_bf = sample_latents_sd3_ref(ref_bf, RefSched(shift=3.0), x0, enc, pool, neg, npool, steps, 5.0, BF)
    o_32 = sample_latents_sd3_ref(ref_32, RefSched(shift=3.0), x0, enc, pool, neg, npool, steps, 5.0, torch.float32)
    e_hip, e_ref = rel(out, o_32), rel(o_bf, o_32)
    print(f"[parity] sd3 sampler: hip_vs_fp32={e_hip:.3e} oracle_bf16_vs_fp32={e_ref:.3e} hip_vs_oracle_bf16={rel(out, o_bf):.3e}")
    assert torch.isfinite(out.float()).all() and out.shape == x0.shape
    assert e_hip <= 1.15 * e_ref + 2e-3
    # without guidance: one conditional pass per step, no negative branch
    out1 = sample_latents_sd3(hip, enc, pool, None, None, Hl, Wl, num_inference_steps=2, guidance_scale=1.0, latents=x0)
    assert torch.isfinite(out1.float()).all() and out1.shape == x0.shape


def test_sd3_real_width_blocks_match_oracle():
    """SD3.5-Medium width (D = 1536 = 24 x 64, text width 4096, pooled 2048, T = 333 = 77 + 256 prompt tokens): one dual
    block + the context_pre_only block, 32 x 48 latents (384 image tokens per image, joint sequence 717)."""
    rcfg, ref_bf, ref_32, hip = _setup(dict(_full=True, num_layers=2, dual_attention_layers=(0,), pos_embed_max_size=64))
    _compare_step(rcfg, ref_bf, ref_32, hip, 2, 32, 48, 333, "real width", check_adamw=False)


def test_sd3_real_width_blocks_at_1024px_token_count():
    """BASELINE config 4 at its own sequence length: SD3.5-Medium width, one dual block + the context_pre_only block, 128 x 128
    latents = 4096 image tokens, T = 333 -> joint sequence 4429 (train_sd35.py:165-194).  B = 1 against the CPU oracle in bf16
    and fp32 (same criteria as every step test), then B = 2 twice: finite everywhere and run-to-run bit-identical (streams,
    fixed-order reductions at 8858 joint rows)."""
    from yat_amd.recipe import SD3Recipe
    rcfg, ref_bf, ref_32, hip = _setup(dict(_full=True, num_layers=2, dual_attention_layers=(0,)))
    _compare_step(rcfg, ref_bf, ref_32, hip, 1, 128, 128, 333, "real width, 128x128 latents (L = 4429)", check_adamw=False)
    del ref_bf, ref_32
    g = torch.Generator().manual_seed(9)
    latents = (torch.randn(2, rcfg.in_channels, 128, 128, generator=g) * 0.5).to(BF)
    emb = (torch.randn(2, 333, rcfg.joint_attention_dim, generator=g).to(BF), torch.randn(2, rcfg.pooled_projection_dim, generator=g).to(BF))
    recipe = SD3Recipe(hip, device=DEV)

    def run():
        loss, pred, _ = recipe.optimize(latents, emb, torch.Generator().manual_seed(5), return_pred=True)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), pred.detach().clone(), hip.flat_grad.clone()
    a, b = run(), run()
    assert hip._saved.L == 4429 and hip._saved.M == 2 * 4096
    for x in a:
        assert torch.isfinite(x.float()).all()
    for x, y in zip(a, b):
        assert torch.equal(x, y), "two runs of the same step differ: stream race or non-deterministic reduction"
    print(f"[parity] sd3 real width, B=2, 128x128 latents (L = 4429): loss {a[0].item():.6f}, finite, run-to-run bit-identical")


def test_sd3_checkpoint_roundtrip_and_determinism(tmp_path):
    from yat_amd.sd3 import SD3Transformer2DModelHIP
    from yat_amd.recipe import SD3Recipe
    from safetensors.torch import load_file
    rcfg, ref_bf, _, hip = _setup({})
    assert set(hip.state_dict()) | {"pos_embed.pos_embed"} == set(ref_bf.state_dict())
    hip.save_pretrained(str(tmp_path / "m"))
    saved = load_file(str(tmp_path / "m" / "diffusion_pytorch_model.safetensors"))
    assert torch.equal(saved["pos_embed.pos_embed"], ref_bf.state_dict()["pos_embed.pos_embed"])     # the persistent table
    again = SD3Transformer2DModelHIP.from_pretrained(str(tmp_path / "m"), device=DEV)
    assert torch.equal(again.flat_param, hip.flat_param)
    g = torch.Generator().manual_seed(1)
    latents = (torch.randn(2, rcfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
    emb = (torch.randn(2, 9, rcfg.joint_attention_dim, generator=g).to(BF), torch.randn(2, rcfg.pooled_projection_dim, generator=g).to(BF))
    recipe = SD3Recipe(hip, device=DEV)

    def run():
        loss, pred, _ = recipe.optimize(latents, emb, torch.Generator().manual_seed(5), return_pred=True)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), pred.detach().clone(), hip.flat_grad.clone()
    a, b = run(), run()
    for x, y in zip(a, b):
        assert torch.equal(x, y), "two runs of the same step differ: stream race or non-deterministic reduction"
    # gradient accumulation adds
    hip.accumulate_grads = True
    c = run()
    hip.accumulate_grads = False
    assert rel(c[2], 2 * a[2].float()) <= 6e-3


def test_sd3_trainer_runs_from_shards(tmp_path, monkeypatch):
    """train_sd35.py end to end on a tiny MMDiT: shards carrying emb.pt + pooled.pt -> bucket sampler -> SD35Trainer.optimize
    -> backward -> clip + AdamW -> checkpoint in the diffusers layout."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from train_sd35 import SD35Trainer
    from yat_amd.common.shards import write_shard
    from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sd3 import SD3Config
    cfg = SD3Config(sample_size=16, in_channels=8, out_channels=8, num_layers=2, attention_head_dim=64, num_attention_heads=2,
                    joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64, pos_embed_max_size=24,
                    dual_attention_layers=(0,))
    g = torch.Generator().manual_seed(0)
    samples = []
    for i in range(24):
        r = ["1.0", "0.5", "2.0"][i % 3]
        Hpx, Wpx = ASPECT_RATIO_1024_BIN[r]
        samples.append(dict(__key__=f"{i:07d}", ratio=r,
                            latent=(torch.randn(cfg.in_channels, int(Hpx) // 128 * 2, int(Wpx) // 128 * 2, generator=g) * 0.5).to(BF),
                            emb=torch.randn(11, cfg.joint_attention_dim, generator=g).to(BF),
                            pooled=torch.randn(cfg.pooled_projection_dim, generator=g).to(BF)))
    path = str(tmp_path / "shard-000000.tar")
    write_shard(path, samples)
    (tmp_path / "config.yaml").write_text("\n".join([
        "urls:", "  - unused", "local_shard_paths:", f"  - {path}", "num_shards: 1", "dataset_seed: 3", "batch_size: 4",
        "learning_rate: 1e-3", "steps: 3", "num_steps_per_validation: 2", "validation_prompts:", "  - x", "bfloat16: true",
        "aspect_ratio: 1024", ""]))
    # cached validation embeddings (what pipe.encode_prompt returns, train_sd35.py:116-118): the validation pass samples
    # latents from them (validate(): 20 Euler steps with CFG 5.0) at every validation step
    torch.save([(torch.randn(1, 11, cfg.joint_attention_dim, generator=g).to(BF),
                 torch.randn(1, 11, cfg.joint_attention_dim, generator=g).to(BF),
                 torch.randn(1, cfg.pooled_projection_dim, generator=g).to(BF),
                 torch.randn(1, cfg.pooled_projection_dim, generator=g).to(BF))], tmp_path / "validation_embeds.pt")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("YAT_TENSORBOARD", "0")
    params = TrainingParameters()
    params.read_yaml(str(tmp_path / "config.yaml"))
    trainer = SD35Trainer(params, config=cfg)
    before = trainer.model.flat_param.clone()
    trainer.run()
    torch.cuda.synchronize()
    losses = [float(l) for l in trainer.loss_history]
    assert len(losses) == 3 and all(l == l for l in losses)
    assert not torch.equal(before, trainer.model.flat_param)
    ck = tmp_path / "models" / "2"
    assert (ck / "config.json").exists() and (ck / "diffusion_pytorch_model.safetensors").exists()
    lat = torch.load(ck / "validation_latents.pt")
    assert len(lat) == 1 and lat[0].shape == (1, cfg.in_channels, cfg.sample_size, cfg.sample_size)
    assert torch.isfinite(lat[0].float()).all()


@pytest.mark.parametrize("algo", ["lora", "lokr"])
def test_sd3_adapter_step_matches_oracle(algo):
    """PEFT adapters on the MMDiT (the reference wraps whatever model the entry point trains, common/trainer.py:212-241): the
    model's lin / dgrad / wgrad hooks are those of SANA and PixArt.  One adapted step on the tiny configuration -- targets incl.
    the fused q|k|v and add_q|k|v projections, both FFNs' ``proj`` layers, the AdaLN ``linear``s and the patch embedding --
    against the oracle's peft-wrapped model in bf16 and fp32."""
    from oracle.sd3_ref import optimize_ref
    from oracle.recipe_ref import FlowMatchSchedule as RefSched
    from yat_amd.recipe import SD3Recipe
    targets = ["to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out", "linear_1",
               "linear_2", "proj", "linear"]
    rcfg, ref_bf, _, hip = _setup({})
    g = torch.Generator().manual_seed(21)
    if algo == "lora":
        from oracle.lora_ref import apply_lora as apply, LoRAWrapped as Wrapped
        from yat_amd.lora import LoRAAdapters
        ad = LoRAAdapters(hip, targets, r=4, alpha=4.0)
        for e in ad.entries:
            _, bt = ad._views(e, ad.flat_param)
            bt[:4].copy_((torch.randn(4, e["out"], generator=g) * 0.05).to(BF))
        names = ("lora_A.weight", "lora_B.weight")
        attrs = ("lora_A", "lora_B")
        wrapped = apply(ref_bf, targets, r=4, alpha=4.0)
    else:
        from oracle.lokr_ref import apply_lokr as apply, LoKrWrapped as Wrapped
        from yat_amd.lokr import LoKrAdapters
        with pytest.raises(NotImplementedError, match="convolution"):       # 'proj' also names the 2x2 patch embedding
            LoKrAdapters(hip, targets, r=2, alpha=4.0)
        targets = [t if t != "proj" else "net.0.proj" for t in targets]
        ad = LoKrAdapters(hip, targets, r=2, alpha=4.0)
        for e in ad.entries:
            w1 = ad._views(e, ad.flat_param)[0]
            w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
        names = attrs = ("lokr_w1", "lokr_w2_a", "lokr_w2_b")
        wrapped = apply(ref_bf, targets, r=2, alpha=4.0)
    assert sorted(wrapped) == sorted(e["module"] for e in ad.entries)
    assert "transformer_blocks.0.norm1.linear" in wrapped and "transformer_blocks.0.ff_context.net.0.proj" in wrapped
    sd = ad.state_dict()
    for name, w in wrapped.items():
        for k, a in zip(names, attrs):
            with torch.no_grad():
                getattr(w, a).copy_(sd[f"base_model.model.{name}.{k}"].cpu())
    ref_32 = copy.deepcopy(ref_bf).float()
    latents = (torch.randn(2, rcfg.in_channels, 12, 8, generator=g) * 0.5).to(BF)
    prompt = torch.randn(2, 10, rcfg.joint_attention_dim, generator=g).to(BF)
    pooled = torch.randn(2, rcfg.pooled_projection_dim, generator=g).to(BF)
    outs = {}
    for tag, model, dt in (("bf16", ref_bf, BF), ("fp32", ref_32, torch.float32)):
        model.train()
        loss, pred, _ = optimize_ref(model, RefSched(), latents, prompt, pooled, torch.Generator().manual_seed(7), dt)
        loss.backward()
        outs[tag] = (pred.detach(), {n: [getattr(m, a).grad for a in attrs] for n, m in model.named_modules()
                                     if isinstance(m, Wrapped)})
    recipe = SD3Recipe(hip, device=DEV)
    hip.train()
    base = hip.flat_param.clone()
    loss, pred, _ = recipe.optimize(latents, (prompt, pooled), torch.Generator().manual_seed(7), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    e_h, e_r = rel(pred, outs["fp32"][0]), rel(outs["bf16"][0], outs["fp32"][0])
    print(f"[parity] sd3 {algo} pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e}")
    assert e_h <= 1.15 * e_r + 1e-3
    hg, bg, fg = [], [], []
    for e in ad.entries:
        views = ad._views(e, ad.flat_grad)
        if algo == "lora":
            mine = [views[0][:4], views[1][:4].t()]
        else:
            mine = list(views)
        hg += [t.float().flatten().cpu() for t in mine]
        bg += [t.float().flatten() for t in outs["bf16"][1][e["module"]]]
        fg += [t.float().flatten() for t in outs["fp32"][1][e["module"]]]
    hg, bg, fg = torch.cat(hg), torch.cat(bg), torch.cat(fg)
    e_h, e_r = rel(hg, fg), rel(bg, fg)
    print(f"[parity] sd3 {algo} adapter grads hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e} (n={hg.numel()})")
    assert torch.isfinite(hg).all() and fg.abs().max() > 0 and e_h <= 1.15 * e_r + 2e-3
    assert torch.equal(base, hip.flat_param)


def test_sd3_launch_plan_replay_and_chains_are_bit_identical():
    """The MMDiT's device path (``SD3Recipe.train_step_device`` -> ``forward_device`` / ``backward_device``; the step is
    train_sd35.py:165-194): (a) eight optimizer steps alternating between two buckets with new inputs and draws every step
    give bit-identical losses and parameters with launch plans on and off; (b) the forward as two image-range chains on two
    streams equals the single-chain forward bit for bit (nothing in the MMDiT mixes images: the joint attention is per image)."""
    from yat_amd.optim import FlatAdamW
    from yat_amd.recipe import SD3Recipe
    from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP
    cfg = SD3Config(sample_size=16, patch_size=2, in_channels=8, out_channels=8, num_layers=3, attention_head_dim=64,
                    num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64,
                    pos_embed_max_size=24, dual_attention_layers=(0, 1))
    B, T = 4, 10
    runs = []
    for plans, chains in ((True, 2), (False, 2), (True, 1)):
        hip = SD3Transformer2DModelHIP(cfg, device=DEV).init_synthetic(4)
        hip.use_plans, hip.fwd_chains = plans, chains
        opt = FlatAdamW(hip, lr=1e-3, weight_decay=0.01, overlap_update=True)
        recipe = SD3Recipe(hip, device=DEV)
        g = torch.Generator().manual_seed(9)
        shapes = ((8, 16), (12, 8))
        lat = [torch.empty(B, cfg.in_channels, h, w, dtype=BF, device=DEV) for h, w in shapes]       # persistent per bucket
        noise = [torch.empty_like(t) for t in lat]
        prompt = torch.empty(B, T, cfg.joint_attention_dim, dtype=BF, device=DEV)
        pooled = torch.empty(B, cfg.pooled_projection_dim, dtype=BF, device=DEV)
        t_dev, s_dev = torch.empty(B, device=DEV), torch.empty(B, dtype=BF, device=DEV)
        loss_dev = torch.zeros(1, device=DEV)
        losses = []
        for step in range(8):
            k = step % 2
            lat[k].copy_((torch.randn(lat[k].shape, generator=g) * 0.5).to(BF))
            noise[k].copy_(torch.randn(lat[k].shape, generator=g).to(BF))
            prompt.copy_(torch.randn(prompt.shape, generator=g).to(BF))
            pooled.copy_(torch.randn(pooled.shape, generator=g).to(BF))
            _, t, sig = recipe.scheduler.sample(B, torch.Generator().manual_seed(100 + step))
            t_dev.copy_(t); s_dev.copy_(sig)
            recipe.train_step_device(lat[k], prompt, pooled, noise[k], t_dev, s_dev, loss_dev)
            losses.append(loss_dev.clone())
            opt.step()
        hip.join_pending_update()
        torch.cuda.synchronize()
        runs.append((torch.cat(losses).cpu(), hip.flat_param.clone(), getattr(hip, "plan_replays", 0), len(hip._plans)))
    (l_a, p_a, replays, nplans), (l_b, p_b, r_b, n_b), (l_c, p_c, _, _) = runs
    print(f"[plans] sd3: {nplans} plans recorded, {replays} replays; losses {l_a.tolist()}")
    assert r_b == 0 and n_b == 0
    assert replays >= 2 * 4 and nplans <= 8
    assert torch.isfinite(l_a).all() and torch.equal(l_a, l_b) and torch.equal(p_a, p_b), "launch plans on vs off differ"
    assert torch.equal(l_a, l_c) and torch.equal(p_a, p_c), "two forward chains vs one differ"


def test_sd3_device_path_equals_autograd_path():
    """``SD3Recipe.optimize_device`` (what ``SD35Trainer.optimize`` runs when training) against ``optimize`` +
    ``loss.backward()`` on the same host batch and global RNG state (train_sd35.py:180,182): loss and every gradient
    bit-identical, also on a replayed plan; ``gscale`` scales the gradient."""
    from yat_amd.recipe import SD3Recipe
    from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP
    cfg = SD3Config(sample_size=16, patch_size=2, in_channels=8, out_channels=8, num_layers=3, attention_head_dim=64,
                    num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64,
                    pos_embed_max_size=24, dual_attention_layers=(0, 1))
    hip = SD3Transformer2DModelHIP(cfg, device=DEV).init_synthetic(4)
    recipe = SD3Recipe(hip, device=DEV)
    g = torch.Generator().manual_seed(3)
    latents = (torch.randn(3, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
    embs = [(torch.randn(10, cfg.joint_attention_dim, generator=g).to(BF), torch.randn(cfg.pooled_projection_dim, generator=g).to(BF))
            for _ in range(3)]

    def seeded(fn):
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        out = fn()
        torch.cuda.synchronize()
        return out.detach().clone(), hip.flat_grad.clone()

    def autograd_path():
        loss = recipe.optimize(latents, embs, None)
        loss.backward()
        return loss
    la, ga = seeded(autograd_path)
    for rep in range(3):
        r0 = getattr(hip, "plan_replays", 0)
        ld, gd = seeded(lambda: recipe.optimize_device(latents, embs, None))
        assert ld.dtype == BF and torch.equal(la, ld) and torch.equal(ga, gd), rep
    assert hip.plan_replays - r0 == 2
    _, gh = seeded(lambda: recipe.optimize_device(latents, embs, None, gscale=0.5))
    assert rel(gh, 0.5 * ga.float()) <= 8e-3
