"""LoHa adapters (``lora_algo: loha``, common/trainer.py:220-224) on the GPU: the Hadamard kernels bit-exact against torch's
bf16 arithmetic, then an adapted SANA training step -- loss, prediction, the four factor gradients of every target -- against
the oracle's peft-wrapped model (oracle/loha_ref.py, HadaWeight with its hand-written backward) in bf16 and fp32."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"
TARGETS = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def test_hadamard_kernels_bit_exact():
    from yat_amd import ops
    g = torch.Generator().manual_seed(0)
    a, b, dd = (torch.randn(40, 72, generator=g).to(BF).to(DEV) for _ in range(3))
    out = torch.empty_like(a)
    for scale in (1.0, 0.25, 2.0 / 3.0):
        ops.hadamard_scale(a, b, scale, out)
        ref = ((a.cpu() * b.cpu()) * torch.tensor(scale))
        assert ref.dtype == BF and torch.equal(out.cpu(), ref), scale
        t1, t2 = torch.empty_like(a), torch.empty_like(a)
        ops.hadamard_bwd(dd, a, b, scale, t1, t2)
        gsc = dd.cpu() * torch.tensor(scale)
        assert torch.equal(t1.cpu(), gsc * b.cpu()) and torch.equal(t2.cpu(), gsc * a.cpu()), scale
    # strided views (a target inside the fused q|k|v delta)
    big = torch.zeros(40, 3 * 72, dtype=BF, device=DEV)
    ops.hadamard_scale(a, b, 0.5, big[:, 72:144])
    assert torch.equal(big[:, 72:144].cpu(), (a.cpu() * b.cpu()) * torch.tensor(0.5)) and big[:, :72].abs().max() == 0


def _names(model):
    from oracle.loha_ref import LoHaWrapped
    return {n: m for n, m in model.named_modules() if isinstance(m, LoHaWrapped)}


def test_loha_training_step_matches_oracle():
    from oracle.sana_ref import SanaConfig as RefCfg, SanaTransformerRef, init_like_pretrained
    from oracle.recipe_ref import FlowMatchSchedule as RefSched, optimize_ref
    from oracle.loha_ref import apply_loha
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.loha import LoHaAdapters
    from yat_amd.optim import FlatAdamW
    rcfg = RefCfg.tiny(num_layers=2)
    ref = SanaTransformerRef(rcfg)
    init_like_pretrained(ref, 0)
    ref_bf = copy.deepcopy(ref).to(BF)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV)
    hip.load_state_dict(ref_bf.state_dict())
    ad = LoHaAdapters(hip, TARGETS, r=2, alpha=4.0)
    g = torch.Generator().manual_seed(7)
    for e in ad.entries:                                    # meaningful adapters: w2_b away from its zero init
        w2b = ad._views(e, ad.flat_param)[3]
        w2b[:ad.r].copy_((torch.randn(ad.r, e["inn"], generator=g) * 0.3).to(BF))
    wrapped = apply_loha(ref_bf, TARGETS, r=2, alpha=4.0)
    assert sorted(wrapped) == sorted(e["module"] for e in ad.entries)
    sd = ad.state_dict()
    keys = ("hada_w1_a", "hada_w1_b", "hada_w2_a", "hada_w2_b")
    for name, w in wrapped.items():
        for k in keys:
            with torch.no_grad():
                getattr(w, k).copy_(sd[f"base_model.model.{name}.{k}"].cpu())
    ref_32 = copy.deepcopy(ref_bf).float()
    latents = (torch.randn(2, rcfg.in_channels, 6, 10, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (9, 30)]
    outs = {}
    for tag, model, dt in (("bf16", ref_bf, BF), ("fp32", ref_32, torch.float32)):
        model.train()
        loss, pred, _ = optimize_ref(model, RefSched(), latents, embs, torch.Generator().manual_seed(3), pad_to=32, dtype=dt)
        loss.backward()
        outs[tag] = (loss.detach(), pred.detach(), {n: [getattr(m, k).grad for k in keys] for n, m in _names(model).items()})
    recipe = SanaRecipe(hip, pad_to=32, device=DEV)
    hip.train()
    loss, pred, _ = recipe.optimize(latents, embs, torch.Generator().manual_seed(3), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    l32, lbf, lh = float(outs["fp32"][0]), float(outs["bf16"][0]), float(loss.detach())
    print(f"[parity] loha loss hip={lh:.6f} oracle_bf16={lbf:.6f} fp32={l32:.6f}")
    assert abs(lh - l32) <= 1.15 * abs(lbf - l32) + 2e-3 * abs(l32)
    e_h, e_r = rel(pred, outs["fp32"][1]), rel(outs["bf16"][1], outs["fp32"][1])
    print(f"[parity] loha pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e}")
    assert e_h <= 1.15 * e_r + 1e-3
    hg, bg, fg = [], [], []
    for e in ad.entries:
        g1a, g1b, g2a, g2b = ad._views(e, ad.flat_grad)
        mine = [g1a[:, :ad.r], g1b[:ad.r], g2a[:, :ad.r], g2b[:ad.r]]
        assert all(t.abs().max() == 0 for t in (g1a[:, ad.r:], g1b[ad.r:], g2a[:, ad.r:], g2b[ad.r:])), "rank padding must stay zero"
        hg += [t.float().flatten().cpu() for t in mine]
        bg += [t.float().flatten() for t in outs["bf16"][2][e["module"]]]
        fg += [t.float().flatten() for t in outs["fp32"][2][e["module"]]]
    hg, bg, fg = torch.cat(hg), torch.cat(bg), torch.cat(fg)
    e_h, e_r = rel(hg, fg), rel(bg, fg)
    print(f"[parity] loha adapter grads hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e} (n={hg.numel()})")
    assert torch.isfinite(hg).all() and fg.abs().max() > 0
    assert e_h <= 1.15 * e_r + 2e-3
    before = hip.flat_param.clone()
    p0 = ad.flat_param.clone()
    FlatAdamW(ad, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0).step()
    torch.cuda.synchronize()
    assert torch.equal(before, hip.flat_param) and not torch.equal(p0, ad.flat_param)
    # checkpoint round trip in the peft layout
    sd2 = ad.state_dict()
    assert sd2["base_model.model.patch_embed.proj.hada_w1_a"].shape == (rcfg.inner_dim, 2)
    ad2 = LoHaAdapters(hip, TARGETS, r=2, alpha=4.0)
    ad2.load_state_dict(sd2)
    assert torch.equal(ad2.flat_param, ad.flat_param)


def test_loha_module_dropout_drops_the_adapter():
    from oracle.sana_ref import SanaConfig as RefCfg
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.loha import LoHaAdapters
    rcfg = RefCfg.tiny(num_layers=1)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV).init_synthetic(1)
    g = torch.Generator().manual_seed(4)
    latents = (torch.randn(2, rcfg.in_channels, 4, 6, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (5, 9)]
    recipe = SanaRecipe(hip, pad_to=16, device=DEV)
    hip.train()
    _, base_pred, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    ad = LoHaAdapters(hip, TARGETS, r=2, alpha=4.0, module_dropout=1.0 - 1e-9)
    for e in ad.entries:
        w2b = ad._views(e, ad.flat_param)[3]
        w2b[:2].copy_((torch.randn(2, e["inn"], generator=g) * 0.3).to(BF))
    loss, pred, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    assert all(not e["active"] for e in ad.entries)
    assert torch.equal(pred, base_pred) and ad.flat_grad.abs().max().item() == 0
    ad.module_dropout = 0.0
    loss2, pred2, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    assert not torch.equal(pred2, base_pred)
