"""DoRA adapters (``lora_algo: lora`` + ``lora_use_dora``, common/trainer.py:215-220) on the GPU: the row kernels against their
torch restatement, then an adapted SANA training step -- loss, prediction, the gradients of lora_A, lora_B and the magnitude
vector of every target -- against the oracle's peft-wrapped model (oracle/dora_ref.py) in bf16 and fp32."""
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


def test_dora_row_kernels_match_torch():
    from yat_amd import ops
    g = torch.Generator().manual_seed(0)
    rows, cols, scaling = 48, 200, 0.5
    W = torch.randn(rows, cols, generator=g).to(BF)
    lw = (torch.randn(rows, cols, generator=g) * 0.3).to(BF)
    mag = (torch.rand(rows, generator=g) * 20 + 5).to(BF)
    dd = torch.randn(rows, cols, generator=g).to(BF)
    u = W + torch.tensor(scaling) * lw                                   # bf16 op by op, as peft computes the norm's argument
    n = torch.linalg.norm(u.float(), dim=1).to(BF)
    s = mag / n
    delta_ref = s.float()[:, None] * (W.float() + scaling * lw.float()) - W.float()
    Wd, lwd, magd, ddd = (t.to(DEV) for t in (W, lw, mag, dd))
    delta = torch.empty_like(Wd)
    s_buf, n_buf = torch.empty(rows, dtype=torch.float32, device=DEV), torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.dora_delta(Wd, lwd, magd, scaling, delta, s_buf, n_buf)
    torch.cuda.synchronize()
    # the norm accumulates in fp32 in a different order than torch: one bf16 ulp of slack on n and s, then exact arithmetic
    assert ((n_buf.cpu() - n.float()).abs() <= 2.0 ** -7 * n.float()).all()
    assert ((s_buf.cpu() - s.float()).abs() <= 2.0 ** -6 * s.float()).all()
    d2 = s_buf.cpu()[:, None] * (W.float() + scaling * lw.float()) - W.float()
    assert torch.equal(delta.cpu(), d2.to(BF))
    assert rel(delta, delta_ref) < 1e-2
    t1, dmag = torch.empty_like(Wd), torch.empty(rows, dtype=BF, device=DEV)
    ops.dora_bwd(ddd, Wd, lwd, scaling, s_buf, n_buf, t1, dmag)
    torch.cuda.synchronize()
    dm_ref = (dd.float() * u.float()).sum(1) / n_buf.cpu()
    assert rel(dmag, dm_ref) < 5e-3
    sc = (s_buf.cpu() * scaling).to(BF)
    assert torch.equal(t1.cpu(), sc[:, None] * dd)
    # strided views: a target inside a fused q|k|v weight keeps its own rows
    big = torch.zeros(rows, 3 * cols, dtype=BF, device=DEV)
    ops.dora_delta(Wd, lwd, magd, scaling, big[:, cols:2 * cols], s_buf, n_buf)
    assert torch.equal(big[:, cols:2 * cols], delta) and big[:, :cols].abs().max() == 0


def _names(model):
    from oracle.dora_ref import DoRAWrapped
    return {n: m for n, m in model.named_modules() if isinstance(m, DoRAWrapped)}


def test_dora_training_step_matches_oracle():
    from oracle.sana_ref import SanaConfig as RefCfg, SanaTransformerRef, init_like_pretrained
    from oracle.recipe_ref import FlowMatchSchedule as RefSched, optimize_ref
    from oracle.dora_ref import apply_dora
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.dora import DoRAAdapters
    from yat_amd.optim import FlatAdamW
    rcfg = RefCfg.tiny(num_layers=2)
    ref = SanaTransformerRef(rcfg)
    init_like_pretrained(ref, 0)
    ref_bf = copy.deepcopy(ref).to(BF)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV)
    hip.load_state_dict(ref_bf.state_dict())
    ad = DoRAAdapters(hip, TARGETS, r=4, alpha=4.0)
    wrapped = apply_dora(ref_bf, TARGETS, r=4, alpha=4.0)
    assert sorted(wrapped) == sorted(e["module"] for e in ad.entries)
    # the magnitude init (row norms of W) must agree before anything is moved
    sd0 = ad.state_dict()
    for name, w in wrapped.items():
        assert torch.equal(sd0[f"base_model.model.{name}.lora_magnitude_vector.weight"].cpu(), w.magnitude.data), name
    g = torch.Generator().manual_seed(7)
    for e in ad.entries:                                    # meaningful adapters: lora_B away from zero, magnitudes off the norm
        b, _, m = ad._views(e, ad.flat_param)
        b[:, :ad.r].copy_((torch.randn(e["out"], ad.r, generator=g) * 0.05).to(BF))
        m.copy_((m.float().cpu() * (1.0 + 0.2 * torch.randn(e["out"], generator=g))).to(BF))
    sd = ad.state_dict()
    keys = (("lora_A", "lora_A.weight"), ("lora_B", "lora_B.weight"), ("magnitude", "lora_magnitude_vector.weight"))
    for name, w in wrapped.items():
        for attr, k in keys:
            with torch.no_grad():
                getattr(w, attr).copy_(sd[f"base_model.model.{name}.{k}"].cpu())
    ref_32 = copy.deepcopy(ref_bf).float()
    latents = (torch.randn(2, rcfg.in_channels, 6, 10, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (9, 30)]
    outs = {}
    for tag, model, dt in (("bf16", ref_bf, BF), ("fp32", ref_32, torch.float32)):
        model.train()
        loss, pred, _ = optimize_ref(model, RefSched(), latents, embs, torch.Generator().manual_seed(3), pad_to=32, dtype=dt)
        loss.backward()
        outs[tag] = (loss.detach(), pred.detach(), {n: [getattr(m, a).grad for a, _ in keys] for n, m in _names(model).items()})
    recipe = SanaRecipe(hip, pad_to=32, device=DEV)
    hip.train()
    base = hip.flat_param.clone()
    loss, pred, _ = recipe.optimize(latents, embs, torch.Generator().manual_seed(3), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    l32, lbf, lh = float(outs["fp32"][0]), float(outs["bf16"][0]), float(loss.detach())
    print(f"[parity] dora loss hip={lh:.6f} oracle_bf16={lbf:.6f} fp32={l32:.6f}")
    assert abs(lh - l32) <= 1.15 * abs(lbf - l32) + 2e-3 * abs(l32)
    e_h, e_r = rel(pred, outs["fp32"][1]), rel(outs["bf16"][1], outs["fp32"][1])
    print(f"[parity] dora pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e}")
    assert e_h <= 1.15 * e_r + 1e-3
    per = {"lora_A": ([], [], []), "lora_B": ([], [], []), "magnitude": ([], [], [])}
    for e in ad.entries:
        gb, ga, gm = ad._views(e, ad.flat_grad)
        assert gb[:, ad.r:].abs().max() == 0 and ga[ad.r:].abs().max() == 0, "rank padding must stay zero"
        mine = {"lora_A": ga[:ad.r], "lora_B": gb[:, :ad.r], "magnitude": gm}
        for i, (attr, _) in enumerate(keys):
            per[attr][0].append(mine[attr].float().flatten().cpu())
            per[attr][1].append(outs["bf16"][2][e["module"]][i].float().flatten())
            per[attr][2].append(outs["fp32"][2][e["module"]][i].float().flatten())
    for attr, (h, b, f) in per.items():
        h, b, f = torch.cat(h), torch.cat(b), torch.cat(f)
        e_h, e_r = rel(h, f), rel(b, f)
        print(f"[parity] dora d_{attr} hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e} (n={h.numel()})")
        assert torch.isfinite(h).all() and f.abs().max() > 0
        assert e_h <= 1.15 * e_r + 2e-3, attr
    assert torch.equal(base, hip.flat_param)
    p0 = ad.flat_param.clone()
    FlatAdamW(ad, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0).step()
    torch.cuda.synchronize()
    assert torch.equal(base, hip.flat_param) and not torch.equal(p0, ad.flat_param)
    # checkpoint round trip in the peft layout
    sd2 = ad.state_dict()
    assert sd2["base_model.model.patch_embed.proj.lora_B.weight"].shape == (rcfg.inner_dim, 4)
    ad2 = DoRAAdapters(hip, TARGETS, r=4, alpha=4.0)
    ad2.load_state_dict(sd2)
    assert torch.equal(ad2.flat_param, ad.flat_param)


def test_dora_with_zero_lora_b_and_norm_magnitude_is_the_base_model():
    """peft's init: lora_B = 0 and m = ||W|| make s = 1 and delta = 0 -- the wrapped model starts as the base model."""
    from oracle.sana_ref import SanaConfig as RefCfg
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.dora import DoRAAdapters
    rcfg = RefCfg.tiny(num_layers=1)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV).init_synthetic(1)
    g = torch.Generator().manual_seed(4)
    latents = (torch.randn(2, rcfg.in_channels, 4, 6, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (5, 9)]
    recipe = SanaRecipe(hip, pad_to=16, device=DEV)
    hip.train()
    _, base_pred, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    ad = DoRAAdapters(hip, TARGETS, r=2, alpha=4.0)
    _, pred, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    torch.cuda.synchronize()
    # s = bf16(m / n) with m = n is exactly 1, so delta = 1 * (W + 0) - W = 0 exactly
    assert ad.delta.abs().max().item() == 0 and torch.equal(pred, base_pred)
    with pytest.raises(NotImplementedError, match="lora_dropout"):
        DoRAAdapters(hip, TARGETS, r=2, alpha=4.0, dropout=0.1)
