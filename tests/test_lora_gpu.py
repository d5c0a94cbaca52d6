"""Plain LoRA adapters (``lora_algo: lora``) on the HIP path against the oracle's restatement of the peft wrap -- GPU.
peft is absent from the container: the adapter arithmetic is [RECALL] on both sides (parity unpinned, oracle/lora_ref.py)."""
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


@pytest.mark.parametrize("r,alpha", [(4, 4.0), (8, 16.0)])
def test_lora_training_step_matches_oracle(r, alpha, tmp_path):
    """One adapted training step (tiny SANA, non-zero lora_B so the adapters matter): loss / prediction / adapter gradients vs
    the oracle's bf16 and fp32 runs of the peft-wrapped model; the optimizer moves only the adapters; peft-layout round trip."""
    from oracle.sana_ref import SanaConfig as RefCfg, SanaTransformerRef, init_like_pretrained
    from oracle.recipe_ref import FlowMatchSchedule as RefSched, optimize_ref
    from oracle.lora_ref import apply_lora, LoRAWrapped
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.lora import LoRAAdapters
    from yat_amd.optim import FlatAdamW
    rcfg = RefCfg.tiny(num_layers=2)
    ref = SanaTransformerRef(rcfg)
    init_like_pretrained(ref, 0)
    ref_bf = copy.deepcopy(ref).to(BF)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV)
    hip.load_state_dict(ref_bf.state_dict())
    ad = LoRAAdapters(hip, TARGETS, r=r, alpha=alpha)
    g = torch.Generator().manual_seed(11)
    for e in ad.entries:                                    # meaningful adapters: lora_B away from its zero init
        _, bt = ad._views(e, ad.flat_param)
        bt[:r].copy_((torch.randn(r, e["out"], generator=g) * 0.05).to(BF))
    wrapped = apply_lora(ref_bf, TARGETS, r=r, alpha=alpha)
    assert sorted(wrapped) == sorted(e["module"] for e in ad.entries)
    sd = ad.state_dict()
    for name, w in wrapped.items():
        pre = f"base_model.model.{name}."
        with torch.no_grad():
            w.lora_A.copy_(sd[pre + "lora_A.weight"].cpu())
            w.lora_B.copy_(sd[pre + "lora_B.weight"].cpu())
    ref_32 = copy.deepcopy(ref_bf).float()
    B, h, w_ = 2, 6, 10
    latents = (torch.randn(B, rcfg.in_channels, h, w_, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (9, 30)]
    outs = {}
    for tag, model, dt in (("bf16", ref_bf, BF), ("fp32", ref_32, torch.float32)):
        model.train()
        loss, pred, _ = optimize_ref(model, RefSched(), latents, embs, torch.Generator().manual_seed(3), pad_to=32, dtype=dt)
        loss.backward()
        outs[tag] = (loss.detach(), pred.detach(), {n: (m.lora_A.grad, m.lora_B.grad) for n, m in model.named_modules()
                                                   if isinstance(m, LoRAWrapped)})
    recipe = SanaRecipe(hip, pad_to=32, device=DEV)
    hip.train()
    loss, pred, _ = recipe.optimize(latents, embs, torch.Generator().manual_seed(3), return_pred=True)
    loss.backward()
    torch.cuda.synchronize()
    l32, lbf, lh = float(outs["fp32"][0]), float(outs["bf16"][0]), float(loss.detach())
    print(f"[parity] lora r={r} loss hip={lh:.6f} oracle_bf16={lbf:.6f} fp32={l32:.6f}")
    assert abs(lh - l32) <= 1.15 * abs(lbf - l32) + 2e-3 * abs(l32)
    e_h, e_r = rel(pred, outs["fp32"][1]), rel(outs["bf16"][1], outs["fp32"][1])
    print(f"[parity] lora pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e}")
    assert e_h <= 1.15 * e_r + 1e-3
    hip_g, bf_g, f_g = [], [], []
    for e in ad.entries:
        ga, gbt = ad._views(e, ad.flat_grad)
        assert ga[r:].abs().sum().item() == 0 and gbt[r:].abs().sum().item() == 0          # rank padding stays untouched
        hip_g += [ga[:r].float().flatten().cpu(), gbt[:r].t().float().flatten().cpu()]
        bf_g += [t.float().flatten() for t in outs["bf16"][2][e["module"]]]
        f_g += [t.float().flatten() for t in outs["fp32"][2][e["module"]]]
    hg, bg, fg = torch.cat(hip_g), torch.cat(bf_g), torch.cat(f_g)
    e_h, e_r = rel(hg, fg), rel(bg, fg)
    print(f"[parity] lora adapter grads hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_r:.3e} (n={hg.numel()})")
    assert torch.isfinite(hg).all() and fg.abs().max() > 0
    assert e_h <= 1.15 * e_r + 2e-3
    before = hip.flat_param.clone()
    opt = FlatAdamW(ad, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    p0 = ad.flat_param.clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(before, hip.flat_param) and not torch.equal(p0, ad.flat_param)
    for e in ad.entries:                                    # the padded rank rows never move
        a, bt = ad._views(e, ad.flat_param)
        assert a[r:].abs().sum().item() == 0 and bt[r:].abs().sum().item() == 0
    ad.save_pretrained(str(tmp_path / "a"))
    from safetensors.torch import load_file
    saved = load_file(str(tmp_path / "a" / "adapter_model.safetensors"))
    again = LoRAAdapters(hip, TARGETS, r=r, alpha=alpha)
    again.load_state_dict(saved)
    assert torch.equal(again.flat_param, ad.flat_param)
    k0 = f"base_model.model.{ad.entries[0]['module']}."
    assert saved[k0 + "lora_A.weight"].shape == (r, ad.entries[0]["inn"]) and saved[k0 + "lora_B.weight"].shape == (ad.entries[0]["out"], r)



def test_lora_scatter_b_and_forward_pair():
    """Plain LoRA through the base GEMM's second operand pair (LoRAAdapters.forward_pair, the default): yat_lora_scatter_b writes
    scaling * lora_B of every adapter into the first R columns of its target's rows of the weight shadow (nothing else), and the
    adapted Linear equals the pre_add form (pair=False) and fp32 -- a fused q|k|v view with one adapter per block and a single
    target, non-zero lora_B, scaling 2."""
    from types import SimpleNamespace
    from yat_amd.lora import LoRAAdapters
    from yat_amd.lokr import adapted_linear
    D, M, r = 640, 520, 4
    names = ["blk.to_q", "blk.to_k", "blk.to_v", "blk.to_out.0"]
    g = torch.Generator().manual_seed(21)

    def make(pair):
        flat = (torch.randn(4 * D * D, generator=torch.Generator().manual_seed(5)) * D ** -0.5).to(BF).to(DEV)
        model = SimpleNamespace(P={n + ".weight": flat[i * D * D:(i + 1) * D * D].view(D, D) for i, n in enumerate(names)},
                                flat_param=flat, flat_grad=torch.zeros_like(flat))
        ad = LoRAAdapters(model, ["to_q", "to_k", "to_v", "to_out.0"], r=r, alpha=2.0 * r, pair=pair)
        assert ad.pair == pair and ad.scale == 2.0
        return model, ad
    (m1, a1), (m2, a2) = make(True), make(False)
    for e in a1.entries:
        _, bt = a1._views(e, a1.flat_param)
        bt[:r].copy_((torch.randn(r, D, generator=g) * 0.1).to(BF))
    a2.flat_param.copy_(a1.flat_param)
    a1.materialize(True); a2.materialize(True)
    shadow = a1._flatB.view(4, D, D)
    for i, e in enumerate(a1.entries):
        _, bt = a1._views(e, a1.flat_param)
        assert torch.equal(shadow[i, :, :a1.R], (bt.float().T * 2.0).to(BF)) and (shadow[i, :, a1.R:] == 0).all()
    x, bias = torch.randn(M, D, generator=g).to(BF).to(DEV), torch.randn(3 * D, generator=g).to(BF).to(DEV)

    def truth(lo, hi):
        W = m1.flat_param[lo * D * D:hi * D * D].view((hi - lo) * D, D).float()
        dW = torch.zeros_like(W)
        for j, e in enumerate(a1.entries[lo:hi]):
            a, bt = (t.float() for t in a1._views(e, a1.flat_param))
            dW[j * D:(j + 1) * D] = (bt.T @ a) * a1.scale
        return x.float() @ (W + dW).T

    def rel(a, b):
        return ((a.float() - b.float()).norm() / b.float().norm()).item()
    y1 = adapted_linear(a1, x, m1.flat_param[:3 * D * D].view(3 * D, D), bias)
    y2 = adapted_linear(a2, x, m2.flat_param[:3 * D * D].view(3 * D, D), bias)
    t = truth(0, 3) + bias.float()
    assert rel(y1, t) <= rel(y2, t) * 1.05 + 1e-4 and rel(y1, t) < 4e-3, (rel(y1, t), rel(y2, t))
    o1 = adapted_linear(a1, x, m1.P["blk.to_out.0.weight"])
    o2 = adapted_linear(a2, x, m2.P["blk.to_out.0.weight"])
    t = truth(3, 4)
    assert rel(o1, t) <= rel(o2, t) * 1.05 + 1e-4 and rel(o1, t) < 4e-3
    # the T kept for d_B is the compact product of both forms; weight gradients agree bit for bit
    dy = torch.randn(M, 3 * D, generator=g).to(BF).to(DEV)
    for m, ad in ((m1, a1), (m2, a2)):
        ad.wgrad(dy, x, m.flat_grad[:3 * D * D].view(3 * D, D))
    torch.cuda.synchronize()
    assert torch.equal(a1.flat_grad, a2.flat_grad) and a1.flat_grad.abs().max().item() > 0


@pytest.mark.parametrize("rows,R,N,ld", [(1000, 8, 2240, 2240), (333, 16, 11200, 11200), (4096, 8, 64, 192), (77, 8, 520, 520)])
def test_rank_expand(rows, R, N, ld):
    """io = bf16(bf16(h w) * scale) and io = bf16(bf16(h w) + io): the K = R products of the LoRA path, bit-exact against the
    fp64 product rounded the same way (a K <= 16 sum of bf16 products is exact in fp32 up to the last rounding)."""
    from yat_amd import ops
    g = torch.Generator().manual_seed(rows + N)
    h, w = (torch.randn(rows, R, generator=g) * 0.3).to(BF).to(DEV), (torch.randn(R, N, generator=g) * 0.2).to(BF).to(DEV)
    buf = torch.randn(rows, ld, generator=g).to(BF).to(DEV)
    prod = (h.double() @ w.double())
    mag = (h.double().abs() @ w.double().abs()).float()           # cancellation: fp32 vs fp64 sums differ by ~1e-7 of this
    out = buf.clone()
    ops.rank_expand(h, w, out[:, :N], scale=0.75)
    want = (prod.float().to(BF).float() * 0.75).to(BF)
    # a one-ulp flip of the rounded sum (fp32 vs fp64 accumulation) times 0.75 and rounded again: up to ~1.4 ulp of the result
    ok = (out[:, :N].float() - want.float()).abs() <= 2.0 ** -6 * want.float().abs() + 1e-5 * mag
    assert ok.all() and torch.equal(out[:, N:], buf[:, N:])             # one-ulp slack: fp32 vs fp64 accumulation of the sum
    out = buf.clone()
    ops.rank_expand(h, w, out[:, :N], residual=True)
    want = (prod.float().to(BF).float() + buf[:, :N].float()).to(BF)
    ok = (out[:, :N].float() - want.float()).abs() <= 2.0 ** -7 * want.float().abs() + 2.0 ** -7 * prod.abs().float() + 1e-5 * mag
    assert ok.all() and torch.equal(out[:, N:], buf[:, N:])


def test_dropout_kernel():
    """Counter-based dropout: exact arithmetic (bf16(x / (1 - p)) or 0), keep rate, determinism per seed, and the backward's
    mask is the forward's."""
    from yat_amd import ops
    n, p = 1 << 20, 0.3
    x = torch.randn(n, generator=torch.Generator().manual_seed(0)).to(BF).to(DEV)
    y = ops.dropout(x, p, seed=12345)
    keep = y != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 3e-3, frac
    want = (x.float() * (1.0 / (1.0 - p))).to(BF)
    assert torch.equal(y[keep], want[keep])
    assert torch.equal(y, ops.dropout(x, p, seed=12345)) and not torch.equal(y, ops.dropout(x, p, seed=12346))
    # neighbouring elements are uncorrelated enough for a mask: lag-1 agreement ~ (1-p)^2 + p^2
    agree = (keep[1:] == keep[:-1]).float().mean().item()
    assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 5e-3, agree
    g, io0 = torch.randn(n, generator=torch.Generator().manual_seed(1)).to(BF).to(DEV), torch.randn(n).to(BF).to(DEV)
    io = ops.dropout_bwd_add(g, p, 12345, io0.clone())
    wantb = (io0.float() + torch.where(keep, (g.float() * (1.0 / (1.0 - p))).to(BF).float(), torch.zeros_like(g, dtype=torch.float32))).to(BF)
    assert torch.equal(io, wantb)


def test_lora_dropout_forward_backward_consistent():
    """lora_dropout > 0: adapter output, input gradient and both weight gradients of one target against the fp64 formulas
    evaluated with the very mask the kernels use (extracted by running the dropout kernel on ones)."""
    from yat_amd import ops
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.lora import LoRAAdapters
    cfg = SanaConfig(num_layers=1, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=8)
    hip = SanaTransformer2DModelHIP(cfg, device=DEV).init_synthetic(0)
    r, p = 4, 0.25
    ad = LoRAAdapters(hip, ["to_out.0"], r=r, alpha=8.0, dropout=p, seed=7)
    e = next(en for en in ad.entries if en["module"] == "transformer_blocks.0.attn1.to_out.0")
    g = torch.Generator().manual_seed(5)
    a, bt = ad._views(e, ad.flat_param)
    bt[:r].copy_((torch.randn(r, e["out"], generator=g) * 0.2).to(BF))
    ad.materialize(training=True)
    M = 200
    x, dy = torch.randn(M, e["inn"], generator=g).to(BF).to(DEV), torch.randn(M, e["out"], generator=g).to(BF).to(DEV)
    dx0 = torch.randn(M, e["inn"], generator=g).to(BF).to(DEV)
    w, gw = hip.P[e["key"]], hip.G[e["key"]]
    mask = ops.dropout(torch.ones_like(x), p, ad._mask_seed(e)) != 0
    assert 0.6 < mask.float().mean().item() < 0.9
    out = ad.forward_term(x, w)
    xd = torch.where(mask, (x.float() / (1 - p)).to(BF).double(), torch.zeros_like(x, dtype=torch.float64))
    A, B, s = a[:r].double(), bt[:r].double().T, ad.scale
    T = (xd @ A.T)
    assert rel(out, (T @ B.T) * s) < 1e-2
    dx = dx0.clone()
    hs = ad.dgrad_term(dy, w, dx)
    dT = (dy.double() @ B) * s
    assert rel(dx.double() - dx0.double(), torch.where(mask, (dT @ A) / (1 - p), torch.zeros_like(xd))) < 2e-2
    ad.wgrad(dy, x, gw, accumulate=False, hs=hs)
    ga, gbt = ad._views(e, ad.flat_grad)
    assert rel(ga[:r], dT.T @ xd) < 1e-2 and rel(gbt[:r], (T.T @ dy.double()) * s) < 1e-2
    # eval mode: no mask
    ad.materialize(training=False)
    assert rel(ad.forward_term(x, w), ((x.double() @ A.T) @ B.T) * s) < 1e-2


@pytest.mark.parametrize("kind", ["lora", "lokr"])
def test_adapter_grad_accumulation_adds(kind):
    """Two micro-steps on the same batch with accumulate_grads=True on the second must give twice the adapter gradients
    (accelerator.accumulate with adapters: d_P / d_A / d_B accumulate in place, the LoKr projection runs on the sum)."""
    from oracle.sana_ref import SanaConfig as RefCfg
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.lora import LoRAAdapters
    from yat_amd.lokr import LoKrAdapters
    rcfg = RefCfg.tiny(num_layers=1)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV).init_synthetic(1)
    g = torch.Generator().manual_seed(2)
    if kind == "lora":
        ad = LoRAAdapters(hip, TARGETS, r=4, alpha=4.0)
        for e in ad.entries:
            ad._views(e, ad.flat_param)[1][:4].copy_((torch.randn(4, e["out"], generator=g) * 0.05).to(BF))
    else:
        ad = LoKrAdapters(hip, TARGETS, r=2, alpha=4.0)
        for e in ad.entries:
            w1 = ad._views(e, ad.flat_param)[0]
            w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
    latents = (torch.randn(2, rcfg.in_channels, 4, 6, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in (5, 9)]
    recipe = SanaRecipe(hip, pad_to=16, device=DEV)
    hip.train()
    recipe.optimize(latents, embs, torch.Generator()).backward()
    g1 = ad.flat_grad.clone()
    assert g1.abs().max() > 0
    hip.accumulate_grads = True
    recipe.optimize(latents, embs, torch.Generator()).backward()
    hip.accumulate_grads = False
    torch.cuda.synchronize()
    assert rel(ad.flat_grad, g1.float() * 2) <= 6e-3
