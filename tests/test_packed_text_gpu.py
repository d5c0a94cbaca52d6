"""Packed text rows (GPU): the text side of the SANA step without the reference's padding rows.

train_sana.py:168-176 pads every prompt to 512 rows; the padding goes through the caption projection and every block's K / V
projection and is then masked with the -10000 bias (patched_sana_transformer.py:275-277) -- probability exactly 0, gradient
exactly 0.  The HIP step can therefore keep the prompts' rows back to back (``yat_pack_mask`` + ``yat_sdpa_*_packed``,
``SanaTransformer2DModelHIP.forward_impl(kv_off=...)``).  Held here:

* ``yat_pack_mask`` against ``yat_pad_mask`` (same mask / bias / kv_len, rows = the source rows then zeros): bit-exact;
* packed attention against the padded entry points on the same keys: out, lse, dQ and the real rows of dK / dV bit-exact,
  rows outside the images' ranges untouched;
* the whole step, packed against padded, same weights / noise / timesteps: loss and prediction bit-identical, every gradient
  that does not sum over text rows bit-identical, the text-side weight gradients (sums over the text rows: the summation is
  cut into different K tiles) within 2e-3 relative L2 -- and the padded path is the one the oracle tests pin.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def test_pack_mask_matches_pad_mask():
    from yat_amd import ops
    B, T, C = 3, 64, 48
    lens = [7, 40, 1]
    g = torch.Generator().manual_seed(0)
    src = torch.randn(sum(lens), C, generator=g).to(BF).to(DEV)
    offs = torch.tensor([0, 7, 47, 48], dtype=torch.int32, device=DEV)
    enc = torch.empty(B, T, C, dtype=BF, device=DEV)
    m0, b0, k0 = (torch.empty(B, T, dtype=torch.int64, device=DEV), torch.empty(B, T, device=DEV),
                  torch.empty(B, dtype=torch.int32, device=DEV))
    ops.pad_mask(src, offs, B, T, C, enc, m0, b0, k0)
    for rows_padded in (48, 256, 64):          # fewer rows than B * T, more, and the prompts' rows exactly + a short tail
        packed = torch.full((rows_padded, C), 7.0, dtype=BF, device=DEV)
        m1, b1, k1 = torch.zeros_like(m0), torch.full_like(b0, 5.0), torch.zeros_like(k0)
        ops.pack_mask(src, offs, B, T, C, packed, m1, b1, k1)
        assert torch.equal(packed[:48], src) and not packed[48:].any()
        assert torch.equal(m0, m1) and torch.equal(b0, b1) and torch.equal(k0, k1)
        for b, L in enumerate(lens):
            assert torch.equal(packed[int(offs[b]):int(offs[b]) + L], enc[b, :L])


@pytest.mark.parametrize("B,N,T,H,dh,lens", [
    (3, 100, 128, 2, 32, [7, 128, 65]),
    (2, 200, 512, 3, 112, [300, 20]),           # SANA's cross-attention head
    (4, 64, 64, 1, 64, [1, 64, 33, 2]),
])
def test_packed_attention_is_bit_identical(B, N, T, H, dh, lens):
    from yat_amd import ops
    D = H * dh
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * N, D, generator=g).to(BF).to(DEV)
    kv = torch.randn(B * T, 2 * D, generator=g).to(BF).to(DEV)
    do = torch.randn(B * N, D, generator=g).to(BF).to(DEV)
    bias = torch.zeros(B, T, device=DEV)
    for b, L in enumerate(lens):
        bias[b, L:] = -9984.0
    kvl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    rows = -(-offs[-1] // 256) * 256
    kvp = torch.full((rows, 2 * D), 3.0, dtype=BF, device=DEV)          # rows of no image: must never be read as keys
    for b, L in enumerate(lens):
        kvp[offs[b]:offs[b] + L] = kv[b * T:b * T + L]
    kv_off = torch.tensor(offs[:B], dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(dh)
    work = ops.kv_work_list(lens, T, DEV)

    def run(packed, use_work):
        out, lse = torch.empty(B * N, D, dtype=BF, device=DEV), torch.empty(B, H, N, device=DEV)
        k_, v_ = (kvp[:, :D], kvp[:, D:]) if packed else (kv[:, :D], kv[:, D:])
        ko = kv_off if packed else None
        ops.sdpa_fwd(q, k_, v_, B, N, T, H, dh, scale, bias, kvl, out, lse, kv_off=ko)
        dq = torch.empty_like(q)
        dkv = torch.full_like(kvp if packed else kv, 9.0)
        delta = torch.empty(B, H, N, device=DEV)
        ops.sdpa_bwd(q, k_, v_, B, N, T, H, dh, scale, bias, kvl, out, do, lse, delta, dq, dkv[:, :D], dkv[:, D:],
                     work=work if use_work else None, kv_off=ko)
        return out, lse, dq, dkv

    for use_work in (True, False):
        o0, l0, dq0, dkv0 = run(False, use_work)
        o1, l1, dq1, dkv1 = run(True, use_work)
        assert torch.equal(o0, o1) and torch.equal(l0, l1) and torch.equal(dq0, dq1)
        for b, L in enumerate(lens):
            assert torch.equal(dkv0[b * T:b * T + L], dkv1[offs[b]:offs[b] + L])
            assert not dkv0[b * T + L:(b + 1) * T].any()                 # padded layout: masked keys get exact zeros
        assert (dkv1[offs[-1]:] == 9.0).all()                            # packed layout: rows of no image are left alone


@pytest.mark.parametrize("B,h,w,lens,pad_to,layers,modified", [
    (3, 6, 10, [7, 40, 1], 64, 2, []),
    (2, 16, 8, [100, 33], 128, 3, [1]),
    (4, 8, 8, [128, 128, 128, 128], 128, 2, []),          # no padding at all: both layouts hold the same rows
])
def test_packed_step_matches_padded_step(B, h, w, lens, pad_to, layers, modified, monkeypatch):
    from oracle.sana_ref import SanaConfig as RefCfg
    from yat_amd.recipe import SanaRecipe
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    rcfg = RefCfg.tiny(num_layers=layers, modified_blocks=list(modified))
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    g = torch.Generator().manual_seed(5)
    latents = (torch.randn(B, rcfg.in_channels, h, w, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in lens]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("YAT_TEXT_PACK", mode)
        hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV).init_synthetic(4)
        recipe = SanaRecipe(hip, pad_to=pad_to, device=DEV)
        assert recipe.packs_text(lens) == (mode == "1")
        losses = [recipe.optimize_device(latents, embs, torch.Generator().manual_seed(77)) for _ in range(2)]   # 2nd: plan replay
        torch.cuda.synchronize()
        assert torch.equal(losses[0], losses[1])
        pred = hip._buf("pred", (B, rcfg.out_channels, h * w)).clone()
        res[mode] = (losses[1].clone(), pred, {k: v.clone() for k, v in hip.G.items()})
        assert (hip._saved.kv_off is not None) == (mode == "1")
    (l0, p0, g0), (l1, p1, g1) = res["0"], res["1"]
    assert torch.equal(l0, l1) and torch.equal(p0, p1)
    text_side = ("caption_projection", "caption_norm", "attn2.to_k", "attn2.to_v")
    worst = 0.0
    for k in g0:
        a, b = g0[k].float(), g1[k].float()
        if any(t in k for t in text_side):
            r = ((a - b).norm() / a.norm().clamp_min(1e-20)).item()
            worst = max(worst, r)
            assert r <= 2e-3, (k, r)
        else:
            assert torch.equal(g0[k], g1[k]), k
    print(f"[parity] packed vs padded step: loss / prediction / image-side gradients bit-identical; "
          f"text-side weight gradients within {worst:.2e} rel L2")


def test_accumulation_with_changing_packed_rows_and_plans(monkeypatch):
    """Gradient accumulation x packed text x launch plans (common/trainer.py:317,343-346: ``accelerator.accumulate``): the
    second micro-step REPLAYS the forward plan (recorded for the first micro-step's row count) and then RECORDS its backward
    plan, because ``accumulate_grads`` flipped.  The recorded backward must run over this batch's text rows, not over the
    rows of the batch the forward was recorded with: flat gradients with plans on are bit-equal to plans off, over two
    accumulation windows whose four batches all have different packed row counts (more rows and fewer rows than recorded)."""
    from oracle.sana_ref import SanaConfig as RefCfg
    from yat_amd.recipe import SanaRecipe
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    rcfg = RefCfg.tiny(num_layers=2)
    kw = {k: getattr(rcfg, k) for k in SanaConfig.__dataclass_fields__}
    B, h, w, pad_to = 2, 6, 10, 512
    g = torch.Generator().manual_seed(11)
    batches = []
    for lens in ([100, 90], [300, 310], [400, 500], [20, 30]):        # packed rows 256, 768, 1024, 256 (of 1024 padded)
        lat = (torch.randn(B, rcfg.in_channels, h, w, generator=g) * 0.5).to(BF)
        batches.append((lat, [torch.randn(L, rcfg.caption_channels, generator=g).to(BF) for L in lens]))
    res = {}
    for plans in ("1", "0"):
        monkeypatch.setenv("YAT_LAUNCH_PLANS", plans)
        hip = SanaTransformer2DModelHIP(SanaConfig(**kw), device=DEV).init_synthetic(4)
        assert hip.use_plans == (plans == "1")
        recipe = SanaRecipe(hip, pad_to=pad_to, device=DEV)
        grads, losses = [], []
        for i, (lat, embs) in enumerate(batches):
            hip.accumulate_grads = bool(i & 1)                         # windows of two micro-steps
            losses.append(recipe.optimize_device(lat, embs, torch.Generator().manual_seed(70 + i), gscale=0.5).clone())
            assert hip._saved.kv_off is not None and hip._saved.Mt == recipe.packed_rows(sum(e.shape[0] for e in embs))
            if i & 1:
                torch.cuda.synchronize()
                grads.append(hip.flat_grad.clone())
        if plans == "1":
            assert getattr(hip, "plan_replays", 0) >= 5                # fwd x3, bwd x2 replayed
        res[plans] = (losses, grads)
    for a, b in zip(res["1"][0], res["0"][0]):
        assert torch.equal(a, b)
    for wi, (a, b) in enumerate(zip(res["1"][1], res["0"][1])):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b), f"window {wi}: {(a != b).sum().item()} gradient elements differ between plans on / off"
    print("[parity] accumulation x packed rows x plans: flat gradients of both windows bit-equal with plans on / off")
