"""Depth parity (GPU): the FULL SANA-1.6B stack -- 20 blocks at the real width (D = 2240, 70 x 32 + 20 x 112 heads, FFN 5600,
caption width 2304, T = 512 ragged) -- on the bench's buckets, HIP vs the CPU oracle in bf16 AND fp32 on fixed seeds.

Follows the reference's stack loop (utils/patched_sana_transformer.py:301-340): the residual stream after blocks 1 / 5 / 10 /
20 is compared through the oracle's per-block ``taps``, so error growth with depth is measured rather than assumed, and a
regression in one block shows at that block rather than only in the loss.  Criteria are the ones of tests/test_sana_gpu.py
(slack tightened in round 6 from 1.3 / 1.5 to what is measured -- ratios 0.95 .. 1.04 on every tap):
|loss_hip - loss_fp32| <= 1.1 |loss_oracle_bf16 - loss_fp32| + 1e-3 |loss_fp32|;
rel_l2(hip, fp32) <= 1.1 rel_l2(oracle_bf16, fp32) + 1e-3 for the prediction, the concatenated gradient and every tap
(gradient buckets: 1.2 x + 2e-3).  And, new in round 6, an ASSERTED bound on the distance to the reference's own arithmetic,
rel_l2(hip, oracle_bf16): two independent bf16 evaluations would sit sqrt(2) x the common error apart; the HIP path keeps the
reference's rounding points and sits 0.75 .. 1.06 x the oracle's own fp32 distance from it -- held to <= 1.3 x that distance
and to absolute caps of 1.25 x the largest value printed on 2026-10-05 (HIP_VS_ORACLE_CAP below).
"""
import copy
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


# rel_l2(hip, oracle_bf16) caps: 1.25 x the largest value any of the five cases printed (gpurun_out of 2026-10-05:
# 5.21e-3 / 8.68e-3 / 1.126e-2 / 1.650e-2 after blocks 1 / 5 / 10 / 20, 1.808e-2 on the prediction; the LoKr step with its
# adapter init seeded: 2.462e-2 in the fused_pair arithmetic, 2.175e-2 in peft's own order (pre_add), cap 2.7e-2)
HIP_VS_ORACLE_CAP = {0: 6.5e-3, 4: 1.09e-2, 9: 1.41e-2, 19: 2.06e-2, "pred": 2.26e-2, "lokr_pred": 2.7e-2}


@pytest.fixture(scope="module")
def full_models():
    """Random SANA-1.6B weights drawn once on the GPU (seconds; 1.6 B CPU normals would take a minute) and shared by the HIP
    model and both oracle precisions."""
    from oracle.sana_ref import SanaConfig as RefCfg, SanaTransformerRef
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    hip = SanaTransformer2DModelHIP(SanaConfig(), device=DEV).init_synthetic(7)
    with torch.no_grad():
        for name, p in hip.P.items():          # gates / shifts of a trained model are O(1), not O(1/sqrt(D))
            if name.endswith("scale_shift_table") and p.shape[0] == 6:
                p.add_(0.5)
    sd = {k: v.detach().cpu() for k, v in hip.state_dict().items()}
    with torch.device("meta"):                 # no 6.4 GB fp32 default init: the weights come from the state dict
        ref_bf = SanaTransformerRef(RefCfg())
    ref_bf = ref_bf.to(BF).to_empty(device="cpu")
    ref_bf.load_state_dict(sd)
    return hip, ref_bf


# Both bench buckets that differ in kind: the square 32 x 32 one and the non-square 24 x 42 one, two images of different text
# length each (the CPU oracle's two passes over 1.6 B parameters are ~25 s of host time per case on 16 cores).
#
# The HIP side runs the path the benchmark and the trainer run -- ``SanaRecipe.optimize_device``: one packed H2D, PACKED text
# rows, launch plans on, two forward chains -- three times, so that what is compared with the oracle is a REPLAYED plan; then
# the padded autograd path (``recipe.optimize`` + ``loss.backward()``, what ``model(...)`` callers get) on the same inputs,
# held to the device path.  At this width the two are NOT bit-identical (they are at the tiny widths of
# tests/test_packed_text_gpu.py): a text-side GEMM over 512 packed rows and the same GEMM over 1024 padded rows get different
# tile / split-K choices from the shape policy (csrc/gemm.hip), i.e. another fp32 summation order for the same row, and a
# bf16 rounding that flips in K / V re-randomises every rounding downstream: the two results are two independent bf16
# evaluations of the same function, as far apart as any two are (the HIP step vs the oracle's bf16 run: 1.6e-2 at this depth;
# measured here 1.4e-2 on the prediction, 4.9e-3 on the gradients).  Where the policy happens to coincide the two paths ARE
# bit-identical at full width (B = 8 test below, buckets 24x42 and 44x22).  What pins the packed path is the oracle
# comparison that follows, not this one; this one bounds the distance at "another bf16 evaluation" (<= 2.5e-2 / 1e-2).
TEXT_SIDE = ("caption_projection", "caption_norm", "attn2.to_k", "attn2.to_v")


# Third case: the batch the benchmark times -- B = 8 images of 32 x 32 latents, eight ragged prompts of 20..300 tokens -- so the
# oracle is compared with exactly the launch shapes (M = 8192 token rows, packed text rows, both forward chains of four images)
# that bench.py's timed region runs, not with a smaller batch of the same model (~60 GB of fp32 oracle activations on the host).
# (16 x 64 and 44 x 22: the other two bench buckets -- their own depthwise-conv tile variants and row geometry -- two images each)
@pytest.mark.parametrize("h,w,lens", [(32, 32, (300, 41)), (24, 42, (41, 233)),
                                      (32, 32, (20, 300, 77, 155, 233, 41, 118, 264)), (16, 64, (190, 64)), (44, 22, (8, 277))])
def test_full_depth_step_matches_oracle(full_models, h, w, lens):
    from oracle.recipe_ref import FlowMatchSchedule as RefSched, optimize_ref
    from yat_amd.recipe import SanaRecipe
    hip, ref_bf = full_models
    cfg = ref_bf.cfg
    g = torch.Generator().manual_seed(1000 + h)
    nb = len(lens)
    latents = (torch.randn(nb, cfg.in_channels, h, w, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in lens]
    tap_blocks = (0, 4, 9, 19)

    recipe = SanaRecipe(hip, pad_to=512, device=DEV)
    assert hip.use_plans and recipe.packs_text(list(lens)) and hip.fwd_chains == 2
    for _ in range(3):           # (a workspace that grows in the first backward drops the plans once: the third call replays)
        replays0 = getattr(hip, "plan_replays", 0)
        loss = recipe.optimize_device(latents, embs, torch.Generator())
    torch.cuda.synchronize()
    assert hip.plan_replays - replays0 == 2 and hip._saved.kv_off is not None      # forward AND backward were replays
    assert hip._saved.Mt == recipe.packed_rows(sum(lens)) < nb * 512
    pred = hip._buf("pred", (nb, cfg.out_channels, h * w)).clone().view(nb, cfg.out_channels, h, w)
    taps_h = {i: hip._saved.blocks[i].x3.detach().clone().view(nb, h * w, -1) for i in tap_blocks}
    grads_dev = hip.flat_grad.detach().clone()
    grads_h = grads_dev.float().cpu()
    assert torch.isfinite(grads_h).all()
    offs, numel = hip._offset, dict(zip(hip._offset, hip._seg_numel))

    # the padded autograd path on the same inputs and draws
    loss_p, pred_p, _ = recipe.optimize(latents, embs, torch.Generator(), return_pred=True)
    loss_p.backward()
    torch.cuda.synchronize()
    assert hip._saved.kv_off is None
    dl = abs(loss_p.item() - loss.item()) / abs(loss.item())
    dp = rel(pred_p.detach(), pred)
    dg = ((hip.flat_grad.float() - grads_dev.float()).norm() / grads_dev.float().norm()).item()
    print(f"[parity] full depth {h}x{w}: device path (packed text, replayed plans) vs padded autograd path: "
          f"loss {dl:.2e}, prediction {dp:.2e}, all gradients {dg:.2e} (relative)")
    assert dl <= 1e-3 and dp <= 2.5e-2 and dg <= 1e-2

    def oracle(model, dtype):
        t0 = time.time()
        model.zero_grad(set_to_none=True)
        taps = {}
        l, p, _ = optimize_ref(model, RefSched(), latents, embs, torch.Generator(), 512, dtype, taps=taps)
        xs = {i: taps[f"block{i}"]["x_out"].detach() for i in tap_blocks}
        del taps
        l.backward()
        flat = torch.zeros(hip.numel_flat)
        for name, q in model.named_parameters():
            flat[offs[name]:offs[name] + numel[name]] = q.grad.float().flatten()
        model.zero_grad(set_to_none=True)
        print(f"[parity] full depth: oracle {dtype} fwd+bwd {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
        return l.item(), p.detach(), xs, flat

    l_b, pred_b, taps_b, g_b = oracle(ref_bf, BF)
    ref_32 = copy.deepcopy(ref_bf).float()
    l_t, pred_t, taps_t, g_t = oracle(ref_32, torch.float32)
    del ref_32

    l_h = loss.item()
    print(f"[parity] full depth {h}x{w}: HIP results below are the device path's (optimize_device: packed text, plans replayed)")
    print(f"[parity] full depth {h}x{w}: loss hip={l_h:.6f} oracle_bf16={l_b:.6f} oracle_fp32={l_t:.6f}")
    for i in tap_blocks:
        e_h, e_b, e_hb = rel(taps_h[i], taps_t[i]), rel(taps_b[i], taps_t[i]), rel(taps_h[i], taps_b[i])
        print(f"[parity] full depth {h}x{w}: residual stream after block {i + 1:2d}: hip_vs_fp32={e_h:.3e} "
              f"oracle_bf16_vs_fp32={e_b:.3e} hip_vs_oracle_bf16={e_hb:.3e}")
        assert e_h <= 1.1 * e_b + 1e-3, (i, e_h, e_b)
        assert e_hb <= 1.3 * e_b and e_hb <= HIP_VS_ORACLE_CAP[i], (i, e_hb, e_b)
    e_h, e_b, e_hb = rel(pred, pred_t), rel(pred_b, pred_t), rel(pred, pred_b)
    print(f"[parity] full depth {h}x{w}: pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_b:.3e} "
          f"hip_vs_oracle_bf16={e_hb:.3e}")
    assert abs(l_h - l_t) <= 1.1 * abs(l_b - l_t) + 1e-3 * abs(l_t)
    assert e_h <= 1.1 * e_b + 1e-3
    assert e_hb <= 1.3 * e_b and e_hb <= HIP_VS_ORACLE_CAP["pred"], (e_hb, e_b)
    den = g_t.norm().item()
    tot_h, tot_b = (grads_h - g_t).norm().item() / den, (g_b - g_t).norm().item() / den
    print(f"[parity] full depth {h}x{w}: grads (1.6 B, concatenated) hip_vs_fp32={tot_h:.3e} oracle_bf16_vs_fp32={tot_b:.3e}")
    assert tot_h <= 1.1 * tot_b + 1e-3
    # per bucket (embedders | block i): where along the depth the gradient error sits
    for bi, (lo, hi) in enumerate(hip.bucket_bounds):
        if bi in (0, 1, 5, 10, 20):
            d = g_t[lo:hi].norm().item()
            e1, e2 = (grads_h[lo:hi] - g_t[lo:hi]).norm().item() / d, (g_b[lo:hi] - g_t[lo:hi]).norm().item() / d
            print(f"[parity] full depth {h}x{w}: grads bucket {bi:2d}: hip={e1:.3e} oracle_bf16={e2:.3e}")
            assert e1 <= 1.2 * e2 + 2e-3, (bi, e1, e2)


def test_lokr_full_depth_step_matches_oracle(full_models):
    """BASELINE config 5 at the real size: LoKr rank 8 (alpha 8, the bench's targets) on the frozen SANA-1.6B base, all 20
    blocks -- loss, prediction and every adapter gradient of one training step against the oracle's restatement of the peft
    wrap (oracle/lokr_ref.py) in bf16 and fp32.  w1 is moved off its zero init so that the adapters act (a fresh LoKr adapter
    is the identity) and their gradients flow through the whole stack.  BOTH adapter arithmetics of the HIP path are held to the
    one oracle run (round 6): ``fused_pair`` (the trainer's default: the adapter term inside the base GEMM, rounded once) and
    ``pre_add`` (peft's own op order -- what the oracle and the reference compute -- selected by ``lora_fused_pair: false`` /
    ``YAT_ADAPTER_PAIR=0``)."""
    from oracle.lokr_ref import LoKrWrapped, apply_lokr
    from oracle.recipe_ref import FlowMatchSchedule as RefSched, optimize_ref
    from yat_amd.lokr import LoKrAdapters
    from yat_amd.recipe import SanaRecipe
    hip, ref_shared = full_models
    targets = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]
    ref_bf = copy.deepcopy(ref_shared)                   # the wrap replaces modules: not on the shared oracle
    cfg = ref_bf.cfg
    h = w_ = 32
    lens = (120, 37)
    hip_runs, sd = {}, None
    for pair in (True, False):
        torch.manual_seed(4077)     # the adapters' own init (peft's kaiming_uniform_) draws from the global CPU RNG: without a seed
        ad = LoKrAdapters(hip, targets, r=8, alpha=8.0, module_dropout=0.0, pair=pair)   # the numbers would depend on test order
        try:
            assert ad.pair == pair
            g = torch.Generator().manual_seed(77)
            for e in ad.entries:
                w1, _, _ = ad._views(e, ad.flat_param)
                w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
            if sd is None:
                sd = {k: v.detach().cpu().clone() for k, v in ad.state_dict().items()}
                entries = [e["module"] for e in ad.entries]
            else:
                assert all(torch.equal(v.cpu(), sd[k]) for k, v in ad.state_dict().items()), "the two adapter sets differ"
            latents = (torch.randn(len(lens), cfg.in_channels, h, w_, generator=g) * 0.5).to(BF)
            embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in lens]
            recipe = SanaRecipe(hip, pad_to=512, device=DEV)
            hip.train()
            loss, pred, _ = recipe.optimize(latents, embs, torch.Generator().manual_seed(3), return_pred=True)
            loss.backward()
            torch.cuda.synchronize()
            hip_runs[pair] = (loss.item(), pred.detach().clone(),
                              torch.cat([t.float().flatten().cpu() for e in ad.entries for t in ad._views(e, ad.flat_grad)]))
        finally:
            hip.adapters = None                          # the shared model goes back to full fine-tuning
    wrapped = apply_lokr(ref_bf, targets, r=8, alpha=8.0)
    assert sorted(wrapped) == sorted(entries) and len(wrapped) > 200
    with torch.no_grad():
        for name, w in wrapped.items():
            pre = f"base_model.model.{name}."
            w.lokr_w1.copy_(sd[pre + "lokr_w1"])
            w.lokr_w2_a.copy_(sd[pre + "lokr_w2_a"])
            w.lokr_w2_b.copy_(sd[pre + "lokr_w2_b"])
    for q in ref_bf.parameters():                        # frozen base: only the adapter factors train (peft's wrap)
        q.requires_grad_(False)
    for m in (m for m in ref_bf.modules() if isinstance(m, LoKrWrapped)):
        for q in (m.lokr_w1, m.lokr_w2_a, m.lokr_w2_b):
            q.requires_grad_(True)

    def oracle(model, dtype):
        t0 = time.time()
        model.train()
        l, p, _ = optimize_ref(model, RefSched(), latents, embs, torch.Generator().manual_seed(3), 512, dtype)
        l.backward()
        ms = {n: m for n, m in model.named_modules() if isinstance(m, LoKrWrapped)}
        flat = torch.cat([t.grad.float().flatten() for n in entries
                          for t in (ms[n].lokr_w1, ms[n].lokr_w2_a, ms[n].lokr_w2_b)])
        model.zero_grad(set_to_none=True)
        print(f"[parity] lokr full depth: oracle {dtype} fwd+bwd {time.time() - t0:.1f} s")
        return l.item(), p.detach(), flat

    l_b, p_b, g_b = oracle(ref_bf, BF)
    ref_32 = ref_bf.float()                              # (in place: the bf16 copy is not needed again)
    l_t, p_t, g_t = oracle(ref_32, torch.float32)
    del ref_32, ref_bf
    for pair, (l_h, pred, g_h) in hip_runs.items():
        form = "fused_pair" if pair else "pre_add"
        print(f"[parity] lokr full depth ({form}): loss hip={l_h:.6f} oracle_bf16={l_b:.6f} oracle_fp32={l_t:.6f}")
        assert abs(l_h - l_t) <= 1.1 * abs(l_b - l_t) + 2e-3 * abs(l_t)
        e_h, e_b, e_hb = rel(pred, p_t), rel(p_b, p_t), rel(pred, p_b)
        print(f"[parity] lokr full depth ({form}): pred hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_b:.3e} hip_vs_oracle_bf16={e_hb:.3e}")
        assert e_h <= 1.1 * e_b + 1e-3
        # (fused_pair rounds base + adapter once where the oracle's peft order rounds three times: the two bf16 evaluations
        #  share fewer rounding points than in full fine-tuning -- up to 1.3 x the oracle's own fp32 distance, measured)
        assert e_hb <= 1.55 * e_b and e_hb <= HIP_VS_ORACLE_CAP["lokr_pred"], (form, e_hb, e_b)
        e_h, e_b = rel(g_h, g_t), rel(g_b, g_t)
        print(f"[parity] lokr full depth ({form}): adapter grads ({g_t.numel() / 1e6:.2f} M) hip_vs_fp32={e_h:.3e} oracle_bf16_vs_fp32={e_b:.3e}")
        assert torch.isfinite(g_h).all() and g_t.abs().max() > 0
        assert e_h <= 1.1 * e_b + 2e-3


@pytest.mark.parametrize("pair", [True, False])
def test_lokr_config5_batch32_step_properties(full_models, pair):
    """BASELINE config 5 at ITS OWN batch (round-5 review item 7): LoKr rank 8 on the frozen SANA-1.6B base, B = 32 images of
    32 x 32 latents with prompts of 20..300 tokens, module dropout 0.05 as the bench runs it.  The oracle proves the step at
    B = 2 (test above; 32 images through the fp32 CPU oracle would take ~10 minutes and ~250 GB); here the B = 32 step is
    held to size-independent properties: finite; deterministic (same draws, same dropout pattern -> the same bits); the
    prediction of its first two images equals the B = 2 step of those two images up to "another bf16 evaluation" (images
    are independent of each other in the forward: common/trainer.py:312-344 batches them, nothing mixes them; the GEMM shape
    policy may pick another tile / split for M = 32768 rows than for 2048); and the adapter gradient of the B = 32 batch is
    not the B = 2 one (every image contributes).  Both adapter arithmetics (``pair``: DESIGN.md section 11)."""
    from yat_amd.lokr import LoKrAdapters
    from yat_amd.recipe import SanaRecipe
    hip, _ = full_models
    targets = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]
    cfg = hip.cfg
    ad = LoKrAdapters(hip, targets, r=8, alpha=8.0, module_dropout=0.05, pair=pair)
    assert ad.pair == pair
    try:
        g = torch.Generator().manual_seed(78)
        for e in ad.entries:
            w1, _, _ = ad._views(e, ad.flat_param)
            w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
        B, h, w_ = 32, 32, 32
        lens = [int(x) for x in torch.randint(20, 301, (B,), generator=g)]
        latents = (torch.randn(B, cfg.in_channels, h, w_, generator=g) * 0.5).to(BF)
        embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in lens]
        recipe = SanaRecipe(hip, pad_to=512, device=DEV)
        hip.train()

        def step(nb):
            torch.manual_seed(1234)                      # the module-dropout draws (torch.rand on the host: peft's own draw)
            ad.flat_grad.zero_()
            loss, pred, _ = recipe.optimize(latents[:nb], embs[:nb], torch.Generator().manual_seed(5), return_pred=True)
            loss.backward()
            torch.cuda.synchronize()
            dropped = [e["module"] for e in ad.entries if not e["active"]]
            return loss.item(), pred.detach().clone(), ad.flat_grad.detach().clone(), dropped
        l1, p1, g1, d1 = step(B)
        l2, p2, g2, d2 = step(B)
        assert 0 < len(d1) < len(ad.entries) // 4 and d1 == d2, "module dropout 0.05 drops a few adapters, the same ones"
        assert torch.isfinite(p1.float()).all() and torch.isfinite(g1.float()).all() and g1.float().abs().max() > 0
        assert l1 == l2 and torch.equal(p1, p2) and torch.equal(g1, g2), "the B = 32 step is not deterministic"
        l_s, p_s, g_s, d_s = step(2)
        assert d_s == d1                                  # same dropout draws: the same adapted model
        # NB the recipe draws noise for the whole batch in one call, so images 0 and 1 of the B = 32 batch see other noise /
        # timesteps than the B = 2 batch: compare through the model call itself on identical inputs instead
        hip.eval()
        with torch.no_grad():
            enc, mask, _, _ = recipe.pad_embeddings(embs)
            t = torch.linspace(40.0, 960.0, B)
            out32 = hip(latents.to(DEV), encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask).sample.clone()
            out2 = hip(latents[:2].to(DEV), encoder_hidden_states=enc[:2], timestep=t[:2], encoder_attention_mask=mask[:2]).sample
        d = rel(out32[:2], out2)
        print(f"[parity] lokr config 5 (pair={pair}): B=32 step loss {l1:.6f}, {len(d1)} of {len(ad.entries)} adapters dropped; "
              f"first two images of the B=32 forward vs the B=2 forward: {d:.2e} (relative)")
        assert d <= 2e-2
        assert rel(g1, g_s) > 1e-2
    finally:
        hip.adapters = None
        hip.train()


def test_adapter_multiplier_zero_is_the_base_model_even_after_a_recorded_plan(full_models):
    """common/trainer.py:270-281,385-397 (`rescale_adapter_scale(model, 0.0)` around validation steps outside the timestep
    whitelist) on the real path (round-5 advisor / review item 7): the base model records and replays a launch plan; adapters
    are attached (no plan may be replayed or recorded from here on: the recorded launches know nothing of them).  With the
    adapter scale multiplied by 0 the step is BIT FOR BIT the step of an identity adapter (w1 = 0: the same launches on the
    same operands -- nothing keeps a stale scale, in either adapter arithmetic), which is the base model up to "another bf16
    evaluation" (with adapters the forward is one chain and every target goes through gemm256's pair / pre_add kernels;
    without, two chains and the shape policy's own pick: other fp32 summation orders for the same rows); with the scale
    restored it is the adapted step again; with the adapters detached the base model's plans are back and still right."""
    from yat_amd.common.trainer import rescale_adapter_scale
    from yat_amd.lokr import LoKrAdapters
    from yat_amd.recipe import SanaRecipe
    hip, _ = full_models
    cfg = hip.cfg
    g = torch.Generator().manual_seed(79)
    B, h, w_ = 2, 32, 32
    latents = (torch.randn(B, cfg.in_channels, h, w_, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (33, 210)]
    recipe = SanaRecipe(hip, pad_to=512, device=DEV)
    hip.train()
    assert hip.use_plans and hip.adapters is None

    def device_pred():
        recipe.optimize_device(latents, embs, torch.Generator())
        torch.cuda.synchronize()
        return hip._buf("pred", (B, cfg.out_channels, h * w_)).clone()
    for _ in range(3):
        r0 = getattr(hip, "plan_replays", 0)
        base = device_pred()
    assert hip.plan_replays - r0 == 2                      # forward and backward of the third call were replays
    for pair in (True, False):
        ad = LoKrAdapters(hip, ["to_q", "to_k", "to_v", "to_out.0", "conv_inverted", "conv_point"], r=8, alpha=8.0,
                          module_dropout=0.0, pair=pair)
        try:
            trained = []
            for e in ad.entries:
                w1, _, _ = ad._views(e, ad.flat_param)
                w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
                trained.append(w1.clone())
            r0 = hip.plan_replays
            adapted = device_pred()
            assert hip.plan_replays == r0, "a plan was replayed with adapters attached"
            with rescale_adapter_scale(ad, 0.0):
                assert ad.scale == 0.0
                zeroed = device_pred()
            again = device_pred()                          # the trained scaling is back
            for e in ad.entries:
                ad._views(e, ad.flat_param)[0].zero_()
            ident = device_pred()                          # a fresh LoKr adapter (w1 = 0) is the identity
            for e, w in zip(ad.entries, trained):
                ad._views(e, ad.flat_param)[0].copy_(w)
            d_ad, d_base = rel(adapted, ident), rel(ident, base)
            print(f"[parity] adapter multiplier 0 (pair={pair}): adapted vs identity {d_ad:.2e}; identity-adapter step vs the "
                  f"base model's replayed plan {d_base:.2e}")
            assert torch.equal(zeroed, ident), f"multiplier 0 is not the identity adapter (pair={pair}): {rel(zeroed, ident):.2e}"
            assert torch.equal(again, adapted) and not torch.equal(adapted, zeroed) and d_ad > 1e-3
            assert d_base <= 2.5e-2
            assert hip.plan_replays == r0
        finally:
            hip.adapters = None
    # detached: the base model again, from plans (a change of the adapter scale drops recorded plans -- rescale_adapter_scale --
    # so the first call may record anew; the second one replays)
    first = device_pred()
    r0 = hip.plan_replays
    assert torch.equal(first, base) and torch.equal(device_pred(), base) and hip.plan_replays - r0 == 2


# ---- the bench step at its own width and batch, on every bench bucket, without the oracle: what bench.py times is
# ``train_step_device`` with packed text rows, replayed launch plans and two forward chains at D = 2240, B = 8 on the buckets
# 32x32 / 16x64 / 24x42 / 44x22 with prompts of 20..300 tokens (train_sana.py:163-219).  The oracle pins the padded path (tiny
# configs, one real-width block, the full-depth test above at B = 2); here the packed / planned step is held to the padded /
# unplanned one AT the bench's shapes: plans on vs off bit-identical everywhere (same launches); packed vs padded equal to
# rounding -- at this width the shape policy gives the text-side GEMMs over ~1400 packed rows another tile / split-K than
# over 4096 padded rows (another fp32 summation order for the same row; bit-identity holds at the tiny widths of
# tests/test_packed_text_gpu.py, where every GEMM takes the same kernel, and on the buckets where the policy happens to
# coincide -- measured: 24x42 and 44x22 give loss / prediction / image-side gradients identical to the bit).  Where it does
# not, the results are two independent bf16 evaluations of the same two blocks (measured 6e-3 on the prediction -- the
# distance of ANY two bf16 evaluations at this depth, cf. hip vs oracle-bf16 5e-3 in tests/test_sana_gpu.py -- and 1.4e-3
# on the gradients): loss <= 2e-4, prediction <= 1e-2, gradients <= 4e-3.
BENCH_BUCKETS = [(32, 32), (16, 64), (24, 42), (44, 22)]


@pytest.fixture(scope="module")
def wide_model():
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    hip = SanaTransformer2DModelHIP(SanaConfig(num_layers=2), device=DEV).init_synthetic(3)
    with torch.no_grad():
        for name, p in hip.P.items():
            if name.endswith("scale_shift_table") and p.shape[0] == 6:
                p.add_(0.5)
    return hip


@pytest.mark.parametrize("h,w", BENCH_BUCKETS)
def test_full_width_bench_step_packed_and_planned(wide_model, h, w, monkeypatch):
    from yat_amd.recipe import SanaRecipe
    hip = wide_model
    cfg = hip.cfg
    B = 8
    g = torch.Generator().manual_seed(1234 + h)
    lens = torch.randint(20, 301, (B,), generator=g).tolist()                 # bench.py's prompt lengths
    latents = (torch.randn(B, cfg.in_channels, h, w, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in lens]
    offs, numel = hip._offset, dict(zip(hip._offset, hip._seg_numel))

    def run(pack, plans, calls):
        monkeypatch.setenv("YAT_TEXT_PACK", pack)
        hip.use_plans = plans
        recipe = SanaRecipe(hip, pad_to=512, device=DEV)
        for _ in range(calls):
            r0 = getattr(hip, "plan_replays", 0)
            loss = recipe.optimize_device(latents, embs, torch.Generator())
        torch.cuda.synchronize()
        assert (hip._saved.kv_off is not None) == (pack == "1")
        if plans:
            assert hip.plan_replays - r0 == 2                                  # the compared result is a replay (fwd + bwd)
        return (loss.clone(), hip._buf("pred", (B, cfg.out_channels, h * w)).clone(), hip.flat_grad.clone(), hip._saved.Mt)

    try:
        l1, p1, g1, mt = run("1", True, 3)          # the bench's configuration, replayed twice
        l2, p2, g2, _ = run("1", False, 1)          # same, every launch re-derived in Python
        l0, p0, g0, mt0 = run("0", True, 3)         # the reference's padded layout
    finally:
        hip.use_plans = True
    assert mt < mt0 == B * 512 and torch.isfinite(g1.float()).all() and torch.isfinite(l1)
    assert torch.equal(l1, l2) and torch.equal(p1, p2) and torch.equal(g1, g2), "launch plans on vs off differ"
    dl = abs(l1.item() - l0.item()) / abs(l0.item())
    dp = rel(p1, p0)
    worst_img, worst_txt = 0.0, 0.0
    for k in hip.P:
        a, b = g1[offs[k]:offs[k] + numel[k]].float(), g0[offs[k]:offs[k] + numel[k]].float()
        r = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        if any(t in k for t in TEXT_SIDE):
            worst_txt = max(worst_txt, r)
        else:
            worst_img = max(worst_img, r)
    dg = ((g1.float() - g0.float()).norm() / g0.float().norm()).item()
    print(f"[parity] bench step D=2240 B=8 {h}x{w} ({sum(lens)} text rows -> {mt} packed of {mt0}): plans on == off bit for bit; "
          f"packed vs padded: loss {dl:.2e}, prediction {dp:.2e}, all gradients {dg:.2e} (worst tensor: image side "
          f"{worst_img:.2e}, text side {worst_txt:.2e})")
    assert dl <= 2e-4 and dp <= 1e-2 and dg <= 4e-3 and worst_img <= 2e-2          # (text side: to_k.bias gradients are ~0)
