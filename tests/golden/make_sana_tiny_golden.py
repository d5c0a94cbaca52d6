#!/usr/bin/env python3
"""Writes tests/golden/sana_tiny_golden.safetensors: SURVEY.md section 8(c)(3)'s drift pin for the oracle.

One training step of the tiny SANA configuration (oracle.sana_ref.SanaConfig.tiny(): D = 64, 2 blocks, same structure as
SANA-1.6B) through the CPU oracle -- inputs, the recipe's draws, per-tap activations, prediction, loss, every gradient, the
clip norm and the parameters after one clip + AdamW step -- in the reference's bf16 dtype flow AND in fp32 on the same inputs.

What this file pins and what it does not: it is written BY the oracle (the reference ships no fixtures and its model math
lives in an absent, unpinned diffusers -- DESIGN.md section 2, "parity unpinned"), so it cannot pin the oracle to the
reference.  It pins the oracle to ITSELF: an edit to oracle/sana_ref.py / oracle/recipe_ref.py that changes an op, an op order
or a rounding point moves these numbers, and tests/test_oracle_golden.py fails instead of the parity bar moving silently.

    python tests/golden/make_sana_tiny_golden.py          # rewrites the fixture (review the diff of the printed digest)

Single-threaded, so that the bf16 flow is reproducible on one machine; the test allows for another CPU's kernels
(tests/test_oracle_golden.py).
"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sana_tiny_golden.safetensors")

H, W, LENS, PAD_TO, SEED = 4, 6, (5, 11, 16), 16, 20261005
TAPS = ("x0", "tmod", "embedded", "enc")
BLOCK_TAPS = ("h1", "attn1", "x_attn1", "x_attn2", "ff", "x_out")
LR, WD = 1e-3, 0.01          # (a visible AdamW update in bf16; weight decay on, so the decoupled-decay term is pinned too)


def inputs(cfg):
    g = torch.Generator().manual_seed(SEED)
    latents = (torch.randn(len(LENS), cfg.in_channels, H, W, generator=g) * 0.5).to(torch.bfloat16)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(torch.bfloat16) for L in LENS]
    return latents, embs


def signature(t):
    """(l2 norm, sum, sum of magnitudes) of a tensor, in fp64 -> fp32: what the fp32 flow keeps of a gradient / updated
    parameter instead of the tensor (the file stays under 1 MB; any changed element moves at least one of the three)."""
    t = t.detach().double()
    return torch.stack([t.norm(), t.sum(), t.abs().sum()]).float()


def run(dtype):
    """One step in ``dtype`` -> dict of tensors.  The bf16 flow (the reference's) is kept whole, stored as bf16 (exact);
    the fp32 flow keeps activations whole and gradients / updated parameters as signatures."""
    keep = (lambda t: t.detach().to(torch.bfloat16).clone()) if dtype == torch.bfloat16 else signature
    from oracle.recipe_ref import FlowMatchSchedule, clip_and_adamw_step, draw_recipe_randoms, optimize_ref
    from oracle.sana_ref import SanaConfig, SanaTransformerRef, init_like_pretrained
    cfg = SanaConfig.tiny()
    torch.manual_seed(0)
    model = SanaTransformerRef(cfg)
    init_like_pretrained(model, seed=3)
    model = model.to(dtype)
    latents, embs = inputs(cfg)
    sched = FlowMatchSchedule()
    out = {}
    # the draws of the reference's fresh, unseeded generator (common/trainer.py:325): pinned separately
    noise, idx, ts, sig = draw_recipe_randoms(latents.shape, len(LENS), sched, torch.Generator(), torch.bfloat16)
    out["draw.noise"], out["draw.indices"], out["draw.timesteps"], out["draw.sigmas"] = noise.float(), idx, ts, sig.float()
    taps = {}
    loss, pred, target = optimize_ref(model, sched, latents, embs, torch.Generator(), PAD_TO, dtype, taps=taps)
    act = (lambda t: t.detach().to(torch.bfloat16)) if dtype == torch.bfloat16 else (lambda t: t.detach().float())
    for k in TAPS:
        out[f"tap.{k}"] = act(taps[k])
    for i in range(cfg.num_layers):
        for k in BLOCK_TAPS:
            out[f"tap.block{i}.{k}"] = act(taps[f"block{i}"][k])
    out["pred"], out["target"], out["loss"] = act(pred), act(target), loss.detach().float().reshape(1)
    loss.backward()
    for name, p in model.named_parameters():
        out[f"grad.{name}"] = keep(p.grad)
    opt = torch.optim.AdamW(model.parameters(), lr=LR, weight_decay=WD)
    total = clip_and_adamw_step(list(model.parameters()), opt, 1.0)
    out["grad_norm"] = total.detach().float().reshape(1)
    for name, p in model.named_parameters():
        # (updated parameters: whole for tensors of <= 8192 elements, signatures above that -- the file stays under 1 MB, and
        #  the update is elementwise in the pinned gradient: a changed AdamW / clip step moves every tensor alike)
        out[f"param_after.{name}"] = keep(p) if p.numel() <= 8192 else signature(p)
    return out, latents, embs


def digest(tensors):
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(tensors[k].contiguous().flatten().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    from safetensors.torch import save_file
    torch.set_num_threads(1)
    tensors = {}
    for tag, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        out, latents, embs = run(dtype)
        for k, v in out.items():
            tensors[f"{tag}.{k}"] = v.contiguous()
    tensors["in.latents"] = latents.contiguous()
    for i, e in enumerate(embs):
        tensors[f"in.emb{i}"] = e.contiguous()
    meta = {"torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(),
            "config": "oracle.sana_ref.SanaConfig.tiny()", "h": str(H), "w": str(W), "lens": json.dumps(LENS),
            "pad_to": str(PAD_TO), "seed": str(SEED), "lr": str(LR), "weight_decay": str(WD),
            "written_by": "tests/golden/make_sana_tiny_golden.py (the oracle itself: a drift pin, not a reference pin)"}
    save_file(tensors, OUT, metadata=meta)
    print(f"{OUT}: {len(tensors)} tensors, {os.path.getsize(OUT)} bytes, digest {digest(tensors)}")
    print(f"loss bf16 {tensors['bf16.loss'].item():.6f} fp32 {tensors['fp32.loss'].item():.6f}; "
          f"grad norm bf16 {tensors['bf16.grad_norm'].item():.6f} fp32 {tensors['fp32.grad_norm'].item():.6f}")


if __name__ == "__main__":
    main()
