#!/usr/bin/env python3
"""Pin the oracle's [RECALL] model restatements against real ``diffusers`` -- WHEN diffusers is importable.

TEST INFRASTRUCTURE, and it cannot run in the build container: diffusers is not installed there and there is no network
(SURVEY.md section 8c), which is exactly why the model math of this repo is "parity unpinned".  On any machine that has the
reference's dependencies (``pip install diffusers``; requirements.txt:7 of the reference), this script closes that gap:

    python -m oracle.pin_against_diffusers            # exit 0 = every restatement reproduces diffusers; 2 = diffusers absent

For each model family it builds the diffusers class the reference trains (train_sana.py:4 ``SanaTransformer2DModel``,
train_pixart_sigma.py ``PixArtTransformer2DModel``, train_sd35.py:4 ``SD3Transformer2DModel``) at a tiny configuration with
random weights, loads ITS state dict into the oracle module with ``strict=True`` (which also pins every checkpoint key
name the HIP models claim to be compatible with), evaluates both in fp32 on fixed inputs and compares outputs and input
gradients.  A mismatch names the family and the maximum deviation; each [RECALL] leaf of the oracle sits in its own small
function so the fix is one edit.  Nothing here is imported by the product path or by the test-suite.
"""
from __future__ import annotations

import sys

import torch


def _report(name, a, b, tol=2e-5):
    err = (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-12)
    print(f"[pin] {name}: max rel deviation {err:.3e} ({'OK' if err <= tol else 'MISMATCH'})")
    return err <= tol


def pin_sana(dm):
    from oracle.sana_ref import SanaConfig, SanaTransformerRef
    cfg = SanaConfig.tiny()
    real = dm.SanaTransformer2DModel(
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_attention_heads=cfg.num_attention_heads,
        attention_head_dim=cfg.attention_head_dim, num_layers=cfg.num_layers,
        num_cross_attention_heads=cfg.num_cross_attention_heads, cross_attention_head_dim=cfg.cross_attention_head_dim,
        cross_attention_dim=cfg.cross_attention_dim, caption_channels=cfg.caption_channels, mlp_ratio=cfg.mlp_ratio,
        sample_size=cfg.sample_size, patch_size=cfg.patch_size, norm_elementwise_affine=False, norm_eps=cfg.norm_eps,
        interpolation_scale=None).float().eval()
    with torch.no_grad():                       # diffusers zero-initialises some tables: make every path matter
        for p in real.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    mine = SanaTransformerRef(cfg).float().eval()
    mine.load_state_dict(real.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, cfg.in_channels, 4, 6, generator=g, requires_grad=True)
    enc = torch.randn(2, 9, cfg.caption_channels, generator=g)
    mask = torch.tensor([[1] * 9, [1] * 4 + [0] * 5])
    t = torch.tensor([731.0, 12.5])
    a = real(x, encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask).sample
    ga, = torch.autograd.grad(a.square().mean(), x)
    b = mine(x, enc, t, encoder_attention_mask=mask)
    gb, = torch.autograd.grad(b.square().mean(), x)
    return _report("SANA forward", b, a) & _report("SANA input gradient", gb, ga)


def pin_pixart(dm):
    from oracle.pixart_ref import PixArtConfig, PixArtTransformerRef
    cfg = PixArtConfig.tiny()
    real = dm.PixArtTransformer2DModel(
        num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, in_channels=cfg.in_channels,
        out_channels=cfg.out_channels, num_layers=cfg.num_layers, cross_attention_dim=cfg.cross_attention_dim,
        sample_size=cfg.sample_size, patch_size=cfg.patch_size, caption_channels=cfg.caption_channels,
        norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=cfg.norm_eps, attention_bias=True,
        activation_fn="gelu-approximate", use_additional_conditions=False).float().eval()
    with torch.no_grad():
        for p in real.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    mine = PixArtTransformerRef(cfg).float().eval()
    mine.load_state_dict(real.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    side = cfg.sample_size
    x = torch.randn(2, cfg.in_channels, side, side, generator=g)
    enc = torch.randn(2, 7, cfg.caption_channels, generator=g)
    mask = torch.tensor([[1] * 7, [1] * 3 + [0] * 4])
    t = torch.tensor([999, 17])
    a = real(x, encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask,
             added_cond_kwargs={"resolution": None, "aspect_ratio": None}).sample
    b = mine(x, enc, t, encoder_attention_mask=mask)
    return _report("PixArt-Sigma forward", b, a)


def pin_sd3(dm):
    from oracle.sd3_ref import SD3Config, SD3TransformerRef
    cfg = SD3Config.tiny()
    real = dm.SD3Transformer2DModel(
        sample_size=cfg.sample_size, patch_size=cfg.patch_size, in_channels=cfg.in_channels, num_layers=cfg.num_layers,
        attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
        joint_attention_dim=cfg.joint_attention_dim, caption_projection_dim=cfg.caption_projection_dim,
        pooled_projection_dim=cfg.pooled_projection_dim, out_channels=cfg.out_channels,
        pos_embed_max_size=cfg.pos_embed_max_size, dual_attention_layers=tuple(cfg.dual_attention_layers),
        qk_norm="rms_norm").float().eval()
    with torch.no_grad():
        for p in real.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    mine = SD3TransformerRef(cfg).float().eval()
    mine.load_state_dict(real.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, cfg.in_channels, 12, 8, generator=g)
    enc = torch.randn(2, 10, cfg.joint_attention_dim, generator=g)
    pooled = torch.randn(2, cfg.pooled_projection_dim, generator=g)
    t = torch.tensor([640.0, 33.0])
    a = real(x, encoder_hidden_states=enc, pooled_projections=pooled, timestep=t).sample
    b = mine(x, enc, pooled, t)
    return _report("SD3.5 MMDiT forward", b, a)


def pin_schedulers(d):
    from oracle.recipe_ref import FlowMatchSchedule
    s = d.FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=3.0)
    m = FlowMatchSchedule(shift=3.0)
    ok = _report("flow-match sigmas", m.sigmas, s.sigmas[:1000], tol=0) & _report("flow-match timesteps", m.timesteps, s.timesteps, tol=0)
    from oracle.recipe_ref import EMAModelRef
    from diffusers.training_utils import EMAModel
    p = [torch.nn.Parameter(torch.randn(5, 3))]
    a, b = EMAModel(p, decay=0.999), EMAModelRef(p, decay=0.999)
    for _ in range(15):
        with torch.no_grad():
            p[0].add_(0.1)
        a.step(p)
        b.step(p)
    return ok & _report("EMAModel shadow after 15 steps", b.shadow_params[0], a.shadow_params[0], tol=0)


def main():
    try:
        import diffusers
        import diffusers.models as dm
    except ImportError:
        print("[pin] diffusers is not importable here: the oracle stays UNPINNED (see oracle/__init__.py).")
        return 2
    ok = True
    for fn, mod in ((pin_sana, dm), (pin_pixart, dm), (pin_sd3, dm), (pin_schedulers, diffusers)):
        try:
            ok &= bool(fn(mod))
        except Exception as e:                  # a constructor argument that moved between diffusers versions, a key mismatch
            print(f"[pin] {fn.__name__}: could not run against diffusers {getattr(diffusers, '__version__', '?')}: {e!r}")
            ok = False
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
