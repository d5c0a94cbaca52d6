"""CPU checks of the PixArt-Sigma oracle (oracle/pixart_ref.py) and of the host-side tables of the product path."""
import math

import numpy as np
import torch

from oracle.pixart_ref import (DDPMSchedule, PixArtConfig, PixArtTransformerRef, init_like_pretrained, pixart_optimize_ref,
                               sincos_2d)

BF = torch.bfloat16


def test_ddpm_tables_known_answers():
    s = DDPMSchedule()
    assert s.timesteps.dtype == torch.int64 and s.timesteps[0] == 999 and s.timesteps[999] == 0
    assert abs(s.betas[0].item() - 1e-4) < 1e-9 and abs(s.betas[-1].item() - 0.02) < 1e-8
    assert abs(s.alphas_cumprod[0].item() - 0.9999) < 1e-6
    # closed form of the linear schedule in float64
    b = np.linspace(1e-4, 0.02, 1000)
    acp = np.cumprod(1 - b)
    assert np.allclose(s.alphas_cumprod.numpy(), acp, rtol=2e-5)
    assert abs(s.alphas_cumprod[-1].item() - 4.0358e-5) < 1e-7
    # product-side tables are the same numbers
    from yat_amd.scheduler import DDPMSchedule as Prod
    p = Prod()
    assert torch.equal(p.alphas_cumprod, s.alphas_cumprod) and torch.equal(p.timesteps, s.timesteps)
    x, n = torch.full((2, 1, 1, 1), 2.0, dtype=BF), torch.full((2, 1, 1, 1), -1.0, dtype=BF)
    t = torch.tensor([10, 900])
    want = s.add_noise(x, n, t).flatten()
    mine = (p.sqrt_alpha_prod[t] * 2.0 + p.sqrt_one_minus_alpha_prod[t] * -1.0)
    assert torch.equal(want, mine)


def test_sincos_table_properties():
    D, h, w, base, interp = 32, 4, 6, 4, 2
    t = sincos_2d(D, h, w, base, interp)
    assert t.shape == (h * w, D) and t.dtype == torch.float32
    q = D // 4
    omega = 1.0 / 10000 ** (np.arange(q) / q)
    for (i, j) in [(0, 0), (1, 5), (3, 2)]:
        col, row = j / (w / base) / interp, i / (h / base) / interp
        want = np.concatenate([np.sin(col * omega), np.cos(col * omega), np.sin(row * omega), np.cos(row * omega)])
        assert np.allclose(t[i * w + j].numpy(), want, atol=1e-6)
    from yat_amd.pixart import sincos_pos_embed
    assert torch.equal(sincos_pos_embed(D, h, w, base, interp), t)


def test_oracle_step_runs_and_learned_sigma_half_gets_no_gradient():
    cfg = PixArtConfig.tiny()
    m = PixArtTransformerRef(cfg)
    init_like_pretrained(m, 0)
    m = m.to(BF)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 4, 8, 12, generator=g).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (3, 9)]
    loss, out, noise, ts = pixart_optimize_ref(m, DDPMSchedule(), lat, embs, None, torch.Generator(), 16, True)
    assert out.shape == (2, 8, 8, 12) and loss.dtype == BF and math.isfinite(loss.item())
    assert ts.dtype == torch.int64 and ((0 <= ts) & (ts < 1000)).all()
    loss.backward()
    gw = m.proj_out.weight.grad.view(2, 2, 8, -1)           # rows ordered (p, q, c): c >= 4 is the dropped half
    assert gw[:, :, :4].abs().sum() > 0 and gw[:, :, 4:].abs().sum() == 0
    keys = set(m.state_dict())
    assert {"pos_embed.proj.weight", "pos_embed.pos_embed", "adaln_single.linear.weight", "caption_projection.linear_2.bias",
            "transformer_blocks.1.attn2.to_out.0.bias", "transformer_blocks.0.ff.net.0.proj.weight",
            "transformer_blocks.0.ff.net.2.bias", "scale_shift_table", "proj_out.bias"} <= keys
