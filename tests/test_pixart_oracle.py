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


def test_dpm_solver_pp_2m_known_answers():
    """The validation sampler's scheduler (yat_amd.scheduler.DPMSolverPP2M / oracle DPMSolverPP2MRef, [RECALL] of diffusers'
    DPMSolverMultistepScheduler as the PixArt-Sigma pipeline configures it): the 20-step table (999, 949, ..., 50; trailing sigma
    0), analytic properties of the update -- a zero noise prediction keeps x0 = x / alpha fixed, so every step is the DDIM
    rescale x <- (alpha_t / alpha_s) x and the last one (sigma 0) returns x0 itself; a noise prediction that is exact for a
    fixed x0 reproduces x0 at the end for any order -- and the two restatements agree bit for bit in fp32 and bf16."""
    from oracle.pixart_ref import DPMSolverPP2MRef
    from yat_amd.scheduler import DPMSolverPP2M
    a = DPMSolverPP2M()
    a.set_timesteps(20)
    assert a.timesteps.tolist() == [int(round(x)) for x in torch.linspace(0, 999, 21).flip(0)[:-1].tolist()]
    assert a.timesteps[0] == 999 and a.timesteps[-1] == 50 and a.sigmas[-1] == 0 and a.sigmas.shape == (21,)
    acp = torch.cumprod(1 - torch.linspace(1e-4, 0.02, 1000), 0)
    assert abs(float(a.sigmas[0]) - float(((1 - acp[999]) / acp[999]) ** 0.5)) < 1e-3 * float(a.sigmas[0])
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 6, 6, generator=g)
    # (1) eps = 0: DDIM rescale, the final step lands on x0 = x_first / alpha_first
    a.set_timesteps(5)
    alpha = 1 / (a.sigmas ** 2 + 1) ** 0.5
    cur = x.clone()
    for i in range(5):
        nxt = a.step(torch.zeros_like(cur), cur)
        assert torch.allclose(nxt, cur * (alpha[i + 1] / alpha[i]), rtol=2e-5, atol=1e-6), i
        cur = nxt
    assert torch.allclose(cur, x / alpha[0], rtol=1e-4)
    # (2) the exact noise for a fixed clean sample x0: x_t = alpha_t x0 + sigma_t' n  ->  the sampler returns x0
    x0, n = torch.randn(2, 4, 6, 6, generator=g), torch.randn(2, 4, 6, 6, generator=g)
    for cls in (DPMSolverPP2M, DPMSolverPP2MRef):
        s = cls()
        s.set_timesteps(7)
        al = 1 / (s.sigmas ** 2 + 1) ** 0.5
        cur = al[0] * x0 + s.sigmas[0] * al[0] * n
        for i in range(7):
            eps = (cur - al[i] * x0) / (s.sigmas[i] * al[i])
            cur = s.step(eps, cur)
        assert torch.allclose(cur, x0, rtol=1e-3, atol=1e-4), cls.__name__
    # (3) the two restatements, step by step, both dtypes
    for dt in (torch.float32, torch.bfloat16):
        p, q = DPMSolverPP2M(), DPMSolverPP2MRef()
        p.set_timesteps(6)
        q.set_timesteps(6)
        xa = xb = torch.randn(2, 4, 8, 8, generator=g).to(dt)
        for i in range(6):
            eps = torch.randn(2, 4, 8, 8, generator=g).to(dt)
            xa, xb = p.step(eps, xa), q.step(eps, xb)
            assert torch.equal(xa, xb), (dt, i)
