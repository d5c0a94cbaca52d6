"""Torch-CPU restatement of the PixArt-Sigma transformer and its training recipe (oracle, test-only; BASELINE config 3).

Follows:
* top-level forward     /root/reference/utils/patch_pixart_sigma_transformer.py:88-198 (the vendored copy of
                        ``PixArtTransformer2DModel.forward``; ctor defaults :30-55).  ``use_repa`` False is the path restated:
                        with REPA the projector output ``repa_proj`` (:161-168) is computed but the trainer's REPA loss is
                        commented out (common/trainer.py:340-341), so it never reaches the loss.
* recipe                /root/reference/train_pixart_sigma.py:151-185

Not vendored by the reference, restated from the published diffusers behaviour and marked [RECALL] -- PARITY UNPINNED for
those (diffusers is an unpinned dependency, requirements.txt:7, absent from this container; the reference has no tests):
``BasicTransformerBlock(norm_type='ada_norm_single')``, ``PatchEmbed`` + ``get_2d_sincos_pos_embed``, ``FeedForward
('gelu-approximate')``, ``AttnProcessor2_0``, ``DDPMScheduler`` tables / ``add_noise``.

Reference quirks kept or decided (documented in DESIGN.md):
* ``optimize(self, latents, embeddings)`` at train_pixart_sigma.py:151 still has the two-argument signature while the
  trainer calls ``optimize(ratio, latents, embeddings, repa_features, generator)`` (common/trainer.py:337): at HEAD the
  PixArt entry point raises TypeError.  The restatement takes the recipe body as written and the trainer's argument list.
* the model is called without ``added_cond_kwargs`` (:179-182), which only works when ``use_additional_conditions`` is
  False (the PixArt-Sigma checkpoints' value); True raises, as in the reference (:97-98).
* the loss is ``MSELoss()(noise_pred.to(noise.dtype), noise)`` -- evaluated in bf16 (:183-184), unlike SANA's fp32.
* ``pos_embed`` is a persistent fp32 buffer that ``pipe.transformer.to(torch.bfloat16)`` (:52) casts to bf16; for aspect
  buckets other than the square base grid PatchEmbed recomputes the table in fp32 on the fly [RECALL].
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .sana_ref import AdaLayerNormSingle, TextProjection
from .recipe_ref import logit_normal_u, pad_embeddings


@dataclass
class PixArtConfig:
    # defaults: utils/patch_pixart_sigma_transformer.py:30-55 (+ PixArt-Sigma-XL-2-1024-MS: caption_channels 4096)
    num_attention_heads: int = 16
    attention_head_dim: int = 72
    in_channels: int = 4
    out_channels: int = 8
    num_layers: int = 28
    cross_attention_dim: int = 1152
    sample_size: int = 128
    patch_size: int = 2
    norm_eps: float = 1e-6
    caption_channels: int = 4096
    interpolation_scale: int | None = None
    use_additional_conditions: bool = False

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @property
    def interp(self) -> int:
        return self.interpolation_scale if self.interpolation_scale is not None else max(self.sample_size // 64, 1)

    @staticmethod
    def tiny(**kw) -> "PixArtConfig":
        base = dict(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=2,
                    cross_attention_dim=48, sample_size=8, patch_size=2, caption_channels=64)
        base.update(kw)
        return PixArtConfig(**base)


# --------------------------------------------------------------------------- [RECALL] diffusers leaf pieces
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """[RECALL] get_1d_sincos_pos_embed_from_grid: float64, sin half first."""
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_h: int, grid_w: int, base_size: int, interpolation_scale: float) -> torch.Tensor:
    """[RECALL] get_2d_sincos_pos_embed(embed_dim, (grid_h, grid_w), base_size=, interpolation_scale=) -> fp32 [h*w, D].
    Token n = i*w + j; the first D/2 channels encode the (scaled) column j, the last D/2 the row i ("w goes first")."""
    gh = np.arange(grid_h, dtype=np.float32) / (grid_h / base_size) / interpolation_scale
    gw = np.arange(grid_w, dtype=np.float32) / (grid_w / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0)               # [2, h, w]: grid[0] = column coordinate, grid[1] = row
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


class PatchEmbed(nn.Module):
    """[RECALL] PatchEmbed(height=width=sample_size, patch_size, in_channels, embed_dim, interpolation_scale):
    Conv2d(k=p, s=p, bias) -> flatten/transpose -> + 2-D sin-cos table (persistent fp32 buffer at the base grid)."""

    def __init__(self, cfg: PixArtConfig):
        super().__init__()
        p = cfg.patch_size
        self.p, self.dim, self.interp = p, cfg.inner_dim, cfg.interp
        self.height = self.width = self.base_size = cfg.sample_size // p
        self.proj = nn.Conv2d(cfg.in_channels, cfg.inner_dim, kernel_size=p, stride=p, bias=True)
        self.register_buffer("pos_embed", sincos_2d(self.dim, self.height, self.width, self.base_size, self.interp)[None])

    def forward(self, latent):
        h, w = latent.shape[-2] // self.p, latent.shape[-1] // self.p
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        if (h, w) != (self.height, self.width):
            pos = sincos_2d(self.dim, h, w, self.base_size, self.interp)[None]      # fp32, recomputed per call
        else:
            pos = self.pos_embed                                                    # follows the module dtype
        return (latent + pos).to(latent.dtype)


class Attention(nn.Module):
    """[RECALL] diffusers Attention + AttnProcessor2_0; all projections biased (attention_bias=True, ctor default :40)."""

    def __init__(self, dim, kv_dim, heads, head_dim):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        inner = heads * head_dim
        self.to_q = nn.Linear(dim, inner, bias=True)
        self.to_k = nn.Linear(kv_dim, inner, bias=True)
        self.to_v = nn.Linear(kv_dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True)])

    def forward(self, x, enc=None, bias=None):
        enc = x if enc is None else enc
        B, N, _ = x.shape
        T = enc.shape[1]
        q = self.to_q(x).view(B, N, self.heads, self.head_dim).transpose(1, 2)
        k = self.to_k(enc).view(B, T, self.heads, self.head_dim).transpose(1, 2)
        v = self.to_v(enc).view(B, T, self.heads, self.head_dim).transpose(1, 2)
        mask = None if bias is None else bias[:, None, :, :].expand(B, self.heads, 1, T)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, -1))


class _GELUProj(nn.Module):                      # [RECALL] diffusers activations.GELU(dim_in, dim_out, approximate='tanh')
    def __init__(self, din, dout):
        super().__init__()
        self.proj = nn.Linear(din, dout)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):                    # [RECALL] FeedForward(dim, activation_fn='gelu-approximate', mult=4)
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class PixArtBlock(nn.Module):
    """[RECALL] BasicTransformerBlock(norm_type='ada_norm_single', norm_elementwise_affine=False), called at
    patch_pixart_sigma_transformer.py:150-158."""

    def __init__(self, cfg: PixArtConfig):
        super().__init__()
        D = cfg.inner_dim
        self.eps = cfg.norm_eps
        self.attn1 = Attention(D, D, cfg.num_attention_heads, cfg.attention_head_dim)
        self.attn2 = Attention(D, cfg.cross_attention_dim, cfg.num_attention_heads, cfg.attention_head_dim)
        self.ff = FeedForward(D)
        self.scale_shift_table = nn.Parameter(torch.randn(6, D) / D ** 0.5)

    def forward(self, x, enc, enc_bias, timestep, taps=None):
        B, _, D = x.shape
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + timestep.reshape(B, 6, -1)).chunk(6, dim=1)
        h = F.layer_norm(x, (D,), None, None, self.eps)
        h = h * (1 + scale_msa) + shift_msa
        a = self.attn1(h)
        x = gate_msa * a + x
        if taps is not None:
            taps["h1"], taps["attn1"], taps["x_attn1"] = h, a, x
        a2 = self.attn2(x, enc, enc_bias)                   # ada_norm_single: no norm in front of the cross-attention
        x = a2 + x
        h = F.layer_norm(x, (D,), None, None, self.eps)
        h = h * (1 + scale_mlp) + shift_mlp
        f = self.ff(h)
        x = gate_mlp * f + x
        if taps is not None:
            taps["x_attn2"], taps["ff"], taps["x_out"] = a2, f, x
        return x


class PixArtTransformerRef(nn.Module):
    def __init__(self, cfg: PixArtConfig):
        super().__init__()
        if cfg.use_additional_conditions:
            raise ValueError("`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`.")
        self.cfg = cfg
        D = cfg.inner_dim
        self.pos_embed = PatchEmbed(cfg)
        self.adaln_single = AdaLayerNormSingle(D)
        self.caption_projection = TextProjection(cfg.caption_channels, D)
        self.transformer_blocks = nn.ModuleList([PixArtBlock(cfg) for _ in range(cfg.num_layers)])
        self.scale_shift_table = nn.Parameter(torch.randn(2, D) / D ** 0.5)
        self.proj_out = nn.Linear(D, cfg.patch_size * cfg.patch_size * cfg.out_channels)

    def forward(self, hidden_states, encoder_hidden_states, timestep, encoder_attention_mask=None, taps=None):
        cfg = self.cfg
        dt = hidden_states.dtype
        bias = None
        if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:      # :119-121
            bias = (1 - encoder_attention_mask.to(dt)) * -10000.0
            bias = bias.unsqueeze(1)
        B = hidden_states.shape[0]
        p = cfg.patch_size
        h, w = hidden_states.shape[-2] // p, hidden_states.shape[-1] // p                # :125-128
        x = self.pos_embed(hidden_states)                                                # :129
        tmod, embedded = self.adaln_single(timestep, dt)                                 # :131-133
        enc = self.caption_projection(encoder_hidden_states).view(B, -1, x.shape[-1])    # :135-137
        if taps is not None:
            taps["x0"], taps["tmod"], taps["embedded"], taps["enc"] = x, tmod, embedded, enc
        for i, blk in enumerate(self.transformer_blocks):                                # :142-158
            bt = {} if taps is not None else None
            x = blk(x, enc, bias, tmod, taps=bt)
            if taps is not None:
                taps[f"block{i}"] = bt
        shift, scale = (self.scale_shift_table[None] + embedded[:, None]).chunk(2, dim=1)   # :172-174
        x = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)                            # :175
        x = x * (1 + scale) + shift                                                      # :177
        x = self.proj_out(x)                                                             # :178
        x = x.reshape(-1, h, w, p, p, cfg.out_channels)                                  # :182-184
        x = torch.einsum("nhwpqc->nchpwq", x)                                            # :185
        return x.reshape(-1, cfg.out_channels, h * p, w * p)                             # :186-188


def init_like_pretrained(model: PixArtTransformerRef, seed: int = 0) -> None:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "scale_shift_table" in name:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5 + (0.5 if p.shape[0] == 6 else 0.0))
            elif p.ndim == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))


# --------------------------------------------------------------------------- recipe
class DDPMSchedule:
    """[RECALL] DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule='linear') tables, as
    ``DDPMScheduler.from_pretrained(pipe_path, subfolder='scheduler')`` builds them (train_pixart_sigma.py:37)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02):
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())      # int64, 999 .. 0

    def add_noise(self, original, noise, timesteps):
        """[RECALL] DDPMScheduler.add_noise: the table is cast to the sample dtype first, every op in that dtype."""
        ac = self.alphas_cumprod.to(dtype=original.dtype)
        a = (ac[timesteps] ** 0.5).flatten()
        b = ((1 - ac[timesteps]) ** 0.5).flatten()
        while a.ndim < original.ndim:
            a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return a * original + b * noise


def pixart_optimize_ref(model, sched: DDPMSchedule, latents, embeddings, noise=None, generator=None, pad_to=300,
                        return_pred=False):
    """train_pixart_sigma.py:151-185.  ``noise`` None draws it from ``generator`` on the CPU (the reference draws on the
    device from the global RNG, :170 -- a stream no other device reproduces, so tests pass the tensor in); the timestep
    draw follows :172-174 exactly (logit-normal u -> index -> scheduler.timesteps[index])."""
    enc, mask = pad_embeddings(embeddings, pad_to)                                        # :158-168
    dt = latents.dtype
    if noise is None:
        noise = torch.randn(latents.shape, generator=generator, dtype=dt)                # :170
    u = logit_normal_u(latents.shape[0], generator)                                      # :172
    indices = (u * sched.num_train_timesteps).long()                                     # :173
    timesteps = sched.timesteps[indices]                                                 # :174
    noisy = sched.add_noise(latents, noise, timesteps)                                   # :176
    out = model(noisy.to(dt), enc.to(dt), timesteps, mask.to(dt))                        # :178-182
    pred = out.chunk(2, 1)[0]
    target = noise[: pred.shape[0], : pred.shape[1], : pred.shape[2], : pred.shape[3]]   # :183
    loss = nn.MSELoss()(pred.to(noise.dtype), target)                                    # :184
    return (loss, out, noise, timesteps) if return_pred else loss


# --------------------------------------------------------------------------- validation sampler (test-only oracle)
class DPMSolverPP2MRef:
    """[RECALL] diffusers DPMSolverMultistepScheduler as the PixArt-Sigma pipelines configure it (dpmsolver++, order 2,
    midpoint, lower_order_final, linear betas, epsilon prediction, linspace spacing, final sigma zero): tables in numpy float32 /
    float64 exactly as ``set_timesteps`` builds them; the step written out formula by formula.  PARITY UNPINNED (no diffusers
    offline)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02):
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        acp = torch.cumprod(1.0 - betas, dim=0)
        self.train_sigmas = (((1 - acp) / acp) ** 0.5).numpy()
        self.n_train = num_train_timesteps

    def set_timesteps(self, n):
        ts = np.linspace(0, self.n_train - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = np.interp(ts, np.arange(0, len(self.train_sigmas)), self.train_sigmas)
        self.timesteps = torch.from_numpy(ts)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.model_outputs, self.step_index, self.lower_order_nums = [None, None], 0, 0

    @staticmethod
    def _a_s(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step(self, model_output, sample):
        i, n = self.step_index, len(self.timesteps)
        lower_order_final = i == n - 1                                  # final_sigmas_type == "zero"
        alpha_t, sigma_t = self._a_s(self.sigmas[i])
        x0 = (sample - sigma_t * model_output) / alpha_t                # convert_model_output (tensor dtype wins over 0-dim fp32)
        self.model_outputs = [self.model_outputs[1], x0]
        sample = sample.to(torch.float32)
        a_t, s_t = self._a_s(self.sigmas[i + 1])
        a_s0, s_s0 = self._a_s(self.sigmas[i])
        lam_t, lam_s0 = torch.log(a_t) - torch.log(s_t), torch.log(a_s0) - torch.log(s_s0)
        h = lam_t - lam_s0
        if self.lower_order_nums < 1 or lower_order_final:
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            a_s1, s_s1 = self._a_s(self.sigmas[i - 1])
            lam_s1 = torch.log(a_s1) - torch.log(s_s1)
            r0 = (lam_s0 - lam_s1) / h
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            d0, d1 = m0, (1.0 / r0) * (m0 - m1)
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * d0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * d1
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(model_output.dtype)


@torch.no_grad()
def sample_latents_pixart_ref(model, latents, prompt_embeds, prompt_mask, negative_embeds, negative_mask,
                              num_inference_steps=20, guidance_scale=5.0, dtype=torch.bfloat16):
    """The denoising loop of utils/patch_pixart_sigma_pipeline.py:158-208 (vendored by the reference) over the oracle
    transformer in ``dtype`` with the [RECALL] DPM-Solver++ scheduler above; dtype=float32: ground truth for the same start."""
    sched = DPMSolverPP2MRef()
    sched.set_timesteps(num_inference_steps)
    x = latents.to(dtype)
    enc = torch.cat([negative_embeds, prompt_embeds]).to(dtype)
    mask = torch.cat([negative_mask, prompt_mask])
    cin = x.shape[1]
    for i, t in enumerate(sched.timesteps):
        x_in = torch.cat([x] * 2)
        cur = t[None].expand(x_in.shape[0])
        eps = model(x_in, enc, cur, encoder_attention_mask=mask)
        e_u, e_c = eps.chunk(2)
        eps = e_u + guidance_scale * (e_c - e_u)
        if eps.shape[1] // 2 == cin:
            eps = eps.chunk(2, dim=1)[0]
        x = sched.step(eps, x)
    return x
