"""Torch-CPU restatement of the SD3.5 (MMDiT) transformer and its training recipe (oracle, test-only).

Recipe: /root/reference/train_sd35.py:165-194 (``SD35Trainer.optimize``), followed line by line in ``optimize_ref``.
Model: the reference trains diffusers' ``SD3Transformer2DModel`` (train_sd35.py:4,28-43,58) whose source is NOT vendored in
/root/reference and not in this container, so everything below ``# [RECALL]`` restates published diffusers behaviour --
PARITY UNPINNED for the model math, like oracle/sana_ref.py.  Each diffusers leaf sits in its own small class so a mismatch
is one edit:

* ``PatchEmbed(patch_size=2, pos_embed_max_size)``: Conv2d(k = s = p) -> flatten -> + centre crop of a persistent sin-cos
  table built for a ``pos_embed_max_size`` grid with ``base_size = sample_size // patch_size`` (the buffer is part of the
  state dict, so a bf16 pipeline holds it in bf16);
* ``CombinedTimestepTextProjEmbeddings``: sinusoid(256, cos first) -> TimestepEmbedding, pooled -> Linear-SiLU-Linear, summed;
* ``JointTransformerBlock``: AdaLayerNormZero (6 chunks: shift/scale/gate msa, shift/scale/gate mlp) on both streams --
  ``SD35AdaLayerNormZeroX`` (9 chunks, a second (shift, scale, gate) for ``attn2``) on the image stream of the
  ``dual_attention_layers`` -- joint attention over [image tokens | text tokens] with per-head RMSNorm(eps 1e-6, affine) on
  q and k of both streams, gated residuals, LayerNorm(no affine) + modulate + FeedForward(GELU-tanh, x4) per stream; the last
  block is ``context_pre_only``: its text stream gets AdaLayerNormContinuous (scale first) and ends inside the attention;
* ``AdaLayerNormContinuous`` + ``proj_out`` + unpatchify ``nhwpqc->nchpwq``.

The module tree reproduces the diffusers attribute names, so ``state_dict()`` keys are the checkpoint keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sana_ref import timestep_sinusoid, RMSNorm
from .pixart_ref import sincos_2d as sincos_pos_embed_ref


@dataclass
class SD3Config:
    # defaults: stabilityai/stable-diffusion-3.5-medium transformer/config.json [RECALL]
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: tuple = tuple(range(13))
    qk_norm: str | None = "rms_norm"

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def tiny(**kw) -> "SD3Config":
        """Same structure, small: D = 128 (2 heads x 64), 3 blocks (0 and 1 dual, 2 context_pre_only), 8-channel latents,
        text width 96, pooled 64, table for a 24 x 24 grid."""
        base = dict(sample_size=16, patch_size=2, in_channels=8, out_channels=8, num_layers=3, attention_head_dim=64,
                    num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64,
                    pos_embed_max_size=24, dual_attention_layers=(0, 1))
        base.update(kw)
        return SD3Config(**base)


# --------------------------------------------------------------------------- #
# [RECALL] diffusers leaf modules
# --------------------------------------------------------------------------- #
class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _TextProjSilu(nn.Module):              # PixArtAlphaTextProjection(act_fn="silu")
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, dim, pooled_dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, dim)
        self.text_embedder = _TextProjSilu(pooled_dim, dim)

    def forward(self, timestep, pooled):
        t = self.timestep_embedder(timestep_sinusoid(timestep).to(pooled.dtype))
        return t + self.text_embedder(pooled)


class PatchEmbedMax(nn.Module):
    def __init__(self, cfg: SD3Config):
        super().__init__()
        p, D = cfg.patch_size, cfg.inner_dim
        self.p, self.max = p, cfg.pos_embed_max_size
        self.proj = nn.Conv2d(cfg.in_channels, D, kernel_size=p, stride=p, bias=True)
        base = cfg.sample_size // p
        table = sincos_pos_embed_ref(D, self.max, self.max, base, 1.0)          # fp32 [max*max, D]
        self.register_buffer("pos_embed", table.float().unsqueeze(0), persistent=True)

    def cropped(self, h, w):
        if h > self.max or w > self.max:
            raise ValueError("latent grid larger than pos_embed_max_size")
        top, left = (self.max - h) // 2, (self.max - w) // 2
        t = self.pos_embed.reshape(1, self.max, self.max, -1)[:, top:top + h, left:left + w]
        return t.reshape(1, h * w, -1)

    def forward(self, latent):
        h, w = latent.shape[-2] // self.p, latent.shape[-1] // self.p
        x = self.proj(latent).flatten(2).transpose(1, 2)
        return (x + self.cropped(h, w)).to(x.dtype)


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim, chunks=6):
        super().__init__()
        self.chunks = chunks
        self.linear = nn.Linear(dim, chunks * dim)

    def forward(self, x, emb):
        e = self.linear(F.silu(emb)).chunk(self.chunks, dim=1)
        n = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)
        out = n * (1 + e[1][:, None]) + e[0][:, None]
        if self.chunks == 6:
            return out, e[2], e[3], e[4], e[5]
        out2 = n * (1 + e[7][:, None]) + e[6][:, None]                  # SD35AdaLayerNormZeroX
        return out, e[2], e[3], e[4], e[5], out2, e[8]


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim, cond_dim):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 2 * dim)

    def forward(self, x, cond):
        emb = self.linear(F.silu(cond).to(x.dtype))
        scale, shift = emb.chunk(2, dim=1)                              # scale FIRST
        return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]


class _GELUProj(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):                # FeedForward(dim, dim_out=dim, activation_fn="gelu-approximate")
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class JointAttention(nn.Module):
    """Attention(query_dim, added_kv_proj_dim=dim, context_pre_only, bias=True, qk_norm='rms_norm', eps=1e-6) with
    JointAttnProcessor2_0; ``joint=False``: the plain self-attention of ``attn2``."""

    def __init__(self, dim, heads, head_dim, joint=True, context_pre_only=False, qk_norm=True):
        super().__init__()
        self.heads, self.head_dim, self.joint, self.context_pre_only = heads, head_dim, joint, context_pre_only
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_q = RMSNorm(head_dim, eps=1e-6) if qk_norm else None
        self.norm_k = RMSNorm(head_dim, eps=1e-6) if qk_norm else None
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        if joint:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
            self.norm_added_q = RMSNorm(head_dim, eps=1e-6) if qk_norm else None
            self.norm_added_k = RMSNorm(head_dim, eps=1e-6) if qk_norm else None
            if not context_pre_only:
                self.to_add_out = nn.Linear(dim, dim)

    def _heads(self, x):
        B = x.shape[0]
        return x.view(B, -1, self.heads, self.head_dim).transpose(1, 2)

    def forward(self, hidden, enc=None):
        B, N, _ = hidden.shape
        q, k, v = self._heads(self.to_q(hidden)), self._heads(self.to_k(hidden)), self._heads(self.to_v(hidden))
        if self.norm_q is not None:
            q, k = self.norm_q(q), self.norm_k(k)
        if enc is not None:
            eq, ek, ev = self._heads(self.add_q_proj(enc)), self._heads(self.add_k_proj(enc)), self._heads(self.add_v_proj(enc))
            if self.norm_added_q is not None:
                eq, ek = self.norm_added_q(eq), self.norm_added_k(ek)
            q, k, v = torch.cat([q, eq], dim=2), torch.cat([k, ek], dim=2), torch.cat([v, ev], dim=2)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, self.heads * self.head_dim).to(q.dtype)
        if enc is None:
            return self.to_out[0](o)
        o, eo = o[:, :N], o[:, N:]
        if not self.context_pre_only:
            eo = self.to_add_out(eo)
        return self.to_out[0](o), eo


class JointTransformerBlock(nn.Module):
    def __init__(self, cfg: SD3Config, context_pre_only: bool, dual: bool):
        super().__init__()
        D = cfg.inner_dim
        self.context_pre_only, self.dual = context_pre_only, dual
        self.norm1 = AdaLayerNormZero(D, 9 if dual else 6)
        self.norm1_context = AdaLayerNormContinuous(D, D) if context_pre_only else AdaLayerNormZero(D, 6)
        qk = cfg.qk_norm == "rms_norm"
        self.attn = JointAttention(D, cfg.num_attention_heads, cfg.attention_head_dim, True, context_pre_only, qk)
        if dual:
            self.attn2 = JointAttention(D, cfg.num_attention_heads, cfg.attention_head_dim, False, False, qk)
        self.ff = FeedForward(D)
        if not context_pre_only:
            self.ff_context = FeedForward(D)

    def forward(self, hidden, enc, temb, taps=None):
        D = hidden.shape[-1]
        if self.dual:
            nh, gate_msa, shift_mlp, scale_mlp, gate_mlp, nh2, gate_msa2 = self.norm1(hidden, temb)
        else:
            nh, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden, temb)
        if self.context_pre_only:
            ne = self.norm1_context(enc, temb)
        else:
            ne, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(enc, temb)
        a, ca = self.attn(nh, ne)
        hidden = hidden + gate_msa.unsqueeze(1) * a
        if self.dual:
            hidden = hidden + gate_msa2.unsqueeze(1) * self.attn2(nh2)
        n2 = F.layer_norm(hidden, (D,), None, None, 1e-6)
        n2 = n2 * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        hidden = hidden + gate_mlp.unsqueeze(1) * self.ff(n2)
        if self.context_pre_only:
            enc = None
        else:
            enc = enc + c_gate_msa.unsqueeze(1) * ca
            n2c = F.layer_norm(enc, (D,), None, None, 1e-6)
            n2c = n2c * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
            enc = enc + c_gate_mlp.unsqueeze(1) * self.ff_context(n2c)
        if taps is not None:
            taps["hidden"], taps["enc"] = hidden, enc
        return enc, hidden


class SD3TransformerRef(nn.Module):
    def __init__(self, cfg: SD3Config):
        super().__init__()
        self.cfg = cfg
        D = cfg.inner_dim
        assert cfg.caption_projection_dim == D
        self.pos_embed = PatchEmbedMax(cfg)
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(D, cfg.pooled_projection_dim)
        self.context_embedder = nn.Linear(cfg.joint_attention_dim, D)
        self.transformer_blocks = nn.ModuleList([
            JointTransformerBlock(cfg, context_pre_only=(i == cfg.num_layers - 1), dual=(i in cfg.dual_attention_layers))
            for i in range(cfg.num_layers)])
        self.norm_out = AdaLayerNormContinuous(D, D)
        self.proj_out = nn.Linear(D, cfg.patch_size * cfg.patch_size * cfg.out_channels)

    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, taps=None):
        cfg = self.cfg
        B, _, H, W = hidden_states.shape
        p = cfg.patch_size
        h, w = H // p, W // p
        x = self.pos_embed(hidden_states)
        temb = self.time_text_embed(timestep, pooled_projections)
        enc = self.context_embedder(encoder_hidden_states)
        if taps is not None:
            taps["x0"], taps["temb"], taps["enc0"] = x, temb, enc
        for i, blk in enumerate(self.transformer_blocks):
            bt = {} if taps is not None else None
            enc, x = blk(x, enc, temb, taps=bt)
            if taps is not None:
                taps[f"block{i}"] = bt
        x = self.norm_out(x, temb)
        x = self.proj_out(x)
        x = x.reshape(B, h, w, p, p, cfg.out_channels)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return x.reshape(B, cfg.out_channels, h * p, w * p)


def init_like_pretrained(model: SD3TransformerRef, seed: int = 0) -> None:
    """Deterministic synthetic weights (no checkpoints offline): N(0, 1/fan_in) matrices, small biases, qk-norm weights near
    1, and modulation linears whose output is O(1) (gates/scales of a trained model are, a zero-init adaLN would hide them)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm_q." in name or ".norm_k." in name or ".norm_added_q." in name or ".norm_added_k." in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g) + (0.3 if ("norm1" in name or "norm_out" in name) else 0.0))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))


# --------------------------------------------------------------------------- #
# recipe: train_sd35.py:165-194
# --------------------------------------------------------------------------- #
def optimize_ref(model, sched, latents, prompt_embeds, pooled, generator=None, dtype=torch.bfloat16, noise=None, taps=None):
    """``SD35Trainer.optimize``: noise = randn_tensor(latents.shape) in the latents' dtype (:180; the reference draws it on
    the device from the global RNG -- here from ``generator`` / the global CPU RNG, or handed in as ``noise``); logit-normal
    u (:182) -> indices (:183) -> ``scheduler.timesteps[indices]`` (:184) -> ``scheduler.scale_noise`` (:185) [RECALL:
    ``sigma * noise + (1 - sigma) * sample`` with sigma looked up by timestep, in the sample's dtype] -> model (:188-191) ->
    target = noise - latents (:192) -> ``MSELoss()(pred.to(noise.dtype), target)`` (:193), i.e. evaluated in bf16.
    ``dtype=float32`` evaluates the same draws in fp32 (ground truth)."""
    from .recipe_ref import logit_normal_u
    B = latents.shape[0]
    latents = latents.to(torch.bfloat16)
    if noise is None:
        noise = torch.randn(latents.shape, generator=generator, dtype=torch.bfloat16)
    u = logit_normal_u(B, generator)
    indices = (u * sched.num_train_timesteps).long()
    timesteps = sched.timesteps[indices]
    sigma = sched.sigmas.to(torch.bfloat16)[indices].to(dtype)
    latents, noise = latents.to(dtype), noise.to(dtype)
    while sigma.ndim < latents.ndim:
        sigma = sigma.unsqueeze(-1)
    noisy = sigma * noise + (1.0 - sigma) * latents
    pred = model(noisy, prompt_embeds.to(dtype), pooled.to(dtype), timesteps, taps=taps)
    target = noise - latents
    loss = F.mse_loss(pred.to(noise.dtype), target)
    return loss, pred, target


@torch.no_grad()
def sample_latents_sd3_ref(model, sched, latents, prompt_embeds, pooled, negative_embeds, negative_pooled,
                           num_inference_steps=20, guidance_scale=5.0, dtype=torch.bfloat16):
    """[RECALL] StableDiffusion3Pipeline.__call__ denoising loop as train_sd35.py:129-142 calls it (guidance 5.0, 20 steps,
    output_type='latent'): CFG batch (unconditional | conditional) of token embeddings and pooled projections, one
    transformer call per step with the timestep expanded to the batch, Euler flow-match step in fp32, cast back.
    ``model``: an SD3TransformerRef in ``dtype``; float32 gives the ground truth for the same initial latents.  Test
    infrastructure (oracle/): never imported by the product path."""
    from .recipe_ref import inference_schedule_ref
    timesteps, sigmas = inference_schedule_ref(sched, num_inference_steps)
    x = latents.to(dtype)
    enc = torch.cat([negative_embeds, prompt_embeds]).to(dtype)
    pool = torch.cat([negative_pooled, pooled]).to(dtype)
    for i in range(num_inference_steps):
        x_in = torch.cat([x, x])
        t = timesteps[i].expand(x_in.shape[0])
        v = model(x_in, enc, pool, t)
        v_u, v_c = v.float().chunk(2)
        v = (v_u + guidance_scale * (v_c - v_u)).to(dtype)
        x = (x.float() + (float(sigmas[i + 1]) - float(sigmas[i])) * v.float()).to(dtype)
    return x
