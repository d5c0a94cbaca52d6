"""Torch-CPU restatement of the SD1.5 recipe and a compact UNet2DConditionModel (oracle / test-only).

BASELINE config 1 is the reference's CPU plumbing run (``train_sd15.py``: SD1.5 UNet, batch 1, 10 steps on CPU from a small
local shard).  The recipe follows /root/reference/train_sd15.py:140-165 line by line.  The UNet is diffusers'
``UNet2DConditionModel`` (train_sd15.py:27,42; not vendored, not in this container): restated here [RECALL] with the SD1.5
block structure -- conv_in, sinusoidal time embedding (cos first) -> TimestepEmbedding, CrossAttnDownBlock2D / DownBlock2D with
ResnetBlock2D(GroupNorm-SiLU-conv, time projection, 1x1 shortcut) and Transformer2DModel(GroupNorm, 1x1 proj_in,
BasicTransformerBlock[LN-self attn, LN-cross attn, LN-GEGLU FF], 1x1 proj_out), stride-2 conv downsample, mid block, up
blocks with skip concatenation and nearest-2x + conv upsample, GroupNorm-SiLU-conv_out -- at configurable (tiny) widths.
PARITY UNPINNED, like every model restatement here; config 1 checks plumbing (shards -> sampler -> optimize -> backward ->
clip -> AdamW), not kernels: the reference defines no GPU work for it.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sana_ref import timestep_sinusoid


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(groups, cin, eps=1e-5), nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2, self.conv2 = nn.GroupNorm(groups, cout, eps=1e-5), nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _Attn(nn.Module):
    def __init__(self, dim, ctx, heads):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim, bias=False), nn.Linear(ctx, dim, bias=False), nn.Linear(ctx, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, N, D = x.shape
        q = self.to_q(x).view(B, N, self.heads, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
        return self.to_out[0](o)


class _GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx, heads):
        super().__init__()
        self.norm1, self.attn1 = nn.LayerNorm(dim), _Attn(dim, dim, heads)
        self.norm2, self.attn2 = nn.LayerNorm(dim), _Attn(dim, ctx, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = nn.ModuleList([_GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        h = self.norm3(x)
        for m in self.ff:
            h = m(h)
        return x + h


class Transformer2DModel(nn.Module):
    def __init__(self, ch, ctx, heads, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, ctx, heads)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        return self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2)) + x


class _Down(nn.Module):
    def __init__(self, cin, cout, temb, groups, ctx, heads, layers, cross, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, ctx, heads, groups) for _ in range(layers)]) if cross else None
        self.downsamplers = nn.ModuleList([nn.Conv2d(cout, cout, 3, stride=2, padding=1)]) if downsample else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class _Up(nn.Module):
    def __init__(self, cin, cout, cprev, temb, groups, ctx, heads, layers, cross, upsample):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(layers + 1):
            skip = cin if i == layers else cout
            self.resnets.append(ResnetBlock2D((cprev if i == 0 else cout) + skip, cout, temb, groups))
        self.attentions = nn.ModuleList([Transformer2DModel(cout, ctx, heads, groups) for _ in range(layers + 1)]) if cross else None
        self.upsamplers = nn.ModuleList([nn.Conv2d(cout, cout, 3, padding=1)]) if upsample else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class UNet2DConditionRef(nn.Module):
    """SD1.5 layout: ``block_out_channels=(320, 640, 1280, 1280)``, ``layers_per_block=2``, three cross-attention down blocks +
    one plain, ``cross_attention_dim=768``, 8 heads, 32 groups -- the defaults here are a tiny instance of the same layout."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1, cross_attention_dim=32,
                 heads=4, groups=8, cross=(True, False)):
        super().__init__()
        c0, temb = block_out_channels[0], 4 * block_out_channels[0]
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(c0, temb), linear_2=nn.Linear(temb, temb)))
        self.c0 = c0
        self.down_blocks = nn.ModuleList()
        ch = c0
        for i, co in enumerate(block_out_channels):
            self.down_blocks.append(_Down(ch, co, temb, groups, cross_attention_dim, heads, layers_per_block, cross[i],
                                          i < len(block_out_channels) - 1))
            ch = co
        self.mid_resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups), ResnetBlock2D(ch, ch, temb, groups)])
        self.mid_attention = Transformer2DModel(ch, cross_attention_dim, heads, groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        rcross = list(reversed(cross))
        prev = ch
        for i, co in enumerate(rev):
            cin = rev[min(i + 1, len(rev) - 1)]
            self.up_blocks.append(_Up(cin, co, prev, temb, groups, cross_attention_dim, heads, layers_per_block, rcross[i],
                                      i < len(rev) - 1))
            prev = co
        self.conv_norm_out = nn.GroupNorm(groups, c0, eps=1e-5)
        self.conv_out = nn.Conv2d(c0, out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def enable_gradient_checkpointing(self):
        pass

    def forward(self, sample, timestep, encoder_hidden_states):
        temb = timestep_sinusoid(timestep.expand(sample.shape[0]), self.c0).to(sample.dtype)
        temb = self.time_embedding["linear_2"](F.silu(self.time_embedding["linear_1"](temb)))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips += outs
        x = self.mid_resnets[0](x, temb)
        x = self.mid_attention(x, encoder_hidden_states)
        x = self.mid_resnets[1](x, temb)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


def sd15_optimize_ref(model, sched, latents, embeddings, generator=None):
    """train_sd15.py:140-165: embeddings stacked and squeezed (:145), bf16 latents (:146), bf16 noise (:149; the reference
    draws it on the device from the global RNG), logit-normal index (:151-152) -> ``scheduler.timesteps[index]`` (:153) ->
    ``add_noise`` (:154) -> UNet (:157-161) -> target = noise (:163) -> MSE evaluated in fp32 (:164)."""
    emb = torch.stack(embeddings).squeeze(1).to(torch.bfloat16)
    latents = latents.to(torch.bfloat16)
    noise = torch.randn(latents.shape, generator=generator, dtype=torch.bfloat16)
    t, a, c = sched.sample(latents.shape[0], generator)
    noisy = a.view(-1, 1, 1, 1) * latents + c.view(-1, 1, 1, 1) * noise
    pred = model(noisy, t, emb)
    return F.mse_loss(pred.float(), noise.float())
