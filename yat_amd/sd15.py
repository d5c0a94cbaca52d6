"""SD1.5 UNet on the host, loaded from a local diffusers-layout directory -- the model behind ``train_sd15.py --config ...``
(BASELINE config 1: the reference's CPU plumbing run; reference train_sd15.py:27 ``UNet2DConditionModel.from_pretrained(
params.pretrained_model_path)``, :39-44 bf16 + gradient checkpointing, :157-161 the call).

Scope: BASELINE defines this configuration as "plumbing, no GPU" -- 10 steps at batch 1 on the host -- and SURVEY.md section 8
row a24 lists it as CPU smoke only, so there is no HIP UNet: this module is plain ``torch.nn`` arithmetic on the CPU, it refuses
a device that is not the CPU, and nothing on the GPU hot path imports it.  What it gives the entry point is what the
reference gets from diffusers: a module built from ``config.json`` whose ``state_dict`` carries the diffusers key names
[RECALL: UNet2DConditionModel, CrossAttnDownBlock2D / DownBlock2D / UNetMidBlock2DCrossAttn / UpBlock2D / CrossAttnUpBlock2D,
ResnetBlock2D, Transformer2DModel with conv projections, BasicTransformerBlock, GEGLU feed-forward, Downsample2D /
Upsample2D ``.conv``], ``from_pretrained`` / ``save_pretrained`` over ``diffusion_pytorch_model.safetensors``.  Checked against
the oracle's independent restatement (oracle/sd15_ref.py) by tests/test_sd15_cpu.py: same weights, same output.
"""
from __future__ import annotations

import json
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

WEIGHTS = "diffusion_pytorch_model.safetensors"


def sinusoid(t, dim):
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos | sin] of t * 10000^(-i / (dim/2))."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class Resnet(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        skip = self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x
        return skip + h


class Attention(nn.Module):
    """diffusers ``Attention`` with bias-free q / k / v and ``to_out = [Linear, Dropout]``."""

    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        src = x if ctx is None else ctx
        B, N, D = x.shape
        split = lambda t: t.view(B, t.shape[1], self.heads, D // self.heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(x)), split(self.to_k(src)), split(self.to_v(src)))
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, D))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        value, gate = self.proj(x).chunk(2, dim=-1)
        return value * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class TransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.norm1, self.attn1 = nn.LayerNorm(dim), Attention(dim, dim, heads)
        self.norm2, self.attn2 = nn.LayerNorm(dim), Attention(dim, ctx_dim, heads)
        self.norm3, self.ff = nn.LayerNorm(dim), FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class SpatialTransformer(nn.Module):
    """``Transformer2DModel`` as SD1.5 uses it: GroupNorm, 1x1 conv in, one BasicTransformerBlock, 1x1 conv out, residual."""

    def __init__(self, ch, ctx_dim, heads, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([TransformerBlock(ch, ctx_dim, heads)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        tokens = self.proj_in(self.norm(x)).flatten(2).transpose(1, 2)           # [B, H*W, C]
        for blk in self.transformer_blocks:
            tokens = blk(tokens, ctx)
        return self.proj_out(tokens.transpose(1, 2).reshape(B, C, H, W)) + x


class Resample(nn.Module):
    """``Downsample2D`` (stride-2 conv) / ``Upsample2D`` (nearest 2x, then conv): both keep their conv under ``.conv``."""

    def __init__(self, ch, down):
        super().__init__()
        self.down = down
        self.conv = nn.Conv2d(ch, ch, 3, stride=2 if down else 1, padding=1)

    def forward(self, x):
        return self.conv(x if self.down else F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Stage(nn.Module):
    """One down / up block: resnets (+ cross-attention transformers) (+ a resampler)."""

    def __init__(self, in_chs, cout, temb_dim, groups, ctx_dim, heads, cross, resample):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(ci, cout, temb_dim, groups) for ci in in_chs])
        if cross:
            self.attentions = nn.ModuleList([SpatialTransformer(cout, ctx_dim, heads, groups) for _ in in_chs])
        if resample == "down":
            self.downsamplers = nn.ModuleList([Resample(cout, True)])
        elif resample == "up":
            self.upsamplers = nn.ModuleList([Resample(cout, False)])

    def layers(self):
        atts = getattr(self, "attentions", [None] * len(self.resnets))
        return zip(self.resnets, atts)


class MidBlock(nn.Module):
    def __init__(self, ch, temb_dim, groups, ctx_dim, heads):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(ch, ch, temb_dim, groups), Resnet(ch, ch, temb_dim, groups)])
        self.attentions = nn.ModuleList([SpatialTransformer(ch, ctx_dim, heads, groups)])

    def forward(self, x, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx), temb)


DEFAULTS = dict(in_channels=4, out_channels=4, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
                cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32,
                down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
                up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"])


class UNet2DConditionCPU(nn.Module):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(DEFAULTS)
        cfg.update({k: v for k, v in config.items() if k in DEFAULTS})
        self.config = cfg
        chans, L, groups = list(cfg["block_out_channels"]), int(cfg["layers_per_block"]), int(cfg["norm_num_groups"])
        ctx = int(cfg["cross_attention_dim"])
        # SD1.5's config files carry the number of heads under "attention_head_dim" (one int, or one per block) [RECALL]
        hd = cfg["attention_head_dim"]
        heads = list(hd) if isinstance(hd, (list, tuple)) else [int(hd)] * len(chans)
        if len(cfg["down_block_types"]) != len(chans) or len(cfg["up_block_types"]) != len(chans):
            raise ValueError("config.json: block type lists and block_out_channels differ in length")
        for name in list(cfg["down_block_types"]) + list(cfg["up_block_types"]):
            if name not in ("CrossAttnDownBlock2D", "DownBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"):
                raise NotImplementedError(f"config.json: block type {name!r} is not part of the SD1.5 layout")
        c0 = chans[0]
        temb = 4 * c0
        self.conv_in = nn.Conv2d(int(cfg["in_channels"]), c0, 3, padding=1)
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(c0, temb), linear_2=nn.Linear(temb, temb)))
        self.down_blocks = nn.ModuleList()
        prev = c0
        for i, co in enumerate(chans):
            self.down_blocks.append(Stage([prev] + [co] * (L - 1), co, temb, groups, ctx, heads[i],
                                          cfg["down_block_types"][i].startswith("CrossAttn"),
                                          "down" if i < len(chans) - 1 else None))
            prev = co
        self.mid_block = MidBlock(prev, temb, groups, ctx, heads[-1])
        self.up_blocks = nn.ModuleList()
        rev, rheads = chans[::-1], heads[::-1]
        for i, co in enumerate(rev):
            below = rev[min(i + 1, len(rev) - 1)]                 # channels of the skip the LAST resnet of this block takes
            ins = [(prev if j == 0 else co) + (below if j == L else co) for j in range(L + 1)]
            self.up_blocks.append(Stage(ins, co, temb, groups, ctx, rheads[i], cfg["up_block_types"][i].startswith("CrossAttn"),
                                        "up" if i < len(rev) - 1 else None))
            prev = co
        self.conv_norm_out = nn.GroupNorm(groups, c0, eps=1e-5)
        self.conv_out = nn.Conv2d(c0, int(cfg["out_channels"]), 3, padding=1)

    # ---- the surface the trainer uses (common/trainer.py:243-253, train_sd15.py:39-44)
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def enable_gradient_checkpointing(self):
        """train_sd15.py:44.  Accepted and ignored: at batch 1 on the host the activations of a UNet fit easily, and
        recomputation changes no result."""

    def forward(self, sample, timestep, encoder_hidden_states):
        if sample.device.type != "cpu":
            raise NotImplementedError("the SD1.5 UNet is BASELINE's CPU plumbing model: there is no HIP path for it")
        c0 = self.config["block_out_channels"][0]
        temb = sinusoid(timestep.reshape(-1).expand(sample.shape[0]), c0).to(sample.dtype)
        temb = self.time_embedding["linear_2"](F.silu(self.time_embedding["linear_1"](temb)))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            for res, att in blk.layers():
                x = res(x, temb)
                if att is not None:
                    x = att(x, encoder_hidden_states)
                skips.append(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
                skips.append(x)
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            for res, att in blk.layers():
                x = res(torch.cat([x, skips.pop()], dim=1), temb)
                if att is not None:
                    x = att(x, encoder_hidden_states)
            if hasattr(blk, "upsamplers"):
                x = blk.upsamplers[0](x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))

    # ---- diffusers directory layout
    @classmethod
    def from_pretrained(cls, path):
        """``UNet2DConditionModel.from_pretrained(<local dir>)``: ``config.json`` + ``diffusion_pytorch_model.safetensors``;
        every key must match (a diffusers checkpoint of another architecture fails loudly)."""
        from safetensors.torch import load_file
        cfg_path, w_path = os.path.join(path, "config.json"), os.path.join(path, WEIGHTS)
        if not os.path.isfile(cfg_path) or not os.path.isfile(w_path):
            raise FileNotFoundError(f"{path}: expected config.json and {WEIGHTS} (a local diffusers-layout UNet directory; "
                                    f"hub names cannot be fetched offline)")
        with open(cfg_path) as f:
            cfg = json.load(f)
        if cfg.get("use_linear_projection") or cfg.get("class_embed_type") or cfg.get("addition_embed_type"):
            raise NotImplementedError("config.json describes a UNet variant outside the SD1.5 layout")
        model = cls(**cfg)
        model.load_state_dict(load_file(w_path), strict=True)
        return model

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(dict(self.config, _class_name="UNet2DConditionModel"), f, indent=2)
        save_file({k: v.detach().contiguous() for k, v in self.state_dict().items()}, os.path.join(path, WEIGHTS))
