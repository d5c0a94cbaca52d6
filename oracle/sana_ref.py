"""Torch-CPU restatement of the SANA transformer forward (oracle, test-only).

Follows, op for op and dtype for dtype, the reference's vendored copy of the
model:

* top-level forward  : /root/reference/utils/patched_sana_transformer.py:229-349
  (ctor / module inventory :88-167)
* transformer block  : /root/reference/utils/patch_sana_attention_layers.py:72-115
  (ctor :19-70)

Sub-modules that the reference imports from ``diffusers`` (not vendored, not in
this container) are restated from their published behaviour and marked
[RECALL]; each sits in its own small function so a mismatch is one edit.

The module tree reproduces the diffusers attribute names so ``state_dict()``
keys are the checkpoint keys.  Autograd supplies the
backward oracle.  Run it in ``torch.bfloat16`` for the reference's dtype flow
(train_sana.py:21-22,39 loads bf16) or ``torch.float32`` for the ground truth
both bf16 paths are measured against.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class SanaConfig:
    # defaults = utils/patched_sana_transformer.py:88-112  (SANA-1.6B)
    in_channels: int = 32
    out_channels: int = 32
    num_attention_heads: int = 70
    attention_head_dim: int = 32
    num_layers: int = 20
    num_cross_attention_heads: int = 20
    cross_attention_head_dim: int = 112
    cross_attention_dim: int = 2240
    caption_channels: int = 2304
    mlp_ratio: float = 2.5
    sample_size: int = 32
    patch_size: int = 1
    norm_eps: float = 1e-6
    modified_blocks: list = field(default_factory=list)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @property
    def ffn_hidden(self) -> int:
        return int(self.inner_dim * self.mlp_ratio)

    @staticmethod
    def tiny(**kw) -> "SanaConfig":
        """Small config with the same structure (D=64, 2 self heads x 32,
        2 cross heads x 32, 2 layers, caption dim 96, FFN hidden 160)."""
        base = dict(in_channels=8, out_channels=8, num_attention_heads=2, attention_head_dim=32,
                    num_layers=2, num_cross_attention_heads=2, cross_attention_head_dim=32,
                    cross_attention_dim=64, caption_channels=96, mlp_ratio=2.5, sample_size=4)
        base.update(kw)
        return SanaConfig(**base)


# --------------------------------------------------------------------------- #
# [RECALL] diffusers leaf modules
# --------------------------------------------------------------------------- #
def timestep_sinusoid(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """[RECALL] diffusers get_timestep_embedding(t, 256, flip_sin_to_cos=True,
    downscale_freq_shift=0, scale=1, max_period=10000) -> fp32 [B, 256], cos half first."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    arg = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class _TimestepEmbedder(nn.Module):          # [RECALL] TimestepEmbedding
    def __init__(self, dim):
        super().__init__()
        self.linear_1 = nn.Linear(256, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _CombinedTimestep(nn.Module):          # [RECALL] PixArtAlphaCombinedTimestepSizeEmbeddings
    def __init__(self, dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(dim)

    def forward(self, timestep, hidden_dtype):
        return self.timestep_embedder(timestep_sinusoid(timestep).to(hidden_dtype))


class AdaLayerNormSingle(nn.Module):         # [RECALL]; used at patched_sana_transformer.py:133,291-293
    def __init__(self, dim):
        super().__init__()
        self.emb = _CombinedTimestep(dim)
        self.linear = nn.Linear(dim, 6 * dim)

    def forward(self, timestep, hidden_dtype):
        embedded = self.emb(timestep, hidden_dtype)
        return self.linear(F.silu(embedded)), embedded


class TextProjection(nn.Module):             # [RECALL] PixArtAlphaTextProjection(act 'gelu_tanh')
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.gelu(self.linear_1(x), approximate="tanh"))


class RMSNorm(nn.Module):                    # [RECALL] diffusers RMSNorm(eps=1e-5, affine)
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.eps)            # fp32 result
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return x * self.weight


class _PatchEmbed(nn.Module):                # [RECALL] PatchEmbed(pos_embed_type=None): conv + flatten/transpose
    def __init__(self, cin, dim, p):
        super().__init__()
        self.proj = nn.Conv2d(cin, dim, kernel_size=p, stride=p, bias=True)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class _SelfAttn(nn.Module):
    """attn1: to_q/k/v (no bias, patched_sana_transformer.py:149 attention_bias=False), to_out.0 (bias)."""

    def __init__(self, dim, heads, head_dim, bias=False):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        self.to_q = nn.Linear(dim, dim, bias=bias)
        self.to_k = nn.Linear(dim, dim, bias=bias)
        self.to_v = nn.Linear(dim, dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim, bias=True)])

    def linear_attention(self, x):
        """[RECALL] SanaLinearAttnProcessor2_0 (imported patch_sana_attention_layers.py:7)."""
        dt = x.dtype
        q = self.to_q(x).transpose(1, 2).unflatten(1, (self.heads, -1))                  # [B,H,C,N]
        k = self.to_k(x).transpose(1, 2).unflatten(1, (self.heads, -1)).transpose(2, 3)  # [B,H,N,C]
        v = self.to_v(x).transpose(1, 2).unflatten(1, (self.heads, -1))                  # [B,H,C,N]
        q, k = F.relu(q), F.relu(k)
        q, k, v = q.float(), k.float(), v.float()
        v = F.pad(v, (0, 0, 0, 1), mode="constant", value=1.0)                           # [B,H,C+1,N]
        scores = torch.matmul(v, k)                                                      # [B,H,C+1,C]
        o = torch.matmul(scores, q)                                                      # [B,H,C+1,N]
        o = o[:, :, :-1] / (o[:, :, -1:] + 1e-15)
        o = o.flatten(1, 2).transpose(1, 2).to(dt)                                       # [B,N,D]
        return self.to_out[0](o)

    def softmax_attention(self, x):
        """AttnProcessor2_0 variant for blocks in ``modified_blocks`` (patch_sana_attention_layers.py:125-131)."""
        B, N, _ = x.shape
        q = self.to_q(x).view(B, N, self.heads, self.head_dim).transpose(1, 2)
        k = self.to_k(x).view(B, N, self.heads, self.head_dim).transpose(1, 2)
        v = self.to_v(x).view(B, N, self.heads, self.head_dim).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, -1))


class _CrossAttn(nn.Module):
    """attn2: all projections biased (ctor arguments as in patch_sana_attention_layers.py:54-65).  The processor this oracle
    restates is ``AttnProcessor2_0`` = ``F.scaled_dot_product_attention`` [RECALL]: what the STOCK diffusers
    ``SanaTransformerBlock`` installs on attn2 -- the model ``train_sana.py:21,62`` loads and trains -- and what
    ``patch_sana_attention_layers`` sets on the blocks it modifies (:128-129).  The vendored block class's own ctor (:48,64)
    passes the classic ``AttnProcessor()`` (bf16 ``baddbmm`` + bf16 softmax instead of the fused kernel): the same function
    with other bf16 rounding points; that class is only instantiated when the patched transformer builds fresh blocks, which
    the SANA entry point never does.  Target of this oracle: the SDPA processor."""

    def __init__(self, dim, cross_dim, heads, head_dim):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        inner = heads * head_dim
        self.to_q = nn.Linear(dim, inner, bias=True)
        self.to_k = nn.Linear(cross_dim, inner, bias=True)
        self.to_v = nn.Linear(cross_dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True)])

    def forward(self, x, enc, bias):
        B, N, _ = x.shape
        T = enc.shape[1]
        q = self.to_q(x).view(B, N, self.heads, self.head_dim).transpose(1, 2)
        k = self.to_k(enc).view(B, T, self.heads, self.head_dim).transpose(1, 2)
        v = self.to_v(enc).view(B, T, self.heads, self.head_dim).transpose(1, 2)
        mask = None if bias is None else bias[:, None, :, :].expand(B, self.heads, 1, T)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, -1))


class GLUMBConv(nn.Module):                  # [RECALL] diffusers GLUMBConv(norm_type=None, residual_connection=False)
    def __init__(self, dim, hidden):
        super().__init__()
        self.conv_inverted = nn.Conv2d(dim, hidden * 2, 1, 1, 0)
        self.conv_depth = nn.Conv2d(hidden * 2, hidden * 2, 3, 1, 1, groups=hidden * 2)
        self.conv_point = nn.Conv2d(hidden, dim, 1, 1, 0, bias=False)

    def forward(self, x):
        x = F.silu(self.conv_inverted(x))
        x = self.conv_depth(x)
        x, gate = torch.chunk(x, 2, dim=1)
        return self.conv_point(x * F.silu(gate))


class SanaBlock(nn.Module):
    """patch_sana_attention_layers.py:19-115."""

    def __init__(self, cfg: SanaConfig, softmax_self_attn: bool):
        super().__init__()
        D = cfg.inner_dim
        self.eps = cfg.norm_eps
        self.softmax_self_attn = softmax_self_attn
        self.attn1 = _SelfAttn(D, cfg.num_attention_heads, cfg.attention_head_dim)
        self.attn2 = _CrossAttn(D, cfg.cross_attention_dim, cfg.num_cross_attention_heads,
                                cfg.cross_attention_head_dim)
        self.ff = GLUMBConv(D, cfg.ffn_hidden)
        self.scale_shift_table = nn.Parameter(torch.randn(6, D) / D ** 0.5)

    def forward(self, x, enc, enc_bias, timestep, height, width, taps=None):
        B = x.shape[0]
        D = x.shape[-1]
        # :85-87
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + timestep.reshape(B, 6, -1)).chunk(6, dim=1)
        # :90-92
        h = F.layer_norm(x, (D,), None, None, self.eps)
        h = h * (1 + scale_msa) + shift_msa
        h = h.to(x.dtype)
        # :94-95
        a = self.attn1.softmax_attention(h) if self.softmax_self_attn else self.attn1.linear_attention(h)
        x = x + gate_msa * a
        if taps is not None:
            taps["h1"], taps["attn1"], taps["x_attn1"] = h, a, x
        # :98-104 (no pre-norm on the cross-attention input)
        a2 = self.attn2(x, enc, enc_bias)
        x = a2 + x
        # :107-108
        h = F.layer_norm(x, (D,), None, None, self.eps)
        h = h * (1 + scale_mlp) + shift_mlp
        # :110-113
        h = h.unflatten(1, (height, width)).permute(0, 3, 1, 2)
        f = self.ff(h)
        f = f.flatten(2, 3).permute(0, 2, 1)
        x = x + gate_mlp * f
        if taps is not None:
            taps["x_attn2"], taps["ff"], taps["x_out"] = a2, f, x
        return x


class SanaTransformerRef(nn.Module):
    """patched_sana_transformer.py:88-167 (ctor) and :229-349 (forward)."""

    def __init__(self, cfg: SanaConfig):
        super().__init__()
        self.cfg = cfg
        D = cfg.inner_dim
        self.patch_embed = _PatchEmbed(cfg.in_channels, D, cfg.patch_size)
        self.time_embed = AdaLayerNormSingle(D)
        self.caption_projection = TextProjection(cfg.caption_channels, D)
        self.caption_norm = RMSNorm(D, eps=1e-5)
        self.transformer_blocks = nn.ModuleList(
            [SanaBlock(cfg, i in cfg.modified_blocks) for i in range(cfg.num_layers)])
        self.scale_shift_table = nn.Parameter(torch.randn(2, D) / D ** 0.5)
        self.proj_out = nn.Linear(D, cfg.patch_size * cfg.patch_size * cfg.out_channels)

    def forward(self, hidden_states, encoder_hidden_states, timestep, encoder_attention_mask=None, taps=None):
        cfg = self.cfg
        dt = hidden_states.dtype
        bias = None
        if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:
            bias = (1 - encoder_attention_mask.to(dt)) * -10000.0          # :275-277
            bias = bias.unsqueeze(1)
        B, _, H, W = hidden_states.shape
        p = cfg.patch_size
        ph, pw = H // p, W // p
        x = self.patch_embed(hidden_states)                                 # :284
        tmod, embedded = self.time_embed(timestep, dt)                      # :291-293
        enc = self.caption_projection(encoder_hidden_states)                # :295
        enc = enc.view(B, -1, x.shape[-1])                                  # :296
        enc = self.caption_norm(enc)                                        # :298
        if taps is not None:
            taps["x0"], taps["tmod"], taps["embedded"], taps["enc"] = x, tmod, embedded, enc
        for i, blk in enumerate(self.transformer_blocks):                   # :314-328
            bt = {} if taps is not None else None
            x = blk(x, enc, bias, tmod, ph, pw, taps=bt)
            if taps is not None:
                taps[f"block{i}"] = bt
        # :331 SanaModulatedNorm [RECALL]
        D = x.shape[-1]
        x = F.layer_norm(x, (D,), None, None, 1e-6)
        shift, scale = (self.scale_shift_table[None] + embedded[:, None]).chunk(2, dim=1)
        x = x * (1 + scale) + shift
        x = self.proj_out(x)                                                # :333
        x = x.reshape(B, ph, pw, p, p, -1).permute(0, 5, 1, 3, 2, 4)        # :336-340
        return x.reshape(B, -1, ph * p, pw * p)


def init_like_pretrained(model: SanaTransformerRef, seed: int = 0, std: float = 0.02) -> None:
    """Deterministic random init used for synthetic-weight tests and benches (no checkpoints
    offline).  Weights ~ N(0, std) scaled so activations stay O(1); scale_shift_tables and
    biases get small non-zero values so every gate / bias path is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("caption_norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "scale_shift_table" in name:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5 + (0.5 if p.shape[0] == 6 else 0.0))
            elif p.ndim == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(fan_in))
