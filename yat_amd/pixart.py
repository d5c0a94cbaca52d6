"""PixArt-Sigma transformer on the HIP C-ABI: explicit forward and hand-scheduled backward (BASELINE config 3).

Mirrors ``PixArtTransformer2DModel`` as the reference trains it (/root/reference/train_pixart_sigma.py:178-182; the vendored
forward at /root/reference/utils/patch_pixart_sigma_transformer.py:88-198, ctor defaults :30-55): same call contract
``model(noisy, encoder_hidden_states=, timestep=, encoder_attention_mask=).sample`` -> [B, 2C, H, W], same ``state_dict()``
keys (diffusers layout: ``pos_embed.proj``, ``adaln_single.*``, ``caption_projection.*``, ``transformer_blocks.i.{attn1,
attn2,ff.net.0.proj,ff.net.2,scale_shift_table}``, ``scale_shift_table``, ``proj_out``).  Sub-modules the reference takes
from diffusers are [RECALL] (oracle/pixart_ref.py restates them; parity unpinned).

Same MI355X design as yat_amd/sana.py (flat parameter / gradient buffers, token-major [B*N, C] rows end to end, every
activation kept -- ~1.7 GB per block at B=8, 46 GB for the 28 blocks, no recompute --, straight-line C-ABI launches, weight
gradients and the text branch on a second stream).  What differs from SANA is only the composition:
* PatchEmbed(k=2, s=2) = ``yat_patch_rearrange`` + GEMM(K = 16) + ``yat_add_pos_embed`` (2-D sin-cos table, fp32);
* attn1 is softmax attention, 16 heads x 72: the flash kernels of yat_amd/csrc/sdpa.hip over the fused [3D] projection
  (head dim padded to 128 inside the kernel's LDS image), N = T = 4096 at 1024 px;
* attn2 reads the projected captions directly (no RMSNorm), T = 300 padded keys;
* the FFN is Linear(D, 4D) -> GELU(tanh) -> Linear(4D, D): two GEMMs with fused bias+activation / bias+gate+residual epilogues;
* the head emits 2C channels (learned sigma); unpatchify is ``yat_patch_rearrange`` in the "nhwpqc" order.
PEFT adapters (yat_amd/lora.py, yat_amd/lokr.py) hook in exactly as in yat_amd/sana.py: ``lin`` / ``dgrad`` / ``wgrad``.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, asdict
from types import SimpleNamespace

import torch

from . import ops
from .flat import FlatParamModule, schedule
from .lokr import adapted_linear

BF16 = torch.bfloat16


@dataclass
class PixArtConfig:
    # defaults: utils/patch_pixart_sigma_transformer.py:30-55; caption_channels of the PixArt-Sigma checkpoints (T5-XXL)
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
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    @property
    def interp(self):
        return self.interpolation_scale if self.interpolation_scale is not None else max(self.sample_size // 64, 1)

    def validate(self):
        D, p = self.inner_dim, self.patch_size
        if self.use_additional_conditions:      # the reference passes no added_cond_kwargs (:179-182) -> its forward raises
            raise ValueError("`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`.")
        if self.cross_attention_dim != D:
            raise ValueError("cross_attention_dim must equal the model dim (the captions are projected to it)")
        if D % 8 or self.caption_channels % 8 or (self.in_channels * p * p) % 8 or (self.out_channels * p * p) % 4:
            raise ValueError("channel sizes must be multiples of 8 (16-byte vector accesses)")
        if self.attention_head_dim > 128 or self.attention_head_dim % 8:
            raise ValueError("attention head dim must be <= 128 and a multiple of 8")
        if self.sample_size % p:
            raise ValueError("sample_size must be a multiple of patch_size")


def _param_specs(cfg: PixArtConfig):
    """(diffusers key, shape) in forward-execution order; q|k|v (weights, then biases) back to back so each fuses."""
    D, Cc, p = cfg.inner_dim, cfg.caption_channels, cfg.patch_size
    specs = [
        ("pos_embed.proj.weight", (D, cfg.in_channels, p, p)), ("pos_embed.proj.bias", (D,)),
        ("adaln_single.emb.timestep_embedder.linear_1.weight", (D, 256)),
        ("adaln_single.emb.timestep_embedder.linear_1.bias", (D,)),
        ("adaln_single.emb.timestep_embedder.linear_2.weight", (D, D)),
        ("adaln_single.emb.timestep_embedder.linear_2.bias", (D,)),
        ("adaln_single.linear.weight", (6 * D, D)), ("adaln_single.linear.bias", (6 * D,)),
        ("caption_projection.linear_1.weight", (D, Cc)), ("caption_projection.linear_1.bias", (D,)),
        ("caption_projection.linear_2.weight", (D, D)), ("caption_projection.linear_2.bias", (D,)),
    ]
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        specs += [
            (b + "scale_shift_table", (6, D)),
            (b + "attn1.to_q.weight", (D, D)), (b + "attn1.to_k.weight", (D, D)), (b + "attn1.to_v.weight", (D, D)),
            (b + "attn1.to_q.bias", (D,)), (b + "attn1.to_k.bias", (D,)), (b + "attn1.to_v.bias", (D,)),
            (b + "attn1.to_out.0.weight", (D, D)), (b + "attn1.to_out.0.bias", (D,)),
            (b + "attn2.to_q.weight", (D, D)), (b + "attn2.to_q.bias", (D,)),
            (b + "attn2.to_k.weight", (D, D)), (b + "attn2.to_v.weight", (D, D)),
            (b + "attn2.to_k.bias", (D,)), (b + "attn2.to_v.bias", (D,)),
            (b + "attn2.to_out.0.weight", (D, D)), (b + "attn2.to_out.0.bias", (D,)),
            (b + "ff.net.0.proj.weight", (4 * D, D)), (b + "ff.net.0.proj.bias", (4 * D,)),
            (b + "ff.net.2.weight", (D, 4 * D)), (b + "ff.net.2.bias", (D,)),
        ]
    specs += [("scale_shift_table", (2, D)), ("proj_out.weight", (p * p * cfg.out_channels, D)),
              ("proj_out.bias", (p * p * cfg.out_channels,))]
    return specs


def sincos_pos_embed(embed_dim: int, grid_h: int, grid_w: int, base_size: int, interpolation_scale: float) -> torch.Tensor:
    """[RECALL diffusers get_2d_sincos_pos_embed] fp32 [grid_h*grid_w, embed_dim] for token n = i*grid_w + j: channels
    [0, D/2) encode the column coordinate j / (grid_w / base_size) / interpolation_scale, [D/2, D) the row coordinate; each
    half is [sin | cos] over omega_k = 10000^(-k / (D/4)), evaluated in float64."""
    q = embed_dim // 4
    omega = 1.0 / 10000 ** (torch.arange(q, dtype=torch.float64) / q)
    gh = (torch.arange(grid_h, dtype=torch.float32) / (grid_h / base_size) / interpolation_scale).double()
    gw = (torch.arange(grid_w, dtype=torch.float32) / (grid_w / base_size) / interpolation_scale).double()
    col = gw[None, :].expand(grid_h, grid_w).reshape(-1, 1) * omega[None]
    row = gh[:, None].expand(grid_h, grid_w).reshape(-1, 1) * omega[None]
    return torch.cat([col.sin(), col.cos(), row.sin(), row.cos()], dim=1).float()


class _WholeModel(torch.autograd.Function):
    """One autograd node for the whole transformer: backward = the hand-scheduled HIP backward."""

    @staticmethod
    def forward(ctx, anchor, model, latents, enc, timestep, mask):
        ctx.model = model
        work, model.next_kv_work = getattr(model, "next_kv_work", None), None      # one-shot hint from the recipe
        return model.forward_impl(latents, enc, timestep, mask, kv_work=work).clone()     # (the prediction lives in the arena)

    @staticmethod
    def backward(ctx, dout):
        ctx.model.backward_impl(dout.contiguous())
        return None, None, None, None, None, None


class PixArtTransformer2DModelHIP(FlatParamModule):
    def __init__(self, cfg: PixArtConfig | None = None, device="cuda", **cfg_kw):
        super().__init__()
        cfg = cfg or PixArtConfig(**cfg_kw)
        cfg.validate()
        self.cfg = cfg
        self.config = SimpleNamespace(**asdict(cfg))
        specs = _param_specs(cfg)
        offs, total = self._alloc_flat(specs, device, bucket_first=lambda k: k.startswith("transformer_blocks.") and k.split(".", 2)[2] == "scale_shift_table")
        self.bucket_bounds = self._block_buckets(specs, offs, total, cfg.num_layers)
        # weight gradients / text branch on a 2nd stream; independent forward chains over image ranges: 241.9 -> 237.9 ms per
        # step with two (same box).  (GELU' in the ff.net.2 dgrad epilogue -- yat_gemm_epilogue.dact_z, 92 us less kernel
        # time per block alone -- was measured SLOWER in the step, 267.3 vs 261.2 ms: the separate HBM-bound pass overlaps
        # the second stream's weight gradients for free, a longer epilogue in a one-workgroup-per-CU GEMM does not.  The
        # branch is gone; the epilogue stays a library feature with its own kernel tests.)
        self.side_wgrad, self.fwd_chains = schedule(2)
        self.split_parts = True           # LN statistics / cross dK,dV off the dependent chain
        # Bias gradients that are their own column-sum pass (weight gradients of < 96 tiles are split along K, and the fused
        # row-sum form needs the whole K in one workgroup) leave the weight-gradient stream: a 14 us column sum queued there
        # waits for CUs that the GEMM workgroups of both streams hold for their whole life -- the kernel trace shows it
        # "running" for a median 216 - 270 us, and the next weight gradient queued behind it.  On the forward's second chain
        # stream (idle during the backward) it trickles in beside them instead.
        self.aux_colsum = True
        self.pos_bf16_base = True         # the base-grid table is a module buffer: ``.to(bfloat16)`` rounds it (:52)
        self._pos = {}

    def init_synthetic(self, seed: int = 0):
        """Deterministic random weights of the right scale (no checkpoints offline)."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.P.items():
                if "scale_shift_table" in name:
                    p.copy_(torch.randn(p.shape, generator=g, device=self.dev) / p.shape[-1] ** 0.5)
                elif p.ndim == 1:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.dev))
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device=self.dev) / math.sqrt(p[0].numel()))
        return self

    def pos_table(self, h, w):
        """PatchEmbed's table for an h x w token grid, cached per bucket (device fp32 [h*w, D])."""
        key = (h, w)
        if key not in self._pos:
            base = self.cfg.sample_size // self.cfg.patch_size
            t = sincos_pos_embed(self.cfg.inner_dim, h, w, base, self.cfg.interp)
            if (h, w) == (base, base) and self.pos_bf16_base:
                t = t.to(BF16).float()
            self._pos[key] = t.to(self.dev).contiguous()
        return self._pos[key]

    # ------------------------------------------------------------------ public forward (reference call contract)
    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, encoder_attention_mask=None,
                added_cond_kwargs=None, return_dict=True, **unused):
        if torch.is_grad_enabled():
            out = _WholeModel.apply(self._anchor, self, hidden_states, encoder_hidden_states, timestep,
                                    encoder_attention_mask)
        else:
            out = self.forward_impl(hidden_states, encoder_hidden_states, timestep, encoder_attention_mask).clone()
        return SimpleNamespace(sample=out) if return_dict else (out,)

    # ------------------------------------------------------------------ device path (launch plans, yat_amd/flat.py)
    def _schedule_flags(self):
        return (self.side_wgrad, self.split_parts, self.fwd_chains, self.aux_colsum, self.training)

    def forward_device(self, latents, enc, timestep, key_bias, kv_len, kv_work=None):
        """``forward_impl`` on device-resident inputs in persistent buffers, replayed from a launch plan when this (shapes,
        addresses, schedule) combination has run before (yat_amd/sana.py does the same).  The prediction is an arena buffer:
        consume it before the next call."""
        pev = self.param_events
        self._require_device(latents=(latents, BF16), enc=(enc, BF16), timestep=(timestep, torch.float32),
                             key_bias=(key_bias, torch.float32), kv_len=(kv_len, torch.int32))
        key = (latents.data_ptr(), tuple(latents.shape), enc.data_ptr(), tuple(enc.shape), timestep.data_ptr(),
               key_bias.data_ptr(), kv_len.data_ptr(), None if pev is None else id(pev[0]), self._schedule_flags())
        out = self.planned("fwd", key, lambda: self.forward_impl(latents, enc, timestep, None, key_bias=key_bias,
                                                                 kv_len=kv_len, kv_work=kv_work))
        self.param_events = None          # consumed by the forward (recorded or replayed)
        self._saved.kv_work = kv_work
        return out

    def backward_device(self, dpred):
        S = self._saved
        work = S.kv_work
        if work is not None:
            self.plan_dynamic["n_work"] = int(work.shape[0])
        key = (id(S), dpred.data_ptr(), self.accumulate_grads, id(self.grad_ready), None if work is None else work.data_ptr(),
               self._schedule_flags())
        self.planned("bwd", key, lambda: self.backward_impl(dpred))

    # ------------------------------------------------------------------ forward
    def forward_impl(self, latents, enc, timestep, mask=None, key_bias=None, kv_len=None, kv_work=None):
        cfg, P = self.cfg, self.P
        ad = self.adapters
        if ad is not None:
            ad.materialize(self.training)                     # yat_amd/lora.py / lokr.py: this step's adapter state
        D, H, dh, p = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim, cfg.patch_size
        B, Cin, Hl, Wl = latents.shape
        if Hl % p or Wl % p:
            raise ValueError("latent size must be a multiple of patch_size")
        h, w = Hl // p, Wl // p
        N, M = h * w, B * h * w
        T = enc.shape[1]
        Mt = B * T
        Kp, Co = Cin * p * p, p * p * cfg.out_channels
        dev = self.dev
        f32 = torch.float32
        latents = latents.to(device=dev, dtype=BF16).contiguous()
        enc2d = enc.to(device=dev, dtype=BF16).contiguous().view(Mt, -1)
        t_f32 = timestep.to(device=dev, dtype=f32).contiguous()
        if key_bias is None:
            if mask is None:
                key_bias = torch.zeros(B, T, dtype=f32, device=dev)
                kv_len = torch.full((B,), T, dtype=torch.int32, device=dev)
            else:                                                        # :119-121, evaluated in bf16 as there
                mdev = mask.to(dev)
                key_bias = ((1 - mdev.to(BF16)) * -10000.0).float().contiguous()
                idx = torch.arange(1, T + 1, device=dev, dtype=torch.int32)
                kv_len = (mdev.to(torch.int32) * idx).amax(dim=1).to(torch.int32).contiguous()
        S = SimpleNamespace(B=B, h=h, w=w, N=N, M=M, T=T, Mt=Mt, Hl=Hl, Wl=Wl, key_bias=key_bias, kv_len=kv_len,
                            kv_work=kv_work, enc2d=enc2d, blocks=[])
        buf = self._buf
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None
        pev, self.param_events = self.param_events, None
        # the embedders / text branch below must not inherit the policy word's stream count from the previous call (1 in the
        # very first step, the backward's value afterwards): yat_amd/sana.py forward_impl has the story
        ops.gemm_concurrency(2 if self.side_wgrad else 1)

        def lin(x_, w_, bias_=None, out=None, **ep):
            """Linear of a (possibly adapted) target: the adapter term is folded in through the GEMM's pre_add epilogue."""
            return adapted_linear(ad, x_, w_, bias_, out=out, **ep)

        def params_ready(bucket, stream=main):
            if pev is not None:
                self._ev_wait(stream, pev[bucket])

        # text branch (caption projection + every block's K/V projection): independent of the latent stream until the
        # first cross-attention -> second stream
        def text_branch():
            S.zc1 = buf("cap_z1", (Mt, D))
            S.c1 = lin(enc2d, P["caption_projection.linear_1.weight"], P["caption_projection.linear_1.bias"],
                       out=buf("cap_c1", (Mt, D)), activation="gelu_tanh", aux_out=S.zc1)
            S.encp = lin(S.c1, P["caption_projection.linear_2.weight"], P["caption_projection.linear_2.bias"],
                         out=buf("cap_c2", (Mt, D)))
            S.kv2, S.kv_ready = [], []
            cur = torch.cuda.current_stream()
            for i in range(cfg.num_layers):
                pre = f"transformer_blocks.{i}."
                params_ready(i + 1, cur)
                wkv, _ = self._fused(pre + "attn2.to_k.weight", 2 * D, D)
                bkv, _ = self._fused(pre + "attn2.to_k.bias", 2 * D)
                S.kv2.append(lin(S.encp, wkv, bkv, out=buf(f"b{i}.kv2", (Mt, 2 * D))))
                if side is not None:
                    S.kv_ready.append(self._ev_record(cur))

        params_ready(0)
        if side is not None:
            self._wait_stream(side, main)
            with torch.cuda.stream(side):
                text_branch()
        else:
            text_branch()

        # 1. PatchEmbed: p x p patches as rows -> GEMM -> + position table
        S.x_tok = ops.patch_rearrange(latents, buf("x_tok", (M, Kp)), B, Cin, Hl, Wl, p, True, True)
        x = lin(S.x_tok, P["pos_embed.proj.weight"].view(D, Kp), P["pos_embed.proj.bias"], out=buf("x0", (M, D)))
        ops.add_pos_embed(x, self.pos_table(h, w))
        # 2. timestep embedding (AdaLayerNormSingle without the size/aspect conditions)
        pre = "adaln_single."
        S.tproj = ops.timestep_embed(t_f32, 256, buf("tproj", (B, 256)))
        S.z1 = buf("te_z1", (B, D))
        S.e1 = lin(S.tproj, P[pre + "emb.timestep_embedder.linear_1.weight"], P[pre + "emb.timestep_embedder.linear_1.bias"],
                   out=buf("te_e1", (B, D)), activation="silu", aux_out=S.z1)
        S.embedded = lin(S.e1, P[pre + "emb.timestep_embedder.linear_2.weight"],
                         P[pre + "emb.timestep_embedder.linear_2.bias"], out=buf("te_emb", (B, D)))
        S.se = ops.act_fwd(S.embedded, "silu", buf("te_se", (B, D)))
        S.tmod = lin(S.se, P[pre + "linear.weight"], P[pre + "linear.bias"], out=buf("te_tmod", (B, 6 * D)))
        # 3. blocks
        scale = 1.0 / math.sqrt(dh)
        # activations live in whole-batch buffers (the backward runs on the whole batch); the forward walks them as
        # `fwd_chains` independent chains over disjoint image ranges, each on its own stream (yat_amd/sana.py does the same)
        for i in range(cfg.num_layers):
            A = SimpleNamespace()
            A.mod = buf(f"b{i}.mod", (B, 6, D))
            A.h1, A.mean1, A.rstd1 = buf(f"b{i}.h1", (M, D)), buf(f"b{i}.mean1", (M,), f32), buf(f"b{i}.rstd1", (M,), f32)
            A.qkv, A.attn, A.lse1 = buf(f"b{i}.qkv", (M, 3 * D)), buf(f"b{i}.attn", (M, D)), buf(f"b{i}.lse1", (B, H, N), f32)
            A.lin1, A.x1, A.q2 = buf(f"b{i}.lin1", (M, D)), buf(f"b{i}.x1", (M, D)), buf(f"b{i}.q2", (M, D))
            A.kv2 = S.kv2[i]
            A.o2, A.lse, A.x2 = buf(f"b{i}.o2", (M, D)), buf(f"b{i}.lse", (B, H, N), f32), buf(f"b{i}.x2", (M, D))
            A.h2, A.mean2, A.rstd2 = buf(f"b{i}.h2", (M, D)), buf(f"b{i}.mean2", (M,), f32), buf(f"b{i}.rstd2", (M,), f32)
            A.z, A.f1 = buf(f"b{i}.z", (M, 4 * D)), buf(f"b{i}.f1", (M, 4 * D))          # z: pre-activation, for GELU'
            A.lin3, A.x3 = buf(f"b{i}.lin3", (M, D)), buf(f"b{i}.x3", (M, D))
            A.x_in = x if i == 0 else S.blocks[i - 1].x3
            S.blocks.append(A)
        S.x_last = S.blocks[-1].x3 if cfg.num_layers else x
        S.modf = buf("modf", (B, 2, D))
        S.hf, S.meanf, S.rstdf = buf("hf", (M, D)), buf("meanf", (M,), f32), buf("rstdf", (M,), f32)
        out_tok = buf("out_tok", (M, Co))
        pred = buf("pred", (B, cfg.out_channels, Hl, Wl))     # (arena: the caller consumes it before the next forward)

        def run_chain(b0, b1, stream):
            nb = b1 - b0
            rs, ts, bs = slice(b0 * N, b1 * N), slice(b0 * T, b1 * T), slice(b0, b1)
            for i in range(cfg.num_layers):
                pre = f"transformer_blocks.{i}."
                A = S.blocks[i]
                xin = A.x_in[rs]
                params_ready(i + 1, stream)
                ops.modulation_fwd(P[pre + "scale_shift_table"], S.tmod[bs], D, A.mod[bs])
                mod2d = A.mod.view(B, 6 * D)[bs]
                ops.ln_modulate_fwd(xin, mod2d[:, 0:D], mod2d[:, D:2 * D], 6 * D, N, cfg.norm_eps, A.h1[rs], A.mean1[rs], A.rstd1[rs])
                wqkv, _ = self._fused(pre + "attn1.to_q.weight", 3 * D, D)
                bqkv, _ = self._fused(pre + "attn1.to_q.bias", 3 * D)
                qkv = lin(A.h1[rs], wqkv, bqkv, out=A.qkv[rs])
                ops.sdpa_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], nb, N, N, H, dh, scale, None, None,
                             A.attn[rs], A.lse1[bs])      # (no key bias: the self-attention instantiations)
                lin(A.attn[rs], P[pre + "attn1.to_out.0.weight"], P[pre + "attn1.to_out.0.bias"], out=A.x1[rs],
                    aux_out=A.lin1[rs], gate=mod2d[:, 2 * D:3 * D], ld_gate=6 * D, residual=xin, rows_per_batch=N)
                lin(A.x1[rs], P[pre + "attn2.to_q.weight"], P[pre + "attn2.to_q.bias"], out=A.q2[rs])
                if side is not None:
                    self._ev_wait(stream, S.kv_ready[i])
                kv = A.kv2[ts]
                ops.sdpa_fwd(A.q2[rs], kv[:, :D], kv[:, D:], nb, N, T, H, dh, scale, key_bias[bs], kv_len[bs], A.o2[rs], A.lse[bs])
                lin(A.o2[rs], P[pre + "attn2.to_out.0.weight"], P[pre + "attn2.to_out.0.bias"], out=A.x2[rs], residual=A.x1[rs])
                ops.ln_modulate_fwd(A.x2[rs], mod2d[:, 3 * D:4 * D], mod2d[:, 4 * D:5 * D], 6 * D, N, cfg.norm_eps, A.h2[rs],
                                    A.mean2[rs], A.rstd2[rs])
                lin(A.h2[rs], P[pre + "ff.net.0.proj.weight"], P[pre + "ff.net.0.proj.bias"], out=A.f1[rs],
                    activation="gelu_tanh", aux_out=A.z[rs])
                lin(A.f1[rs], P[pre + "ff.net.2.weight"], P[pre + "ff.net.2.bias"], out=A.x3[rs], aux_out=A.lin3[rs],
                    gate=mod2d[:, 5 * D:6 * D], ld_gate=6 * D, residual=A.x2[rs], rows_per_batch=N)
            # output head: modulated norm + proj_out + unpatchify
            ops.modulation_fwd(P["scale_shift_table"], S.embedded[bs], 0, S.modf[bs])
            modf2d = S.modf.view(B, 2 * D)[bs]
            ops.ln_modulate_fwd(S.x_last[rs], modf2d[:, 0:D], modf2d[:, D:2 * D], 2 * D, N, 1e-6, S.hf[rs], S.meanf[rs], S.rstdf[rs])
            lin(S.hf[rs], P["proj_out.weight"], P["proj_out.bias"], out=out_tok[rs])
            ops.patch_rearrange(out_tok[rs], pred[bs], nb, cfg.out_channels, Hl, Wl, p, False, False)

        nchain = 1 if ad is not None else max(1, min(self.fwd_chains, B))
        ops.gemm_concurrency(nchain)              # the GEMM policy plans each launch for its share of the chip
        if nchain == 1:
            run_chain(0, B, main)
        else:
            bounds = [(B * c) // nchain for c in range(nchain + 1)]
            fork = self._ev_record(main)
            joins = []
            for c in range(1, nchain):
                st = self._chain_stream(c)
                self._ev_wait(st, fork)
                with torch.cuda.stream(st):
                    run_chain(bounds[c], bounds[c + 1], st)
                    joins.append(self._ev_record(st))
            run_chain(bounds[0], bounds[1], main)
            for ev in joins:
                self._ev_wait(main, ev)
        self._saved = S
        return pred

    # ------------------------------------------------------------------ backward
    def backward_impl(self, dpred):
        S = self._saved
        if S is None:
            raise RuntimeError("backward_impl called without a saved forward")
        ops.gemm_concurrency(2 if self.side_wgrad else 1)     # dgrad chain beside the weight gradients' stream
        cfg, P, G = self.cfg, self.P, self.G
        D, H, dh, p = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim, cfg.patch_size
        B, N, M, T, Mt = S.B, S.N, S.M, S.T, S.Mt
        Kp, Co = cfg.in_channels * p * p, p * p * cfg.out_channels
        acc = self.accumulate_grads
        buf = self._buf
        f32, u8 = torch.float32, torch.uint8
        lib = ops._lib()
        ws_col = buf("ws_col", (int(lib.yat_colsum_workspace_bytes(max(M, Mt), 6 * D)),), u8)
        ws_ln = buf("ws_ln", (ops.ln_bwd_workspace_bytes(M, D, N),), u8)
        ws_gate = buf("ws_gate", (int(lib.yat_gate_bwd_workspace_bytes(M, D, N)),), u8)
        scale = 1.0 / math.sqrt(dh)
        ad = self.adapters
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None
        aux = self._chain_stream(1) if (side is not None and self.aux_colsum and ad is None) else None
        aux_used = [False]
        pending_ad = []                   # adapter weight gradients wait for the H product of the dgrad of the same dy

        def dgrad(dy_, w_, out=None, residual=None):
            r_ = ops.linear_dgrad(dy_, w_, out=out, residual=residual)
            if ad is not None:
                hs = ad.dgrad_term(dy_, w_, r_)
                keep = []
                for item in pending_ad:
                    if item[0].data_ptr() == dy_.data_ptr():
                        off_chain(lambda item=item, hs=hs: ad.wgrad(*item, accumulate=acc, hs=hs))
                    else:
                        keep.append(item)
                pending_ad[:] = keep
            return r_

        def off_chain(fn):
            """Weight / bias / table gradients: nothing on the dependent chain reads them -> second stream, right behind
            their producer, filling the CUs the chain's launches leave idle."""
            if side is None:
                fn()
                return
            self._wait_stream(side, main)
            with torch.cuda.stream(side):
                fn()

        def wgrad(dy, x, gw, gbias=None, dgrad_follows=True):
            if ad is not None:            # frozen base: only the adapters' share, launched by the dgrad() of the same dy
                if dgrad_follows:
                    pending_ad.append((dy, x, gw))
                else:
                    off_chain(lambda: ad.wgrad(dy, x, gw, accumulate=acc))
                return

            own_pass = aux is not None and gbias is not None and not ops.wgrad_fuses_bias(dy.shape[1], x.shape[1])

            def run():
                ops.linear_wgrad(dy, x, gw, accumulate=acc, bias_grad=None if own_pass else gbias, colsum_ws=ws_col)
            off_chain(run)
            if own_pass:                      # the separate column-sum pass: third stream (see __init__)
                self._wait_stream(aux, torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    ops.colsum(dy, gbias, ws_col, accumulate=acc)
                aux_used[0] = True

        # ---- output head
        d_out_tok = ops.patch_rearrange(dpred.to(BF16).contiguous(), buf("d_out_tok", (M, Co)), B, cfg.out_channels, S.Hl,
                                        S.Wl, p, False, True)
        wgrad(d_out_tok, S.hf, G["proj_out.weight"], G["proj_out.bias"])
        dhf = dgrad(d_out_tok, P["proj_out.weight"], out=buf("dh", (M, D)))
        dmodf = ops.zero_(buf("dmodf", (B, 2, D), f32))
        dtmod = ops.zero_(buf("dtmod", (B, 6 * D), f32))
        demb = ops.zero_(buf("demb", (B, D), f32))
        dxa, dxb = buf("dx_a", (M, D)), buf("dx_b", (M, D))
        dmodf2d = dmodf.view(B, 2 * D)
        dx = ops.ln_modulate_bwd(S.x_last, S.meanf, S.rstdf, S.modf.view(B, 2 * D)[:, D:2 * D], 2 * D, N, dhf, None, dxa,
                                 dmodf2d[:, 0:D], dmodf2d[:, D:2 * D], 2 * D, ws_ln)
        ops.modulation_bwd(dmodf, G["scale_shift_table"], demb, 0, accumulate_table=acc)
        denc = buf("denc", (Mt, D))
        # ---- blocks, last to first.  The gradient buffers the second stream reads alternate between two sets by block
        # parity: the chain is already writing block i-1's while the second stream still reads block i's.
        set_done = [None, None]
        for i in reversed(range(cfg.num_layers)):
            pre = f"transformer_blocks.{i}."
            A = S.blocks[i]
            par = i & 1
            if set_done[par] is not None:
                self._ev_wait(main, set_done[par])
                set_done[par] = None
            mod2d = A.mod.view(B, 6 * D)
            dmod = ops.zero_(buf(f"dmod.{par}", (B, 6, D), f32))
            dmod2d = dmod.view(B, 6 * D)
            # x3 = x2 + gate_mlp * lin3,  lin3 = f1 W2^T + b2
            dlin3 = buf(f"dlin3.{par}", (M, D))
            ops.gate_bwd(dx, A.lin3, mod2d[:, 5 * D:6 * D], 6 * D, N, dlin3, dmod2d[:, 5 * D:6 * D], 6 * D, ws_gate,
                         dbias=G[pre + "ff.net.2.bias"], accumulate_bias=acc)
            wgrad(dlin3, A.f1, G[pre + "ff.net.2.weight"])
            df1 = dgrad(dlin3, P[pre + "ff.net.2.weight"], out=buf("df1", (M, 4 * D)))
            dz = ops.act_bwd(A.z, df1, "gelu_tanh", buf(f"dz.{par}", (M, 4 * D)))
            wgrad(dz, A.h2, G[pre + "ff.net.0.proj.weight"], G[pre + "ff.net.0.proj.bias"])
            # this step is bound by the dependent chain (serialized kernel time 264 ms vs 247 ms per step), so what only
            # parameters or the text side need leaves it: LayerNorm column statistics and the cross-attention dK/dV go to the
            # second stream (their inputs then live in the parity-alternating buffers)
            split = side is not None and self.split_parts
            dh2 = dgrad(dz, P[pre + "ff.net.0.proj.weight"], out=buf(f"dh2.{par}" if split else "dh", (M, D)))
            other = dxb if dx is dxa else dxa
            ln2 = (A.x2, A.mean2, A.rstd2, mod2d[:, 4 * D:5 * D], 6 * D, N, dh2)
            dx2 = ops.ln_modulate_bwd(*ln2, dx, buf(f"dx2.{par}", (M, D)), dmod2d[:, 3 * D:4 * D], dmod2d[:, 4 * D:5 * D], 6 * D,
                                      ws_ln, parts=1 if split else 3)
            if split:
                off_chain(lambda ln2=ln2, dmod2d=dmod2d: ops.ln_modulate_bwd(
                    *ln2, None, None, dmod2d[:, 3 * D:4 * D], dmod2d[:, 4 * D:5 * D], 6 * D, ws_ln, parts=2))
            # x2 = x1 + to_out(o2)
            wgrad(dx2, A.o2, G[pre + "attn2.to_out.0.weight"], G[pre + "attn2.to_out.0.bias"])
            do2 = dgrad(dx2, P[pre + "attn2.to_out.0.weight"], out=buf(f"do2.{par}" if split else "do", (M, D)))
            dq2 = buf(f"dq2.{par}", (M, D))
            dkv2 = buf(f"dkv2.{par}", (Mt, 2 * D))
            sd = (A.q2, A.kv2[:, :D], A.kv2[:, D:], B, N, T, H, dh, scale, S.key_bias, S.kv_len, A.o2, do2, A.lse,
                  buf(f"delta2.{par}" if split else "delta", (B, H, N), f32), dq2, dkv2[:, :D], dkv2[:, D:])
            ops.sdpa_bwd(*sd, work=S.kv_work, parts=1 if split else 3)
            if split:
                off_chain(lambda sd=sd: ops.sdpa_bwd(*sd, work=S.kv_work, parts=2))
            wgrad(dq2, A.x1, G[pre + "attn2.to_q.weight"], G[pre + "attn2.to_q.bias"])
            dx1 = dgrad(dq2, P[pre + "attn2.to_q.weight"], out=other, residual=dx2)      # dx1 = dx2 + dq2 Wq
            wkv, gkv = self._fused(pre + "attn2.to_k.weight", 2 * D, D)
            _, gbkv = self._fused(pre + "attn2.to_k.bias", 2 * D)

            def text_grads(dkv2=dkv2, wkv=wkv, gkv=gkv, gbkv=gbkv, first=(i == cfg.num_layers - 1)):
                wgrad(dkv2, S.encp, gkv, gbkv)
                dgrad(dkv2, wkv, out=denc, residual=None if first else denc)      # the text-side chain lives on this stream
            off_chain(text_grads)
            # x1 = x + gate_msa * lin1
            dlin1 = buf(f"dlin1.{par}", (M, D))
            ops.gate_bwd(dx1, A.lin1, mod2d[:, 2 * D:3 * D], 6 * D, N, dlin1, dmod2d[:, 2 * D:3 * D], 6 * D, ws_gate,
                         dbias=G[pre + "attn1.to_out.0.bias"], accumulate_bias=acc)
            wgrad(dlin1, A.attn, G[pre + "attn1.to_out.0.weight"])
            dattn = dgrad(dlin1, P[pre + "attn1.to_out.0.weight"], out=buf("do", (M, D)))
            dqkv = buf(f"dqkv.{par}", (M, 3 * D))
            ops.sdpa_bwd(A.qkv[:, :D], A.qkv[:, D:2 * D], A.qkv[:, 2 * D:], B, N, N, H, dh, scale, None, None,
                         A.attn, dattn, A.lse1, buf("delta", (B, H, N), f32), dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
            wqkv, gqkv = self._fused(pre + "attn1.to_q.weight", 3 * D, D)
            _, gbqkv = self._fused(pre + "attn1.to_q.bias", 3 * D)
            wgrad(dqkv, A.h1, gqkv, gbqkv)
            dh1 = dgrad(dqkv, wqkv, out=buf(f"dh1.{par}" if split else "dh", (M, D)))
            ln1 = (A.x_in, A.mean1, A.rstd1, mod2d[:, D:2 * D], 6 * D, N, dh1)
            dx = ops.ln_modulate_bwd(*ln1, dx1, dx, dmod2d[:, 0:D], dmod2d[:, D:2 * D], 6 * D, ws_ln, parts=1 if split else 3)
            if split:
                off_chain(lambda ln1=ln1, dmod2d=dmod2d: ops.ln_modulate_bwd(
                    *ln1, None, None, dmod2d[:, 0:D], dmod2d[:, D:2 * D], 6 * D, ws_ln, parts=2))

            def block_done(dmod=dmod, pre=pre, i=i):
                ops.modulation_bwd(dmod, G[pre + "scale_shift_table"], dtmod, D, accumulate_table=acc)
                if self.grad_ready is not None:
                    self._callback(self.grad_ready, i + 1)      # DDP hook records on the CURRENT (second) stream
            if side is None:
                block_done()
            else:
                self._wait_stream(side, main)
                if aux_used[0]:
                    self._wait_stream(side, aux)      # this bucket's bias gradients; the buffers they read (parity sets)
                    aux_used[0] = False
                with torch.cuda.stream(side):
                    block_done()
                    set_done[par] = self._ev_record(side)
        # ---- embedders (small: back on one stream)
        if side is not None:
            if aux_used[0]:
                self._wait_stream(side, aux)
            self._wait_stream(main, side)
            side = aux = None
        wgrad(dx, S.x_tok, G["pos_embed.proj.weight"].view(D, Kp), G["pos_embed.proj.bias"], dgrad_follows=False)   # + pos_embed: identity
        # caption branch: encp = linear_2(gelu_tanh(linear_1(enc)))
        wgrad(denc, S.c1, G["caption_projection.linear_2.weight"], G["caption_projection.linear_2.bias"])
        dc1 = dgrad(denc, P["caption_projection.linear_2.weight"], out=buf("dc1", (Mt, D)))
        dzc1 = ops.act_bwd(S.zc1, dc1, "gelu_tanh", buf("dzc1", (Mt, D)))
        wgrad(dzc1, S.enc2d, G["caption_projection.linear_1.weight"], G["caption_projection.linear_1.bias"], dgrad_follows=False)
        # timestep branch
        pre = "adaln_single."
        dtmod_b = ops.f32_to_bf16(dtmod, buf("dtmod_b", (B, 6 * D)))
        wgrad(dtmod_b, S.se, G[pre + "linear.weight"], G[pre + "linear.bias"])
        dse = dgrad(dtmod_b, P[pre + "linear.weight"], out=buf("te_d1", (B, D)))
        demb_a = ops.act_bwd(S.embedded, dse, "silu", buf("te_d2", (B, D)))
        demb_b = ops.f32_to_bf16(demb, buf("te_d3", (B, D)))
        d_emb = ops.add_bf16(demb_a, demb_b, buf("te_d1", (B, D)))
        wgrad(d_emb, S.e1, G[pre + "emb.timestep_embedder.linear_2.weight"], G[pre + "emb.timestep_embedder.linear_2.bias"])
        de1 = dgrad(d_emb, P[pre + "emb.timestep_embedder.linear_2.weight"], out=buf("te_d2", (B, D)))
        dz1 = ops.act_bwd(S.z1, de1, "silu", buf("te_d3", (B, D)))
        wgrad(dz1, S.tproj, G[pre + "emb.timestep_embedder.linear_1.weight"], G[pre + "emb.timestep_embedder.linear_1.bias"],
              dgrad_follows=False)
        if self.grad_ready is not None:
            self._callback(self.grad_ready, 0)
        if ad is not None:
            assert not pending_ad, "an adapter weight gradient was queued without a following dgrad()"
            ad.project()                  # adapter gradients complete (LoKr: d_P -> d_w1, d_w2_a; DDP hook)

    # ------------------------------------------------------------------ checkpoint I/O (diffusers layout)
    def save_pretrained(self, path):
        import json
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.join_pending_update()
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        base = self.cfg.sample_size // self.cfg.patch_size
        sd["pos_embed.pos_embed"] = sincos_pos_embed(self.cfg.inner_dim, base, base, base, self.cfg.interp)[None].to(BF16)
        save_file(sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        cfgd = asdict(self.cfg)
        cfgd.update({"_class_name": "PixArtTransformer2DModel", "attention_bias": True, "dropout": 0.0,
                     "activation_fn": "gelu-approximate", "norm_type": "ada_norm_single", "norm_elementwise_affine": False,
                     "num_embeds_ada_norm": 1000, "upcast_attention": False, "attention_type": "default"})
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfgd, f, indent=2)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        state_dict = {k: v for k, v in state_dict.items() if k != "pos_embed.pos_embed"}     # a buffer: recomputed here
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    @classmethod
    def from_pretrained(cls, path, device="cuda", **_):
        import json
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        known = {k: raw[k] for k in PixArtConfig.__dataclass_fields__ if k in raw and raw[k] is not None}
        if raw.get("use_additional_conditions") is None and raw.get("sample_size") == 128:
            known["use_additional_conditions"] = True     # [RECALL] the ctor's default for 1024 px when the key is absent
        model = cls(PixArtConfig(**known), device=device)
        model.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return model
