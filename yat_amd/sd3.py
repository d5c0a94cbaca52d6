"""SD3.5 MMDiT transformer on the HIP C-ABI: explicit forward and hand-scheduled backward (BASELINE config 4).

Mirrors diffusers' ``SD3Transformer2DModel`` as the reference trains it (/root/reference/train_sd35.py:28-43,58 loads it,
:188-191 calls ``model(noisy, encoder_hidden_states=, timestep=, pooled_projections=).sample``): same call contract, same
``state_dict()`` keys (``pos_embed.{proj,pos_embed}``, ``time_text_embed.{timestep_embedder,text_embedder}.linear_{1,2}``,
``context_embedder``, ``transformer_blocks.i.{norm1.linear, norm1_context.linear, attn.{to_q,to_k,to_v,norm_q,norm_k,
add_q_proj,add_k_proj,add_v_proj,norm_added_q,norm_added_k,to_out.0,to_add_out}, attn2.*, ff.net.{0.proj,2},
ff_context.net.{0.proj,2}}``, ``norm_out.linear``, ``proj_out``).  The model source is not vendored in the reference: the
block internals are [RECALL] (oracle/sd3_ref.py restates them; parity unpinned).

Same MI355X design as yat_amd/sana.py / pixart.py: one flat bf16 parameter buffer in forward order (q|k|v and the text
side's add_q|add_k|add_v back to back, so each is ONE [3D, D] GEMM), token-major rows, every activation kept (no
recompute), straight-line C-ABI launches, weight gradients on a second stream.  What is new for MMDiT:

* two residual streams (image tokens [B*N, D], text tokens [B*T, D]) that meet only inside the joint attention: the q | k
  projections of both get their per-head RMSNorm and are written, with v, into ONE joint buffer laid out [image tokens |
  text tokens] per image (``yat_qknorm_concat_fwd``) -- the ``torch.cat(dim=2)`` of JointAttnProcessor2_0 costs no extra
  pass over q and k; the flash kernels of csrc/sdpa.hip run on the joint rows; ``yat_joint_rows`` splits the output;
* AdaLN-Zero: every block's modulation is a Linear of silu(temb) -- [B, 6D] (9D with the second attention of the
  dual-attention blocks, 2D for the last block's text side and the output norm).  They depend on temb only, so all 2L + 1 of
  them are issued before the first block (second stream), and LayerNorm-modulate / gate epilogues read their column blocks;
* ``attn2`` of the dual-attention blocks is the same machinery with T = 0.
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
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    def validate(self):
        D, p = self.inner_dim, self.patch_size
        if self.caption_projection_dim != D:
            raise ValueError("caption_projection_dim must equal the model dim")
        if self.qk_norm != "rms_norm":
            raise ValueError("qk_norm='rms_norm' only (SD3.5); SD3.0's un-normalised attention is not built")
        if self.attention_head_dim not in (32, 64, 128):
            raise ValueError("attention head dim must be 32, 64 or 128")
        if D % 8 or self.joint_attention_dim % 8 or self.pooled_projection_dim % 8 or (self.in_channels * p * p) % 8 \
                or (self.out_channels * p * p) % 4:
            raise ValueError("channel sizes must be multiples of 8 (16-byte vector accesses)")
        if any(not (0 <= int(i) < self.num_layers) for i in self.dual_attention_layers):
            raise ValueError("dual_attention_layers out of range")
        if self.num_layers < 1:
            raise ValueError("num_layers >= 1")


def _param_specs(cfg: SD3Config):
    """(diffusers key, shape) in forward-execution order."""
    D, p, dh = cfg.inner_dim, cfg.patch_size, cfg.attention_head_dim
    specs = [
        ("pos_embed.proj.weight", (D, cfg.in_channels, p, p)), ("pos_embed.proj.bias", (D,)),
        ("time_text_embed.timestep_embedder.linear_1.weight", (D, 256)), ("time_text_embed.timestep_embedder.linear_1.bias", (D,)),
        ("time_text_embed.timestep_embedder.linear_2.weight", (D, D)), ("time_text_embed.timestep_embedder.linear_2.bias", (D,)),
        ("time_text_embed.text_embedder.linear_1.weight", (D, cfg.pooled_projection_dim)),
        ("time_text_embed.text_embedder.linear_1.bias", (D,)),
        ("time_text_embed.text_embedder.linear_2.weight", (D, D)), ("time_text_embed.text_embedder.linear_2.bias", (D,)),
        ("context_embedder.weight", (D, cfg.joint_attention_dim)), ("context_embedder.bias", (D,)),
    ]

    def attn(b, name, joint, add_out):
        s = [(b + f"{name}.to_{x}.weight", (D, D)) for x in "qkv"] + [(b + f"{name}.to_{x}.bias", (D,)) for x in "qkv"]
        s += [(b + f"{name}.norm_q.weight", (dh,)), (b + f"{name}.norm_k.weight", (dh,))]
        if joint:
            s += [(b + f"{name}.add_{x}_proj.weight", (D, D)) for x in "qkv"]
            s += [(b + f"{name}.add_{x}_proj.bias", (D,)) for x in "qkv"]
            s += [(b + f"{name}.norm_added_q.weight", (dh,)), (b + f"{name}.norm_added_k.weight", (dh,))]
        s += [(b + f"{name}.to_out.0.weight", (D, D)), (b + f"{name}.to_out.0.bias", (D,))]
        if add_out:
            s += [(b + f"{name}.to_add_out.weight", (D, D)), (b + f"{name}.to_add_out.bias", (D,))]
        return s

    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        last, dual = i == cfg.num_layers - 1, i in cfg.dual_attention_layers
        n1, nc = (9 if dual else 6), (2 if last else 6)
        specs += [(b + "norm1.linear.weight", (n1 * D, D)), (b + "norm1.linear.bias", (n1 * D,)),
                  (b + "norm1_context.linear.weight", (nc * D, D)), (b + "norm1_context.linear.bias", (nc * D,))]
        specs += attn(b, "attn", True, not last)
        if dual:
            specs += attn(b, "attn2", False, False)
        specs += [(b + "ff.net.0.proj.weight", (4 * D, D)), (b + "ff.net.0.proj.bias", (4 * D,)),
                  (b + "ff.net.2.weight", (D, 4 * D)), (b + "ff.net.2.bias", (D,))]
        if not last:
            specs += [(b + "ff_context.net.0.proj.weight", (4 * D, D)), (b + "ff_context.net.0.proj.bias", (4 * D,)),
                      (b + "ff_context.net.2.weight", (D, 4 * D)), (b + "ff_context.net.2.bias", (D,))]
    specs += [("norm_out.linear.weight", (2 * D, D)), ("norm_out.linear.bias", (2 * D,)),
              ("proj_out.weight", (p * p * cfg.out_channels, D)), ("proj_out.bias", (p * p * cfg.out_channels,))]
    return specs


def sincos_crop(embed_dim, max_size, base_size, top, left, h, w, device="cpu"):
    """[RECALL diffusers PatchEmbed.cropped_pos_embed over get_2d_sincos_pos_embed(embed_dim, pos_embed_max_size,
    base_size=sample_size // patch_size)] the [h*w, D] centre crop of the max-size table, computed for the crop only (the
    full 384 x 384 x 1536 table is 0.9 GB in fp32): channels [0, D/2) encode the column coordinate, [D/2, D) the row, each
    half [sin | cos] over omega_k = 10000^(-k / (D/4)), float64."""
    q = embed_dim // 4
    omega = 1.0 / 10000 ** (torch.arange(q, dtype=torch.float64, device=device) / q)
    gh = (torch.arange(top, top + h, dtype=torch.float32, device=device) / (max_size / base_size)).double()
    gw = (torch.arange(left, left + w, dtype=torch.float32, device=device) / (max_size / base_size)).double()
    col = gw[None, :].expand(h, w).reshape(-1, 1) * omega[None]
    row = gh[:, None].expand(h, w).reshape(-1, 1) * omega[None]
    return torch.cat([col.sin(), col.cos(), row.sin(), row.cos()], dim=1).float()


class _WholeModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, latents, enc, pooled, timestep):
        ctx.model = model
        return model.forward_impl(latents, enc, pooled, timestep).clone()          # (the prediction lives in the arena)

    @staticmethod
    def backward(ctx, dout):
        ctx.model.backward_impl(dout.contiguous())
        return None, None, None, None, None, None


class SD3Transformer2DModelHIP(FlatParamModule):
    def __init__(self, cfg: SD3Config | None = None, device="cuda", **cfg_kw):
        super().__init__()
        cfg = cfg or SD3Config(**cfg_kw)
        cfg.dual_attention_layers = tuple(int(i) for i in cfg.dual_attention_layers)
        cfg.validate()
        self.cfg = cfg
        self.config = SimpleNamespace(**asdict(cfg))
        specs = _param_specs(cfg)
        offs, total = self._alloc_flat(specs, device, bucket_first=lambda k: k.startswith("transformer_blocks.") and k.split(".", 2)[2] == "norm1.linear.weight")
        self.bucket_bounds = self._block_buckets(specs, offs, total, cfg.num_layers, first_key="norm1.linear.weight")
        # weight gradients / modulation linears on a 2nd stream (flat.schedule)
        # (a third stream for the text tokens' own chain was measured in round 2 -- 405.0 vs 399.7 ms per step, no gain -- and
        # is gone).  Forward as independent chains over image ranges, as in yat_amd/sana.py: two chains won in round 3 (350.5 ->
        # 343.9 ms); on the round-4 GEMM schedule ONE chain does -- 347.8 / 348.5 ms against 351.7 / 352.7 (two) and 349.1 /
        # 351.7 (three), same box, two rounds (profiles/r04_i_*): M = 35 k rows per launch already fill the chip
        self.side_wgrad, self.fwd_chains = schedule(1)
        self._pos = {}

    def init_synthetic(self, seed: int = 0):
        """Deterministic random weights of the right scale (no checkpoints offline): N(0, 1/fan_in) matrices, small biases,
        q/k norm weights near 1, modulation biases around 0.3 so that gates and scales are O(1) as in a trained model."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.P.items():
                if ".norm_q." in name or ".norm_k." in name or ".norm_added_" in name:
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=self.dev))
                elif p.ndim == 1:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.dev)
                            + (0.3 if ("norm1" in name or "norm_out" in name) else 0.0))
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device=self.dev) / math.sqrt(p[0].numel()))
        return self

    def pos_table(self, h, w):
        """Centre crop of the (bf16, as a bf16 pipeline holds the persistent buffer) position table for an h x w token grid,
        cached per bucket as device fp32 [h*w, D]: bf16(x + bf16 table) is what ``(latent + pos_embed).to(dtype)`` computes."""
        key = (h, w)
        if key not in self._pos:
            c = self.cfg
            mx = c.pos_embed_max_size
            if h > mx or w > mx:
                raise ValueError(f"token grid {h}x{w} exceeds pos_embed_max_size {mx}")
            t = sincos_crop(c.inner_dim, mx, c.sample_size // c.patch_size, (mx - h) // 2, (mx - w) // 2, h, w)
            self._pos[key] = t.to(BF16).float().to(self.dev).contiguous()
        return self._pos[key]

    # ------------------------------------------------------------------ public forward (reference call contract)
    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, return_dict=True,
                **unused):
        if torch.is_grad_enabled():
            out = _WholeModel.apply(self._anchor, self, hidden_states, encoder_hidden_states, pooled_projections, timestep)
        else:
            out = self.forward_impl(hidden_states, encoder_hidden_states, pooled_projections, timestep).clone()
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def _block_meta(self, i):
        cfg = self.cfg
        last, dual = i == cfg.num_layers - 1, i in cfg.dual_attention_layers
        return last, dual, (9 if dual else 6), (2 if last else 6)

    # ------------------------------------------------------------------ device path (launch plans, yat_amd/flat.py)
    def _schedule_flags(self):
        return (self.side_wgrad, self.fwd_chains, self.training)

    def forward_device(self, latents, enc, pooled, timestep):
        """``forward_impl`` on device-resident inputs in persistent buffers, replayed from a launch plan when this (shapes,
        addresses, schedule) combination has run before (yat_amd/sana.py does the same).  The prediction is an arena buffer:
        consume it before the next call."""
        pev = self.param_events
        self._require_device(latents=(latents, BF16), enc=(enc, BF16), pooled=(pooled, BF16), timestep=(timestep, torch.float32))
        key = (latents.data_ptr(), tuple(latents.shape), enc.data_ptr(), tuple(enc.shape), pooled.data_ptr(), timestep.data_ptr(),
               None if pev is None else id(pev[0]), self._schedule_flags())
        out = self.planned("fwd", key, lambda: self.forward_impl(latents, enc, pooled, timestep))
        self.param_events = None          # consumed by the forward (recorded or replayed)
        return out

    def backward_device(self, dpred):
        key = (id(self._saved), dpred.data_ptr(), self.accumulate_grads, id(self.grad_ready), self._schedule_flags())
        self.planned("bwd", key, lambda: self.backward_impl(dpred))

    # ------------------------------------------------------------------ forward
    def forward_impl(self, latents, enc, pooled, timestep):
        cfg, P = self.cfg, self.P
        ad = self.adapters
        if ad is not None:
            ad.materialize(self.training)                     # yat_amd/lora.py / lokr.py / loha.py: this step's adapter state
        D, H, dh, p = cfg.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim, cfg.patch_size
        B, Cin, Hl, Wl = latents.shape
        if Hl % p or Wl % p:
            raise ValueError("latent size must be a multiple of patch_size")
        h, w = Hl // p, Wl // p
        N, M = h * w, B * h * w
        T = enc.shape[1]
        Mt, L = B * T, N + T
        Kp, Co = Cin * p * p, p * p * cfg.out_channels
        dev, f32 = self.dev, torch.float32
        latents = latents.to(device=dev, dtype=BF16).contiguous()
        enc2d = enc.to(device=dev, dtype=BF16).contiguous().view(Mt, -1)
        pooled = pooled.to(device=dev, dtype=BF16).contiguous()
        t_f32 = timestep.to(device=dev, dtype=f32).contiguous()
        S = SimpleNamespace(B=B, h=h, w=w, N=N, M=M, T=T, Mt=Mt, L=L, Hl=Hl, Wl=Wl, enc2d=enc2d, pooled=pooled, blocks=[])
        buf = self._buf
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None
        pev, self.param_events = self.param_events, None
        # (with adapters: one chain, so that an adapter's products cover the whole batch and the backward can reuse them)
        nchain = 1 if ad is not None else max(1, min(self.fwd_chains, B))
        ops.gemm_concurrency(nchain)              # the GEMM policy plans each launch for its share of the chip

        def lin(x_, w_, bias_=None, out=None, **ep):
            """Linear of a (possibly adapted) target: the adapter term is folded in through the GEMM's pre_add epilogue."""
            return adapted_linear(ad, x_, w_, bias_, out=out, **ep)

        def params_ready(bucket, stream=main):
            if pev is not None:
                self._ev_wait(stream, pev[bucket])

        params_ready(0)
        # 1. PatchEmbed: p x p patches as rows -> GEMM -> + centre crop of the position table
        S.x_tok = ops.patch_rearrange(latents, buf("x_tok", (M, Kp)), B, Cin, Hl, Wl, p, True, True)
        x0 = lin(S.x_tok, P["pos_embed.proj.weight"].view(D, Kp), P["pos_embed.proj.bias"], out=buf("x0", (M, D)))
        ops.add_pos_embed(x0, self.pos_table(h, w))
        # 2. conditioning: temb = TimestepEmbedding(sinusoid(t)) + TextProjection(pooled); every AdaLN takes silu(temb)
        pre = "time_text_embed."
        S.tproj = ops.timestep_embed(t_f32, 256, buf("tproj", (B, 256)))
        S.z1 = buf("te_z1", (B, D))
        S.e1 = lin(S.tproj, P[pre + "timestep_embedder.linear_1.weight"], P[pre + "timestep_embedder.linear_1.bias"],
                   out=buf("te_e1", (B, D)), activation="silu", aux_out=S.z1)
        t_emb = lin(S.e1, P[pre + "timestep_embedder.linear_2.weight"], P[pre + "timestep_embedder.linear_2.bias"],
                    out=buf("te_t", (B, D)))
        S.zp = buf("te_zp", (B, D))
        S.p1 = lin(pooled, P[pre + "text_embedder.linear_1.weight"], P[pre + "text_embedder.linear_1.bias"],
                   out=buf("te_p1", (B, D)), activation="silu", aux_out=S.zp)
        p_emb = lin(S.p1, P[pre + "text_embedder.linear_2.weight"], P[pre + "text_embedder.linear_2.bias"],
                    out=buf("te_p", (B, D)))
        S.temb = ops.add_bf16(t_emb, p_emb, buf("temb", (B, D)))
        S.se = ops.act_fwd(S.temb, "silu", buf("te_se", (B, D)))
        # 3. text tokens -> model width
        c0 = lin(enc2d, P["context_embedder.weight"], P["context_embedder.bias"], out=buf("c0", (Mt, D)))

        # 4. every AdaLN modulation (a Linear of silu(temb) each): independent of the token streams -> second stream, up front
        S.mod_ready = []

        def modulations():
            cur = torch.cuda.current_stream()
            for i in range(cfg.num_layers):
                b_ = f"transformer_blocks.{i}."
                _, _, n1, nc = self._block_meta(i)
                params_ready(i + 1, cur)
                e1 = lin(S.se, P[b_ + "norm1.linear.weight"], P[b_ + "norm1.linear.bias"], out=buf(f"b{i}.emb1", (B, n1 * D)))
                ec = lin(S.se, P[b_ + "norm1_context.linear.weight"], P[b_ + "norm1_context.linear.bias"],
                         out=buf(f"b{i}.embc", (B, nc * D)))
                S.blocks.append(SimpleNamespace(emb1=e1, embc=ec))
                if side is not None:
                    S.mod_ready.append(self._ev_record(cur))
            S.embf = lin(S.se, P["norm_out.linear.weight"], P["norm_out.linear.bias"], out=buf("embf", (B, 2 * D)))
            if side is not None:
                S.embf_ready = self._ev_record(cur)

        if side is not None:
            self._wait_stream(side, main)
            with torch.cuda.stream(side):
                modulations()
        else:
            modulations()

        scale = 1.0 / math.sqrt(dh)
        eps = 1e-6
        # Activations live in whole-batch buffers (the backward runs on the whole batch); the forward walks them as
        # ``fwd_chains`` independent chains over disjoint image ranges, each on its own stream: nothing in the MMDiT mixes
        # images (the joint attention is per image), so while one chain sits in a memory-bound kernel the other one's GEMM
        # has the matrix cores, and a chain's single-round GEMMs no longer leave the other CUs idle (yat_amd/sana.py).
        for i in range(cfg.num_layers):
            last, dual, n1, nc = self._block_meta(i)
            A = S.blocks[i]
            A.last, A.dual, A.n1, A.nc = last, dual, n1, nc
            A.x_in = x0 if i == 0 else S.blocks[i - 1].x3
            A.c_in = c0 if i == 0 else S.blocks[i - 1].c3

            def nb(tag, shape, dtype=BF16, i=i):
                return buf(f"b{i}.{tag}", shape, dtype)
            A.h1, A.mean1, A.rstd1 = nb("h1", (M, D)), nb("h1.mean", (M,), f32), nb("h1.rstd", (M,), f32)
            A.hc, A.cmean1, A.crstd1 = nb("hc", (Mt, D)), nb("hc.mean", (Mt,), f32), nb("hc.rstd", (Mt,), f32)
            A.qkv_c, A.qkv = nb("qkv_c", (Mt, 3 * D)), nb("qkv", (M, 3 * D))
            A.joint, A.jrstd = nb("joint", (B * L, 3 * D)), nb("jrstd", (B * L, 2 * H), f32)
            A.o, A.lse = nb("o", (B * L, D)), nb("lse", (B, H, L), f32)
            A.o_i, A.o_c = nb("o_i", (M, D)), (None if last else nb("o_c", (Mt, D)))
            A.lin1, A.x1 = nb("lin1", (M, D)), nb("x1", (M, D))
            if dual:
                A.h1b, A.qkv2 = nb("h1b", (M, D)), nb("qkv2", (M, 3 * D))
                A.j2, A.j2rstd = nb("j2", (M, 3 * D)), nb("j2rstd", (M, 2 * H), f32)
                A.o2, A.lse2 = nb("o2", (M, D)), nb("lse2", (B, H, N), f32)
                A.lin1b, A.x1b = nb("lin1b", (M, D)), nb("x1b", (M, D))
            A.xa = A.x1b if dual else A.x1
            A.h2, A.mean2, A.rstd2 = nb("h2", (M, D)), nb("h2.mean", (M,), f32), nb("h2.rstd", (M,), f32)
            A.z, A.f1 = nb("z", (M, 4 * D)), nb("f1", (M, 4 * D))
            A.lin3, A.x3 = nb("lin3", (M, D)), nb("x3", (M, D))
            A.c3 = None
            if not last:
                A.clin1, A.c1 = nb("clin1", (Mt, D)), nb("c1", (Mt, D))
                A.hc2, A.cmean2, A.crstd2 = nb("hc2", (Mt, D)), nb("hc2.mean", (Mt,), f32), nb("hc2.rstd", (Mt,), f32)
                A.zc, A.fc = nb("zc", (Mt, 4 * D)), nb("fc", (Mt, 4 * D))
                A.clin3, A.c3 = nb("clin3", (Mt, D)), nb("c3", (Mt, D))
        S.x_last = S.blocks[-1].x3
        S.hf, S.meanf, S.rstdf = buf("hf", (M, D)), buf("meanf", (M,), f32), buf("rstdf", (M,), f32)
        out_tok = buf("out_tok", (M, Co))
        pred = buf("pred", (B, cfg.out_channels, Hl, Wl))     # (arena: the caller consumes it before the next forward)
        ln_sm, ln_sr = buf("ln_scratch_mean", (M,), f32), buf("ln_scratch_rstd", (M,), f32)

        def run_chain(b0, b1, stream):
            nbt = b1 - b0
            rs, ts, js, bs = slice(b0 * N, b1 * N), slice(b0 * T, b1 * T), slice(b0 * L, b1 * L), slice(b0, b1)
            for i in range(cfg.num_layers):
                b_ = f"transformer_blocks.{i}."
                A = S.blocks[i]
                last, dual, n1, nc = A.last, A.dual, A.n1, A.nc
                params_ready(i + 1, stream)
                if side is not None:
                    self._ev_wait(stream, S.mod_ready[i])
                e1, ec = A.emb1[bs], A.embc[bs]
                ld1, ldc = n1 * D, nc * D
                x, c = A.x_in[rs], A.c_in[ts]
                # AdaLayerNormZero on both streams (the last block's text side: AdaLayerNormContinuous, scale first)
                ops.ln_modulate_fwd(x, e1[:, 0:D], e1[:, D:2 * D], ld1, N, eps, A.h1[rs], A.mean1[rs], A.rstd1[rs])
                # joint attention: fused q|k|v projections of both streams -> per-head RMSNorm on q, k + row concatenation
                wqkv, _ = self._fused(b_ + "attn.to_q.weight", 3 * D, D)
                bqkv, _ = self._fused(b_ + "attn.to_q.bias", 3 * D)
                waqkv, _ = self._fused(b_ + "attn.add_q_proj.weight", 3 * D, D)
                baqkv, _ = self._fused(b_ + "attn.add_q_proj.bias", 3 * D)
                if last:
                    ops.ln_modulate_fwd(c, ec[:, D:2 * D], ec[:, 0:D], ldc, T, eps, A.hc[ts], A.cmean1[ts], A.crstd1[ts])
                else:
                    ops.ln_modulate_fwd(c, ec[:, 0:D], ec[:, D:2 * D], ldc, T, eps, A.hc[ts], A.cmean1[ts], A.crstd1[ts])
                lin(A.hc[ts], waqkv, baqkv, out=A.qkv_c[ts])
                lin(A.h1[rs], wqkv, bqkv, out=A.qkv[rs])
                ops.qknorm_concat_fwd(A.qkv[rs], A.qkv_c[ts], nbt, N, T, H, dh, eps, P[b_ + "attn.norm_q.weight"],
                                      P[b_ + "attn.norm_k.weight"], P[b_ + "attn.norm_added_q.weight"],
                                      P[b_ + "attn.norm_added_k.weight"], A.joint[js], A.jrstd[js])
                jt = A.joint[js]
                ops.sdpa_fwd(jt[:, :D], jt[:, D:2 * D], jt[:, 2 * D:], nbt, L, L, H, dh, scale, None, None,
                             A.o[js], A.lse[bs])          # (no mask in JointAttnProcessor2_0: the no-bias instantiations)
                ops.joint_rows(A.o[js], A.o_i[rs], None if last else A.o_c[ts], nbt, N, T, to_joint=False)
                # hidden = hidden + gate_msa * to_out(attn)
                lin(A.o_i[rs], P[b_ + "attn.to_out.0.weight"], P[b_ + "attn.to_out.0.bias"], out=A.x1[rs],
                    aux_out=A.lin1[rs], gate=e1[:, 2 * D:3 * D], ld_gate=ld1, residual=x, rows_per_batch=N)
                if dual:
                    # second, image-only attention from the same LayerNorm with its own (shift, scale, gate)
                    ops.ln_modulate_fwd(x, e1[:, 6 * D:7 * D], e1[:, 7 * D:8 * D], ld1, N, eps, A.h1b[rs], ln_sm[rs], ln_sr[rs])
                    w2, _ = self._fused(b_ + "attn2.to_q.weight", 3 * D, D)
                    b2, _ = self._fused(b_ + "attn2.to_q.bias", 3 * D)
                    lin(A.h1b[rs], w2, b2, out=A.qkv2[rs])
                    ops.qknorm_concat_fwd(A.qkv2[rs], None, nbt, N, 0, H, dh, eps, P[b_ + "attn2.norm_q.weight"],
                                          P[b_ + "attn2.norm_k.weight"], None, None, A.j2[rs], A.j2rstd[rs])
                    j2 = A.j2[rs]
                    ops.sdpa_fwd(j2[:, :D], j2[:, D:2 * D], j2[:, 2 * D:], nbt, N, N, H, dh, scale, None, None,
                                 A.o2[rs], A.lse2[bs])
                    lin(A.o2[rs], P[b_ + "attn2.to_out.0.weight"], P[b_ + "attn2.to_out.0.bias"], out=A.x1b[rs],
                        aux_out=A.lin1b[rs], gate=e1[:, 8 * D:9 * D], ld_gate=ld1, residual=A.x1[rs], rows_per_batch=N)
                xa = A.xa[rs]
                # image feed-forward
                ops.ln_modulate_fwd(xa, e1[:, 3 * D:4 * D], e1[:, 4 * D:5 * D], ld1, N, eps, A.h2[rs], A.mean2[rs], A.rstd2[rs])
                lin(A.h2[rs], P[b_ + "ff.net.0.proj.weight"], P[b_ + "ff.net.0.proj.bias"], out=A.f1[rs],
                    activation="gelu_tanh", aux_out=A.z[rs])
                lin(A.f1[rs], P[b_ + "ff.net.2.weight"], P[b_ + "ff.net.2.bias"], out=A.x3[rs], aux_out=A.lin3[rs],
                    gate=e1[:, 5 * D:6 * D], ld_gate=ld1, residual=xa, rows_per_batch=N)
                # text stream (ends inside the attention of the last block)
                if not last:
                    lin(A.o_c[ts], P[b_ + "attn.to_add_out.weight"], P[b_ + "attn.to_add_out.bias"], out=A.c1[ts],
                        aux_out=A.clin1[ts], gate=ec[:, 2 * D:3 * D], ld_gate=ldc, residual=c, rows_per_batch=T)
                    ops.ln_modulate_fwd(A.c1[ts], ec[:, 3 * D:4 * D], ec[:, 4 * D:5 * D], ldc, T, eps, A.hc2[ts], A.cmean2[ts],
                                        A.crstd2[ts])
                    lin(A.hc2[ts], P[b_ + "ff_context.net.0.proj.weight"], P[b_ + "ff_context.net.0.proj.bias"], out=A.fc[ts],
                        activation="gelu_tanh", aux_out=A.zc[ts])
                    lin(A.fc[ts], P[b_ + "ff_context.net.2.weight"], P[b_ + "ff_context.net.2.bias"], out=A.c3[ts],
                        aux_out=A.clin3[ts], gate=ec[:, 5 * D:6 * D], ld_gate=ldc, residual=A.c1[ts], rows_per_batch=T)
            # 5. output head: AdaLayerNormContinuous (scale first) + proj_out + unpatchify
            if side is not None:
                self._ev_wait(stream, S.embf_ready)
            embf = S.embf[bs]
            ops.ln_modulate_fwd(S.x_last[rs], embf[:, D:2 * D], embf[:, 0:D], 2 * D, N, eps, S.hf[rs], S.meanf[rs], S.rstdf[rs])
            lin(S.hf[rs], P["proj_out.weight"], P["proj_out.bias"], out=out_tok[rs])
            ops.patch_rearrange(out_tok[rs], pred[bs], nbt, cfg.out_channels, Hl, Wl, p, False, False)

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
        B, N, M, T, Mt, L = S.B, S.N, S.M, S.T, S.Mt, S.L
        Kp, Co = cfg.in_channels * p * p, p * p * cfg.out_channels
        acc = self.accumulate_grads
        buf = self._buf
        f32, u8 = torch.float32, torch.uint8
        lib = ops._lib()
        ws_col = buf("ws_col", (int(lib.yat_colsum_workspace_bytes(max(M, Mt), 9 * D)),), u8)
        ws_ln = buf("ws_ln", (ops.ln_bwd_workspace_bytes(M, D, N),), u8)
        ws_lnc = buf("ws_lnc", (ops.ln_bwd_workspace_bytes(Mt, D, T),), u8)
        ws_gate = buf("ws_gate", (int(lib.yat_gate_bwd_workspace_bytes(M, D, N)),), u8)
        ws_gatec = buf("ws_gatec", (int(lib.yat_gate_bwd_workspace_bytes(Mt, D, T)),), u8)
        ws_qk = buf("ws_qk", (ops.qknorm_concat_bwd_workspace_bytes(B, N, T, dh),), u8)
        scale = 1.0 / math.sqrt(dh)
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None

        def off_chain(fn):
            """Weight / bias / modulation gradients: nothing on the dependent chain reads them -> second stream, right
            behind their producer."""
            if side is None:
                fn()
                return
            self._wait_stream(side, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()

        ad = self.adapters
        pending_ad = []                   # adapter weight gradients wait for the H product of the dgrad of the same dy

        def wgrad(dy, x, gw, gbias=None, dgrad_follows=True):
            if ad is not None:            # frozen base: only the adapters' share, launched by the dgrad() of the same dy
                if dgrad_follows:
                    pending_ad.append((dy, x, gw))
                else:
                    off_chain(lambda: ad.wgrad(dy, x, gw, accumulate=acc))
                return

            def run():
                ops.linear_wgrad(dy, x, gw, accumulate=acc, bias_grad=gbias, colsum_ws=ws_col)      # bias gradient: same launch
            off_chain(run)

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
        # d(silu(temb)): every modulation Linear adds its share (bf16 accumulation, as autograd sums the bf16 branches);
        # lives on the second stream (the modulation gradients are produced there)
        dse = buf("dse", (B, D))
        dse_started = [False]

        def mod_grads(demb_f32, wkey, bkey, tag):
            """demb [B, k*D] fp32 accumulators -> Linear(silu(temb)) gradients: weight, bias, and the share of d silu(temb)."""
            d_b = ops.f32_to_bf16(demb_f32, buf(f"demb_b.{tag}", tuple(demb_f32.shape)))
            if ad is None:
                ops.linear_wgrad(d_b, S.se, G[wkey], accumulate=acc, bias_grad=G[bkey], colsum_ws=ws_col)
            else:
                pending_ad.append((d_b, S.se, G[wkey]))
            dgrad(d_b, P[wkey], out=dse, residual=dse if dse_started[0] else None)
            dse_started[0] = True

        # ---- output head
        d_out_tok = ops.patch_rearrange(dpred.to(BF16).contiguous(), buf("d_out_tok", (M, Co)), B, cfg.out_channels, S.Hl,
                                        S.Wl, p, False, True)
        wgrad(d_out_tok, S.hf, G["proj_out.weight"], G["proj_out.bias"])
        dhf = dgrad(d_out_tok, P["proj_out.weight"], out=buf("dh.0", (M, D)))
        dembf = ops.zero_(buf("dembf", (B, 2 * D), f32))
        dx = ops.ln_modulate_bwd(S.x_last, S.meanf, S.rstdf, S.embf[:, 0:D], 2 * D, N, dhf, None, buf("dx.a", (M, D)),
                                 dembf[:, D:2 * D], dembf[:, 0:D], 2 * D, ws_ln)
        off_chain(lambda: mod_grads(dembf, "norm_out.linear.weight", "norm_out.linear.bias", "f"))
        dc = None                                          # the text stream ends in the last block
        set_done = [None, None]
        for i in reversed(range(cfg.num_layers)):
            b_ = f"transformer_blocks.{i}."
            A = S.blocks[i]
            last, dual, n1, nc = A.last, A.dual, A.n1, A.nc
            e1, ec = A.emb1, A.embc
            ld1, ldc = n1 * D, nc * D
            par = i & 1
            if set_done[par] is not None:
                self._ev_wait(main, set_done[par])         # block i+2's second-stream work has read this buffer set
                set_done[par] = None

            def pb(name, shape, dtype=BF16):
                return buf(f"{name}.{par}", shape, dtype)
            demb1 = ops.zero_(buf(f"demb1.{par}.{n1}", (B, ld1), f32))
            dembc = ops.zero_(buf(f"dembc.{par}.{nc}", (B, ldc), f32))
            # ---- image feed-forward: x3 = xa + gate_mlp * (f1 W2^T + b2)
            dlin3 = pb("dlin3", (M, D))
            ops.gate_bwd(dx, A.lin3, e1[:, 5 * D:6 * D], ld1, N, dlin3, demb1[:, 5 * D:6 * D], ld1, ws_gate,
                         dbias=G[b_ + "ff.net.2.bias"], accumulate_bias=acc)
            wgrad(dlin3, A.f1, G[b_ + "ff.net.2.weight"])
            df1 = dgrad(dlin3, P[b_ + "ff.net.2.weight"], out=buf("df1", (M, 4 * D)))
            dz = ops.act_bwd(A.z, df1, "gelu_tanh", pb("dz", (M, 4 * D)))
            wgrad(dz, A.h2, G[b_ + "ff.net.0.proj.weight"], G[b_ + "ff.net.0.proj.bias"])
            dh2 = dgrad(dz, P[b_ + "ff.net.0.proj.weight"], out=buf("dh.0", (M, D)))
            dxa = ops.ln_modulate_bwd(A.xa, A.mean2, A.rstd2, e1[:, 4 * D:5 * D], ld1, N, dh2, dx, pb("dxa", (M, D)),
                                      demb1[:, 3 * D:4 * D], demb1[:, 4 * D:5 * D], ld1, ws_ln)
            dh1b = None
            if dual:
                # x1b = x1 + gate_msa2 * to_out2(sdpa(norm(qkv2)))
                dlin1b = pb("dlin1b", (M, D))
                ops.gate_bwd(dxa, A.lin1b, e1[:, 8 * D:9 * D], ld1, N, dlin1b, demb1[:, 8 * D:9 * D], ld1, ws_gate,
                             dbias=G[b_ + "attn2.to_out.0.bias"], accumulate_bias=acc)
                wgrad(dlin1b, A.o2, G[b_ + "attn2.to_out.0.weight"])
                do2 = dgrad(dlin1b, P[b_ + "attn2.to_out.0.weight"], out=buf("do2", (M, D)))
                dj2 = buf("dj2", (M, 3 * D))
                ops.sdpa_bwd(A.j2[:, :D], A.j2[:, D:2 * D], A.j2[:, 2 * D:], B, N, N, H, dh, scale, None, None, A.o2,
                             do2, A.lse2, buf("delta2", (B, H, N), f32), dj2[:, :D], dj2[:, D:2 * D], dj2[:, 2 * D:])
                dqkv2 = pb("dqkv2", (M, 3 * D))
                ops.qknorm_concat_bwd(A.qkv2, None, B, N, 0, H, dh, P[b_ + "attn2.norm_q.weight"], P[b_ + "attn2.norm_k.weight"],
                                      None, None, A.j2rstd, dj2, dqkv2, None, G[b_ + "attn2.norm_q.weight"],
                                      G[b_ + "attn2.norm_k.weight"], None, None, ws_qk, accumulate_dw=acc)
                w2, g2 = self._fused(b_ + "attn2.to_q.weight", 3 * D, D)
                _, gb2 = self._fused(b_ + "attn2.to_q.bias", 3 * D)
                wgrad(dqkv2, A.h1b, g2, gb2)
                dh1b = dgrad(dqkv2, w2, out=buf("dh.1", (M, D)))
            # ---- x1 = x + gate_msa * to_out(attn image rows)
            dlin1 = pb("dlin1", (M, D))
            ops.gate_bwd(dxa, A.lin1, e1[:, 2 * D:3 * D], ld1, N, dlin1, demb1[:, 2 * D:3 * D], ld1, ws_gate,
                         dbias=G[b_ + "attn.to_out.0.bias"], accumulate_bias=acc)
            wgrad(dlin1, A.o_i, G[b_ + "attn.to_out.0.weight"])
            do_i = dgrad(dlin1, P[b_ + "attn.to_out.0.weight"], out=buf("do_i", (M, D)))
            # ---- text stream, first half: FFN and to_add_out backward -> the text rows of the attention-output gradient
            do_c = dc1 = None
            if not last:
                def text_bwd_a(dc=dc):
                    dclin3 = pb("dclin3", (Mt, D))
                    ops.gate_bwd(dc, A.clin3, ec[:, 5 * D:6 * D], ldc, T, dclin3, dembc[:, 5 * D:6 * D], ldc, ws_gatec,
                                 dbias=G[b_ + "ff_context.net.2.bias"], accumulate_bias=acc)
                    wgrad(dclin3, A.fc, G[b_ + "ff_context.net.2.weight"])
                    dfc = dgrad(dclin3, P[b_ + "ff_context.net.2.weight"], out=buf("dfc", (Mt, 4 * D)))
                    dzc = ops.act_bwd(A.zc, dfc, "gelu_tanh", pb("dzc", (Mt, 4 * D)))
                    wgrad(dzc, A.hc2, G[b_ + "ff_context.net.0.proj.weight"], G[b_ + "ff_context.net.0.proj.bias"])
                    dhc2 = dgrad(dzc, P[b_ + "ff_context.net.0.proj.weight"], out=buf("dhc.0", (Mt, D)))
                    dc1_ = ops.ln_modulate_bwd(A.c1, A.cmean2, A.crstd2, ec[:, 4 * D:5 * D], ldc, T, dhc2, dc, pb("dc1", (Mt, D)),
                                               dembc[:, 3 * D:4 * D], dembc[:, 4 * D:5 * D], ldc, ws_lnc)
                    dclin1 = pb("dclin1", (Mt, D))
                    ops.gate_bwd(dc1_, A.clin1, ec[:, 2 * D:3 * D], ldc, T, dclin1, dembc[:, 2 * D:3 * D], ldc, ws_gatec,
                                 dbias=G[b_ + "attn.to_add_out.bias"], accumulate_bias=acc)
                    wgrad(dclin1, A.o_c, G[b_ + "attn.to_add_out.weight"])
                    return dc1_, dgrad(dclin1, P[b_ + "attn.to_add_out.weight"], out=buf("do_c", (Mt, D)))
                dc1, do_c = text_bwd_a()
            # ---- joint attention backward
            do_j = buf("do_j", (B * L, D))
            ops.joint_rows(do_j, do_i, do_c, B, N, T, to_joint=True)       # (no text gradient in the last block: zeros)
            dj = buf("dj", (B * L, 3 * D))
            ops.sdpa_bwd(A.joint[:, :D], A.joint[:, D:2 * D], A.joint[:, 2 * D:], B, L, L, H, dh, scale, None,
                         None, A.o, do_j, A.lse, buf("delta", (B, H, L), f32), dj[:, :D], dj[:, D:2 * D], dj[:, 2 * D:])
            dqkv, dqkv_c = pb("dqkv", (M, 3 * D)), pb("dqkv_c", (Mt, 3 * D))
            ops.qknorm_concat_bwd(A.qkv, A.qkv_c, B, N, T, H, dh, P[b_ + "attn.norm_q.weight"], P[b_ + "attn.norm_k.weight"],
                                  P[b_ + "attn.norm_added_q.weight"], P[b_ + "attn.norm_added_k.weight"], A.jrstd, dj, dqkv,
                                  dqkv_c, G[b_ + "attn.norm_q.weight"], G[b_ + "attn.norm_k.weight"],
                                  G[b_ + "attn.norm_added_q.weight"], G[b_ + "attn.norm_added_k.weight"], ws_qk,
                                  accumulate_dw=acc)
            wqkv, gqkv = self._fused(b_ + "attn.to_q.weight", 3 * D, D)
            _, gbqkv = self._fused(b_ + "attn.to_q.bias", 3 * D)
            waqkv, gaqkv = self._fused(b_ + "attn.add_q_proj.weight", 3 * D, D)
            _, gbaqkv = self._fused(b_ + "attn.add_q_proj.bias", 3 * D)
            wgrad(dqkv, A.h1, gqkv, gbqkv)
            dh1 = dgrad(dqkv, wqkv, out=buf("dh.0", (M, D)))
            # ---- AdaLayerNormZero backward: both modulations of the image stream share one LayerNorm
            nxt = buf("dx.b", (M, D)) if dx.data_ptr() == buf("dx.a", (M, D)).data_ptr() else buf("dx.a", (M, D))
            if dual:
                tmp = ops.ln_modulate_bwd(A.x_in, A.mean1, A.rstd1, e1[:, D:2 * D], ld1, N, dh1, dxa, pb("dxin", (M, D)),
                                          demb1[:, 0:D], demb1[:, D:2 * D], ld1, ws_ln)
                dx = ops.ln_modulate_bwd(A.x_in, A.mean1, A.rstd1, e1[:, 7 * D:8 * D], ld1, N, dh1b, tmp, nxt,
                                         demb1[:, 6 * D:7 * D], demb1[:, 7 * D:8 * D], ld1, ws_ln)
            else:
                dx = ops.ln_modulate_bwd(A.x_in, A.mean1, A.rstd1, e1[:, D:2 * D], ld1, N, dh1, dxa, nxt,
                                         demb1[:, 0:D], demb1[:, D:2 * D], ld1, ws_ln)
            # ---- text stream, second half: the text side's q | k | v projection and its AdaLayerNorm backward
            def text_bwd_b(dc=dc, dc1=dc1):
                wgrad(dqkv_c, A.hc, gaqkv, gbaqkv)
                dhc = dgrad(dqkv_c, waqkv, out=buf("dhc.0", (Mt, D)))
                dcn = buf("dc.b", (Mt, D)) if (dc is not None and dc.data_ptr() == buf("dc.a", (Mt, D)).data_ptr()) \
                    else buf("dc.a", (Mt, D))
                if last:                                   # AdaLayerNormContinuous: scale first, no residual path
                    return ops.ln_modulate_bwd(A.c_in, A.cmean1, A.crstd1, ec[:, 0:D], ldc, T, dhc, None, dcn,
                                               dembc[:, D:2 * D], dembc[:, 0:D], ldc, ws_lnc)
                return ops.ln_modulate_bwd(A.c_in, A.cmean1, A.crstd1, ec[:, D:2 * D], ldc, T, dhc, dc1, dcn,
                                           dembc[:, 0:D], dembc[:, D:2 * D], ldc, ws_lnc)
            dc = text_bwd_b()

            def block_done(demb1=demb1, dembc=dembc, b_=b_, i=i):
                mod_grads(demb1, b_ + "norm1.linear.weight", b_ + "norm1.linear.bias", f"1.{i & 1}")
                mod_grads(dembc, b_ + "norm1_context.linear.weight", b_ + "norm1_context.linear.bias", f"c.{i & 1}")
                if self.grad_ready is not None:
                    self._callback(self.grad_ready, i + 1)     # DDP hook records on the CURRENT (second) stream
            if side is None:
                block_done()
            else:
                self._wait_stream(side, main)
                with torch.cuda.stream(side):
                    block_done()
                    set_done[par] = self._ev_record(side)
        # ---- embedders (small: back on one stream)
        if side is not None:
            self._wait_stream(main, side)
            side = None
        wgrad(dx, S.x_tok, G["pos_embed.proj.weight"].view(D, Kp), G["pos_embed.proj.bias"], dgrad_follows=False)   # + pos_embed: identity
        wgrad(dc, S.enc2d, G["context_embedder.weight"], G["context_embedder.bias"], dgrad_follows=False)
        # conditioning: d temb = silu'(temb) * dse; temb = timestep branch + pooled branch
        dtemb = ops.act_bwd(S.temb, dse, "silu", buf("te_d0", (B, D)))
        pre = "time_text_embed."
        wgrad(dtemb, S.e1, G[pre + "timestep_embedder.linear_2.weight"], G[pre + "timestep_embedder.linear_2.bias"])
        de1 = dgrad(dtemb, P[pre + "timestep_embedder.linear_2.weight"], out=buf("te_d1", (B, D)))
        dz1 = ops.act_bwd(S.z1, de1, "silu", buf("te_d2", (B, D)))
        wgrad(dz1, S.tproj, G[pre + "timestep_embedder.linear_1.weight"], G[pre + "timestep_embedder.linear_1.bias"],
              dgrad_follows=False)
        wgrad(dtemb, S.p1, G[pre + "text_embedder.linear_2.weight"], G[pre + "text_embedder.linear_2.bias"])
        dp1 = dgrad(dtemb, P[pre + "text_embedder.linear_2.weight"], out=buf("te_d3", (B, D)))
        dzp = ops.act_bwd(S.zp, dp1, "silu", buf("te_d4", (B, D)))
        wgrad(dzp, S.pooled, G[pre + "text_embedder.linear_1.weight"], G[pre + "text_embedder.linear_1.bias"],
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
        c = self.cfg
        mx = c.pos_embed_max_size
        sd["pos_embed.pos_embed"] = sincos_crop(c.inner_dim, mx, c.sample_size // c.patch_size, 0, 0, mx, mx,
                                                device=self.dev)[None].to(BF16).cpu()
        save_file(sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        cfgd = asdict(c)
        cfgd["dual_attention_layers"] = list(c.dual_attention_layers)
        cfgd["_class_name"] = "SD3Transformer2DModel"
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
        known = {k: raw[k] for k in SD3Config.__dataclass_fields__ if k in raw and raw[k] is not None}
        if "dual_attention_layers" in known:
            known["dual_attention_layers"] = tuple(known["dual_attention_layers"])
        model = cls(SD3Config(**known), device=device)
        model.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return model
