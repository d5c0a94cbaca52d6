"""SANA transformer on the HIP C-ABI: explicit forward and hand-scheduled backward.

Mirrors ``SanaTransformer2DModel`` as the reference trains it
(/root/reference/utils/patched_sana_transformer.py:88-167,229-349 and
/root/reference/utils/patch_sana_attention_layers.py:19-115; called at train_sana.py:210-215):
same call contract ``model(hidden_states, encoder_hidden_states=, timestep=, encoder_attention_mask=).sample``,
same ``state_dict()`` keys (diffusers checkpoint layout), an ``nn.Module`` with
``.parameters()``, ``.dtype``, ``enable_gradient_checkpointing()``.

MI355X-first design rather than an autograd graph:

* all parameters live in ONE flat bf16 HBM buffer in forward-execution order (gradients in a
  second one, AdamW states in two more): the optimizer and the gradient-norm are single launches,
  gradient buckets for the data-parallel all-reduce are contiguous slices that complete in
  reverse order during backward (``grad_ready`` callback per bucket);
* to_q/to_k/to_v (and attn2 to_k/to_v) sit back to back, so QKV is one [3D, D] GEMM while the
  checkpoint still sees three tensors;
* the token-major [B*N, C] layout is kept end to end (GLUMBConv's NCHW permutes vanish);
* every activation the backward needs is kept (288 GB HBM: no gradient checkpointing, the
  reference's recompute at train_sana.py:63 is unnecessary) in a persistent arena, so a step
  allocates nothing;
* forward/backward are straight-line sequences of C-ABI launches on the current HIP stream.

Autograd integration: ``forward`` returns a tensor attached to a single autograd node whose
backward runs :meth:`backward_impl` and writes parameter gradients straight into the flat
gradient buffer (``p.grad`` are views of it).
"""
from __future__ import annotations

import contextlib
import math
import os
from dataclasses import dataclass, field, asdict
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .flat import FlatParamModule, schedule
from .lokr import adapted_linear

BF16 = torch.bfloat16


@dataclass
class SanaConfig:
    # defaults: utils/patched_sana_transformer.py:88-112 (SANA-1.6B)
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
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim

    @property
    def ffn_hidden(self):
        return int(self.inner_dim * self.mlp_ratio)

    def validate(self):
        D = self.inner_dim
        if self.attention_head_dim != 32:
            raise ValueError("linear attention kernel is built for head dim 32 (SANA)")
        if self.patch_size != 1:
            raise ValueError("patch_size 1 only (SANA)")
        if any(not (0 <= int(b) < self.num_layers) for b in self.modified_blocks):
            raise ValueError("modified_blocks out of range")
        if self.num_cross_attention_heads * self.cross_attention_head_dim != D or self.cross_attention_dim != D:
            raise ValueError("cross attention inner dim must equal the model dim")
        if D % 8 or self.ffn_hidden % 8 or self.caption_channels % 8 or self.in_channels % 8 or self.out_channels % 4:
            raise ValueError("channel sizes must be multiples of 8 (16-byte vector accesses)")
        if self.cross_attention_head_dim > 128 or self.cross_attention_head_dim % 8:
            raise ValueError("cross attention head dim must be <= 128 and a multiple of 8")


def _param_specs(cfg: SanaConfig):
    """(diffusers key, shape) in forward-execution order."""
    D, Hc, Cc = cfg.inner_dim, cfg.ffn_hidden, cfg.caption_channels
    p = cfg.patch_size
    specs = [
        ("patch_embed.proj.weight", (D, cfg.in_channels, p, p)), ("patch_embed.proj.bias", (D,)),
        ("time_embed.emb.timestep_embedder.linear_1.weight", (D, 256)),
        ("time_embed.emb.timestep_embedder.linear_1.bias", (D,)),
        ("time_embed.emb.timestep_embedder.linear_2.weight", (D, D)),
        ("time_embed.emb.timestep_embedder.linear_2.bias", (D,)),
        ("time_embed.linear.weight", (6 * D, D)), ("time_embed.linear.bias", (6 * D,)),
        ("caption_projection.linear_1.weight", (D, Cc)), ("caption_projection.linear_1.bias", (D,)),
        ("caption_projection.linear_2.weight", (D, D)), ("caption_projection.linear_2.bias", (D,)),
        ("caption_norm.weight", (D,)),
    ]
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        specs += [
            (b + "scale_shift_table", (6, D)),
            (b + "attn1.to_q.weight", (D, D)), (b + "attn1.to_k.weight", (D, D)), (b + "attn1.to_v.weight", (D, D)),
            (b + "attn1.to_out.0.weight", (D, D)), (b + "attn1.to_out.0.bias", (D,)),
            (b + "attn2.to_q.weight", (D, D)), (b + "attn2.to_q.bias", (D,)),
            (b + "attn2.to_k.weight", (D, D)), (b + "attn2.to_v.weight", (D, D)),
            (b + "attn2.to_k.bias", (D,)), (b + "attn2.to_v.bias", (D,)),
            (b + "attn2.to_out.0.weight", (D, D)), (b + "attn2.to_out.0.bias", (D,)),
            (b + "ff.conv_inverted.weight", (2 * Hc, D, 1, 1)), (b + "ff.conv_inverted.bias", (2 * Hc,)),
            (b + "ff.conv_depth.weight", (2 * Hc, 1, 3, 3)), (b + "ff.conv_depth.bias", (2 * Hc,)),
            (b + "ff.conv_point.weight", (D, Hc, 1, 1)),
        ]
    specs += [("scale_shift_table", (2, D)), ("proj_out.weight", (cfg.out_channels, D)),
              ("proj_out.bias", (cfg.out_channels,))]
    return specs


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


class SanaTransformer2DModelHIP(FlatParamModule):
    def __init__(self, cfg: SanaConfig | None = None, device="cuda", **cfg_kw):
        super().__init__()
        cfg = cfg or SanaConfig(**cfg_kw)
        cfg.validate()
        self.cfg = cfg
        self.config = SimpleNamespace(**asdict(cfg))          # diffusers-style `.config.sample_size`
        specs = _param_specs(cfg)
        offs, total = self._alloc_flat(specs, device, bucket_first=lambda k: k.startswith("transformer_blocks.") and k.split(".", 2)[2] == "scale_shift_table")
        # bucket boundaries for data parallel reduction: head | one per block (+tail on the last)
        self.bucket_bounds = self._block_buckets(specs, offs, total, cfg.num_layers)
        # weight gradients on a second stream; independent forward chains over image ranges (flat.schedule: YAT_SERIAL=1
        # puts the whole step on one stream).  Variants measured and rejected in rounds 2-4 -- weight gradients deferred to
        # the block's end / as one grouped launch per block, the three big ones grouped -- are gone (profiles/LOG_r01_r03.md)
        self.side_wgrad, self.fwd_chains = schedule(2)
        self.keep_glu_u = True                # keep the depthwise-conv output (183 MB/block) for the backward
        self.group_small_wgrad = True         # the three D x D weight gradients of a block as one grouped launch

    def init_synthetic(self, seed: int = 0):
        """Deterministic random weights of the right scale (no checkpoints offline): weights ~ N(0, 1/fan_in),
        small biases, scale/shift tables as the ctor of the reference draws them (randn / sqrt(D))."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.P.items():
                if name == "caption_norm.weight":
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=self.dev))
                elif "scale_shift_table" in name:
                    p.copy_(torch.randn(p.shape, generator=g, device=self.dev) / p.shape[-1] ** 0.5)
                elif p.ndim == 1:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.dev))
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device=self.dev) / math.sqrt(p[0].numel()))
        return self

    # ------------------------------------------------------------------ public forward (reference call contract)
    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, encoder_attention_mask=None,
                return_dict=True, **unused):
        if torch.is_grad_enabled():
            out = _WholeModel.apply(self._anchor, self, hidden_states, encoder_hidden_states, timestep,
                                    encoder_attention_mask)
        else:
            out = self.forward_impl(hidden_states, encoder_hidden_states, timestep, encoder_attention_mask).clone()
        return SimpleNamespace(sample=out) if return_dict else (out,)

    # ------------------------------------------------------------------ device path (launch plans, yat_amd/flat.py)
    def _schedule_flags(self):
        return (self.side_wgrad, self.keep_glu_u, self.fwd_chains, self.group_small_wgrad, self.training)

    packed_text = True         # forward_device / forward_impl accept the text rows without padding (``kv_off``)

    def forward_device(self, latents, enc, timestep, key_bias, kv_len, kv_work=None, kv_off=None):
        """``forward_impl`` on device-resident inputs in persistent buffers, replayed from a launch plan when this (shapes,
        addresses, schedule) combination has run before.  The prediction is an arena buffer: consume it before the next call.
        ``kv_off`` (device int32 row offsets, one per image): ``enc`` is the packed [rows, C] text matrix -- the prompts'
        rows back to back, then fewer than 256 zero rows (see ``forward_impl``); the plan key then carries the row count."""
        pev = self.param_events
        self._require_device(latents=(latents, BF16), enc=(enc, BF16), timestep=(timestep, torch.float32),
                             key_bias=(key_bias, torch.float32), kv_len=(kv_len, torch.int32), kv_off=(kv_off, torch.int32))
        # (packed text: the row count is a dynamic integer of the plan -- ops.text_rows -- not part of its key)
        enc_shape = tuple(enc.shape) if kv_off is None else ("packed",) + tuple(enc.shape[1:])
        if kv_off is not None:
            self._text_rows = self.plan_dynamic["text_rows"] = int(enc.shape[0])
        key = (latents.data_ptr(), tuple(latents.shape), enc.data_ptr(), enc_shape, timestep.data_ptr(),
               key_bias.data_ptr(), kv_len.data_ptr(), None if kv_off is None else kv_off.data_ptr(),
               None if pev is None else id(pev[0]), self._schedule_flags())
        out = self.planned("fwd", key, lambda: self.forward_impl(latents, enc, timestep, None, key_bias=key_bias,
                                                                 kv_len=kv_len, kv_work=kv_work, kv_off=kv_off))
        self.param_events = None          # consumed by the forward (recorded or replayed)
        self._saved.kv_work = kv_work
        if kv_off is not None and self._saved.Mt != int(enc.shape[0]):
            # a REPLAYED forward hands back the saved state of the step that was recorded: its text-side views are cut for
            # that batch's row count.  A backward that is recorded now (first micro-step with another accumulate flag, a new
            # gradient hook, ...) must see THIS batch's rows, or it would sum stale rows into the text-side gradients.
            self._retarget_text(self._saved, enc)
        return out

    def _retarget_text(self, S, enc):
        """Re-cut the text-side views of a saved forward for the packed row count of ``enc`` (same addresses: the buffers
        are sized for the padded layout once, ``Mt_cap``)."""
        Mt, D, f32 = int(enc.shape[0]), self.cfg.inner_dim, torch.float32
        if Mt > S.Mt_cap:
            raise RuntimeError("packed text rows exceed the capacity the forward was recorded with")

        def tb(name, cols, dtype=BF16):
            return (self._buf(name, (S.Mt_cap, cols), dtype) if cols else self._buf(name, (S.Mt_cap,), dtype))[:Mt]
        S.Mt, S.enc2d = Mt, enc.view(Mt, -1)
        S.zc1, S.c1, S.c2, S.encn = tb("cap_z1", D), tb("cap_c1", D), tb("cap_c2", D), tb("cap_n", D)
        S.enc_rstd = tb("cap_rstd", 0, f32)
        S.kv2 = [tb(f"b{i}.kv2", 2 * D) for i in range(self.cfg.num_layers)]
        for A, kv in zip(S.blocks, S.kv2):
            A.kv2 = kv

    def backward_device(self, dpred):
        S = self._saved
        work = S.kv_work
        if work is not None:
            self.plan_dynamic["n_work"] = int(work.shape[0])
        if S.kv_off is not None:
            self.plan_dynamic["text_rows"] = self._text_rows          # this batch's, not the recorded one's
        key = (id(S), dpred.data_ptr(), self.accumulate_grads, id(self.grad_ready), None if work is None else work.data_ptr(),
               self._schedule_flags())
        self.planned("bwd", key, lambda: self.backward_impl(dpred))

    # ------------------------------------------------------------------ forward
    def forward_impl(self, latents, enc, timestep, mask=None, key_bias=None, kv_len=None, kv_work=None, kv_off=None):
        """``kv_off`` None: ``enc`` is the reference's padded [B, T, C] batch (train_sana.py:168-180).  Otherwise ``enc`` is the
        PACKED text matrix [Mt, C]: image b's prompt occupies rows [kv_off[b], kv_off[b] + kv_len[b]) (kv_len >= 1), rows past
        the last prompt are zero and there are fewer than 256 of them; ``key_bias`` [B, T] / ``kv_len`` as before.  The
        padding rows the reference carries through the caption projection and every block's K / V projection never reach a
        result -- their keys get the -10000 bias, probability exactly 0, gradient exactly 0 -- so the text side runs on the
        real rows only: same prediction bit for bit, text-side GEMMs over ~Sum(len) instead of B * 512 rows."""
        cfg, P = self.cfg, self.P
        D, Hc, H1, H2, dh2 = cfg.inner_dim, cfg.ffn_hidden, cfg.num_attention_heads, cfg.num_cross_attention_heads, \
            cfg.cross_attention_head_dim
        B, Cin, h, w = latents.shape
        N, M = h * w, B * h * w
        packed = kv_off is not None
        if packed and (key_bias is None or kv_len is None or enc.dim() != 2 or enc.shape[0] % 256 or not enc.shape[0]):
            raise ValueError("packed text: enc [rows, C] (rows a multiple of 256) with key_bias [B, T], kv_len [B], kv_off [B]")
        T = key_bias.shape[1] if packed else enc.shape[1]
        Mt = enc.shape[0] if packed else B * T
        # text-side buffers are sized once, for the padded layout (rounded up to whole row tiles: the most a packed batch
        # of this bucket can bring, so a replayed plan never meets a longer matrix than its buffers)
        Mt_cap = max(-(-(B * T) // 256) * 256 if packed else B * T, Mt)
        Cout = cfg.out_channels
        dev = self.dev
        latents = latents.to(device=dev, dtype=BF16).contiguous()
        enc2d = enc.to(device=dev, dtype=BF16).contiguous().view(Mt, -1)
        t_f32 = timestep.to(device=dev, dtype=torch.float32).contiguous()
        if key_bias is None:
            # mask -> additive bias in bf16 as the reference does (patched_sana_transformer.py:275-277)
            if mask is None:
                key_bias = torch.zeros(B, T, dtype=torch.float32, device=dev)
                kv_len = torch.full((B,), T, dtype=torch.int32, device=dev)
            else:
                mdev = mask.to(dev)
                key_bias = ((1 - mdev.to(BF16)) * -10000.0).float().contiguous()
                idx = torch.arange(1, T + 1, device=dev, dtype=torch.int32)
                kv_len = (mdev.to(torch.int32) * idx).amax(dim=1).to(torch.int32).contiguous()
        S = SimpleNamespace(B=B, h=h, w=w, N=N, M=M, T=T, Mt=Mt, Mt_cap=Mt_cap, key_bias=key_bias, kv_len=kv_len,
                            kv_work=kv_work, kv_off=kv_off, enc2d=enc2d, blocks=[])
        buf = self._buf

        def tbuf(name, cols, dtype=BF16):
            """text-side activation [Mt, cols]: the first rows of a buffer sized for the padded layout, so a batch with more
            text rows finds it at the same address (recorded launch plans hold addresses)"""
            return buf(name, (Mt_cap, cols), dtype)[:Mt] if cols else buf(name, (Mt_cap,), dtype)[:Mt]
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None
        pev, self.param_events = self.param_events, None      # per-bucket events of an AdamW update still in flight
        # The policy word of every GEMM launch below carries "how many GEMM streams share the chip" (ops.gemm_concurrency:
        # host state of yat_amd/ops.py).  The embedders and the text branch are launched before the chains set their own
        # value: without this line they inherited whatever the previous call left -- 1 in the very first step, the backward's
        # value in every later one -- so step 0 picked other tiles / split-K for three small GEMMs than steps 1.. (another fp32
        # summation order: 1-ulp differences in the modulation tables; found by the run-to-run determinism test of round 6,
        # tests/test_fulldepth_gpu.py).  Set to what every step but the first has always seen.
        ops.gemm_concurrency(2 if self.side_wgrad else 1)
        ad = self.adapters
        if ad is not None:
            ad.materialize(self.training)                     # delta_w of every target for this step (yat_amd/lokr.py)

        def lin(x_, w_, bias_=None, out=None, **ep):
            """Linear of a (possibly adapted) target: with adapters, x delta_w^T is computed first and folded into the
            base GEMM right after its bias rounding (peft: base_layer(x) + F.linear(x, delta_w))."""
            return adapted_linear(ad, x_, w_, bias_, out=out, **ep)

        def params_ready(bucket, stream=main):
            if pev is not None:
                self._ev_wait(stream, pev[bucket])

        # The text branch (caption projection, RMSNorm, every block's K/V projection) does not depend on the latent
        # stream until the first cross-attention: it runs on the second stream, filling CUs the single-round GEMMs of
        # the main chain leave idle.
        text_scope = (lambda: ops.text_rows(Mt)) if packed else contextlib.nullcontext     # see ops.text_rows

        def text_branch():
            with text_scope():
                text_branch_()

        def text_branch_():
            S.zc1 = tbuf("cap_z1", D)
            S.c1 = lin(enc2d, P["caption_projection.linear_1.weight"], P["caption_projection.linear_1.bias"],
                                  out=tbuf("cap_c1", D), activation="gelu_tanh", aux_out=S.zc1)
            S.c2 = lin(S.c1, P["caption_projection.linear_2.weight"], P["caption_projection.linear_2.bias"],
                                  out=tbuf("cap_c2", D))
            S.encn, S.enc_rstd = ops.rmsnorm_fwd(S.c2, P["caption_norm.weight"], 1e-5, tbuf("cap_n", D),
                                                 tbuf("cap_rstd", 0, torch.float32))
            S.kv2, S.kv_ready = [], []
            cur = torch.cuda.current_stream()
            for i in range(cfg.num_layers):
                pre = f"transformer_blocks.{i}."
                params_ready(i + 1, cur)
                wkv, _ = self._fused(pre + "attn2.to_k.weight", 2 * D, D)
                bkv, _ = self._fused(pre + "attn2.to_k.bias", 2 * D)
                S.kv2.append(lin(S.encn, wkv, bkv, out=tbuf(f"b{i}.kv2", 2 * D)))
                if side is not None:
                    S.kv_ready.append(self._ev_record(cur))

        params_ready(0)
        if side is not None:
            self._wait_stream(side, main)
            with torch.cuda.stream(side):
                text_branch()
        else:
            text_branch()

        # 1. patch embed (1x1 conv == Linear over channels) on token-major rows
        S.x_tok = ops.transpose(latents.view(B, Cin, N), buf("x_tok", (B, N, Cin))).view(M, Cin)
        x = lin(S.x_tok, P["patch_embed.proj.weight"].view(D, Cin), P["patch_embed.proj.bias"],
                           out=buf("x0", (M, D)))
        # 2. timestep embedding (AdaLayerNormSingle)
        S.tproj = ops.timestep_embed(t_f32, 256, buf("tproj", (B, 256)))
        S.z1 = buf("te_z1", (B, D))
        S.e1 = lin(S.tproj, P["time_embed.emb.timestep_embedder.linear_1.weight"],
                              P["time_embed.emb.timestep_embedder.linear_1.bias"], out=buf("te_e1", (B, D)),
                              activation="silu", aux_out=S.z1)
        S.embedded = lin(S.e1, P["time_embed.emb.timestep_embedder.linear_2.weight"],
                                    P["time_embed.emb.timestep_embedder.linear_2.bias"], out=buf("te_emb", (B, D)))
        S.se = ops.act_fwd(S.embedded, "silu", buf("te_se", (B, D)))
        S.tmod = ops.linear_fwd(S.se, P["time_embed.linear.weight"], P["time_embed.linear.bias"],
                                out=buf("te_tmod", (B, 6 * D)))
        # 3. caption projection + RMSNorm + K/V projections: text_branch() above
        scale2 = 1.0 / math.sqrt(dh2)
        # 4. transformer blocks + 5. output head.  Activations live in whole-batch buffers (the backward runs on the whole
        # batch); the forward walks them as `fwd_chains` independent chains over disjoint image ranges, each on its own
        # stream: while one chain sits in a memory-bound kernel (norms, attention, depthwise conv) the other one's
        # GEMM has the matrix cores, and a chain's single-round GEMMs no longer leave the other CUs idle.
        f32 = torch.float32
        for i in range(cfg.num_layers):
            A = SimpleNamespace()
            A.mod = buf(f"b{i}.mod", (B, 6, D))
            A.h1, A.mean1, A.rstd1 = buf(f"b{i}.h1", (M, D)), buf(f"b{i}.mean1", (M,), f32), buf(f"b{i}.rstd1", (M,), f32)
            A.qkv = buf(f"b{i}.qkv", (M, 3 * D))
            A.la_state = buf(f"b{i}.la_state", (B * H1 * 33 * 32,), f32)                  # kept for the backward
            A.softmax1 = i in cfg.modified_blocks             # attn1 as softmax attention (patch_sana_attention_layers.py:125-131)
            A.lse1 = buf(f"b{i}.lse1", (B, H1, N), f32) if A.softmax1 else None
            A.attn, A.lin1, A.x1, A.q2 = (buf(f"b{i}.{n}", (M, D)) for n in ("attn", "lin1", "x1", "q2"))
            A.kv2 = S.kv2[i]
            A.o2, A.lse, A.x2 = buf(f"b{i}.o2", (M, D)), buf(f"b{i}.lse", (B, H2, N), f32), buf(f"b{i}.x2", (M, D))
            A.h2, A.mean2, A.rstd2 = buf(f"b{i}.h2", (M, D)), buf(f"b{i}.mean2", (M,), f32), buf(f"b{i}.rstd2", (M,), f32)
            A.z, A.s = buf(f"b{i}.z", (M, 2 * Hc)), buf(f"b{i}.s", (M, 2 * Hc))          # z: pre-activation, for SiLU'
            A.y, A.lin3, A.x3 = buf(f"b{i}.y", (M, Hc)), buf(f"b{i}.lin3", (M, D)), buf(f"b{i}.x3", (M, D))
            A.u = buf(f"b{i}.u", (M, 2 * Hc)) if (self.keep_glu_u and ad is None) else None      # depthwise-conv output, for the backward
            A.x_in = x if i == 0 else S.blocks[i - 1].x3
            S.blocks.append(A)
        S.x_last = S.blocks[-1].x3 if cfg.num_layers else x
        S.modf = buf("modf", (B, 2, D))
        S.hf, S.meanf, S.rstdf = buf("hf", (M, D)), buf("meanf", (M,), f32), buf("rstdf", (M,), f32)
        out_tok = buf("out_tok", (M, Cout))
        pred = buf("pred", (B, Cout, N))                 # (arena: the caller consumes it before the next forward)
        la_per_image = H1 * 33 * 32

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
                ops.ln_modulate_fwd(xin, mod2d[:, 0:D], mod2d[:, D:2 * D], 6 * D, N, cfg.norm_eps, A.h1[rs], A.mean1[rs],
                                    A.rstd1[rs])
                wqkv, _ = self._fused(pre + "attn1.to_q.weight", 3 * D, D)
                lin(A.h1[rs], wqkv, out=A.qkv[rs])
                if A.softmax1:
                    # AttnProcessor2_0 on attn1: plain softmax attention over the N tokens, 70 heads x 32, no mask --
                    # the cross-attention kernel with q, k, v = the three column blocks of the fused projection
                    qkv_ = A.qkv[rs]
                    ops.sdpa_fwd(qkv_[:, :D], qkv_[:, D:2 * D], qkv_[:, 2 * D:], nb, N, N, H1, D // H1, 1.0 / math.sqrt(D // H1),
                                 None, None, A.attn[rs], A.lse1[bs])
                else:
                    ops.linear_attn_fwd(A.qkv[rs], nb, N, H1, D, 2 * D, A.attn[rs],
                                        A.la_state[b0 * la_per_image:b1 * la_per_image])
                lin(A.attn[rs], P[pre + "attn1.to_out.0.weight"], P[pre + "attn1.to_out.0.bias"], out=A.x1[rs],
                               aux_out=A.lin1[rs], gate=mod2d[:, 2 * D:3 * D], ld_gate=6 * D, residual=xin,
                               rows_per_batch=N)
                lin(A.x1[rs], P[pre + "attn2.to_q.weight"], P[pre + "attn2.to_q.bias"], out=A.q2[rs])
                if side is not None:
                    self._ev_wait(stream, S.kv_ready[i])
                kv = A.kv2 if packed else A.kv2[ts]          # packed: the whole matrix + this chain's row offsets
                with text_scope():
                    ops.sdpa_fwd(A.q2[rs], kv[:, :D], kv[:, D:], nb, N, T, H2, dh2, scale2, key_bias[bs], kv_len[bs],
                                 A.o2[rs], A.lse[bs], kv_off=kv_off[bs] if packed else None)
                lin(A.o2[rs], P[pre + "attn2.to_out.0.weight"], P[pre + "attn2.to_out.0.bias"], out=A.x2[rs],
                               residual=A.x1[rs])
                ops.ln_modulate_fwd(A.x2[rs], mod2d[:, 3 * D:4 * D], mod2d[:, 4 * D:5 * D], 6 * D, N, cfg.norm_eps, A.h2[rs],
                                    A.mean2[rs], A.rstd2[rs])
                lin(A.h2[rs], P[pre + "ff.conv_inverted.weight"].view(2 * Hc, D), P[pre + "ff.conv_inverted.bias"],
                               out=A.s[rs], activation="silu", aux_out=A.z[rs])
                ops.dwconv_glu_fwd(A.s[rs], nb, h, w, Hc, P[pre + "ff.conv_depth.weight"].view(2 * Hc, 9),
                                   P[pre + "ff.conv_depth.bias"], A.y[rs], u_out=None if A.u is None else A.u[rs])
                lin(A.y[rs], P[pre + "ff.conv_point.weight"].view(D, Hc), None, out=A.x3[rs], aux_out=A.lin3[rs],
                               gate=mod2d[:, 5 * D:6 * D], ld_gate=6 * D, residual=A.x2[rs], rows_per_batch=N)
            # output head: modulated norm + proj_out + unpatchify
            ops.modulation_fwd(P["scale_shift_table"], S.embedded[bs], 0, S.modf[bs])
            modf2d = S.modf.view(B, 2 * D)[bs]
            ops.ln_modulate_fwd(S.x_last[rs], modf2d[:, 0:D], modf2d[:, D:2 * D], 2 * D, N, 1e-6, S.hf[rs], S.meanf[rs],
                                S.rstdf[rs])
            ops.linear_fwd(S.hf[rs], P["proj_out.weight"], P["proj_out.bias"], out=out_tok[rs])
            ops.transpose(out_tok[rs].view(nb, N, Cout), pred[bs])

        # (with adapters: one chain, so that the T1 product of every target covers the whole batch and the backward reuses it)
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
        return pred.view(B, Cout, h, w)

    # ------------------------------------------------------------------ backward
    def backward_impl(self, dpred):
        S = self._saved
        if S is None:
            raise RuntimeError("backward_impl called without a saved forward")
        ops.gemm_concurrency(2 if self.side_wgrad else 1)     # dgrad chain beside the weight gradients' stream
        cfg, P, G = self.cfg, self.P, self.G
        D, Hc, H1, H2, dh2 = cfg.inner_dim, cfg.ffn_hidden, cfg.num_attention_heads, cfg.num_cross_attention_heads, \
            cfg.cross_attention_head_dim
        B, h, w, N, M, T, Mt = S.B, S.h, S.w, S.N, S.M, S.T, S.Mt
        Cout, Cin = cfg.out_channels, cfg.in_channels
        acc = self.accumulate_grads
        buf = self._buf
        packed, Mt_cap = S.kv_off is not None, S.Mt_cap

        def tbuf(name, cols, dtype=BF16):
            return buf(name, (Mt_cap, cols), dtype)[:Mt]

        text_scope = (lambda: ops.text_rows(Mt)) if packed else contextlib.nullcontext     # see ops.text_rows
        f32, u8 = torch.float32, torch.uint8
        ws_col = buf("ws_col", (int(ops._lib().yat_colsum_workspace_bytes(max(M, Mt_cap), max(2 * Hc, 6 * D, 3 * D))),), u8)
        ws_ln = buf("ws_ln", (ops.ln_bwd_workspace_bytes(M, D, N),), u8)
        ws_gate = buf("ws_gate", (int(ops._lib().yat_gate_bwd_workspace_bytes(M, D, N)),), u8)
        ws_dw = buf("ws_dw", (ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc),), u8)
        la_ws = buf("la_ws", (ops.linear_attn_workspace_bytes(B, N, H1),), u8)
        scale2 = 1.0 / math.sqrt(dh2)
        ad = self.adapters

        def dgrad(dy_, w_, out=None, residual=None):
            """Input gradient through a (possibly adapted) target: dy W, plus dy delta_w accumulated in place."""
            r_ = ops.linear_dgrad(dy_, w_, out=out, residual=residual)
            if ad is not None:
                flush_adapter_wgrads(dy_, ad.dgrad_term(dy_, w_, r_))
            return r_

        # With adapters the weight gradient of a target needs H = dy P, which the input gradient of the same dy computes
        # anyway: emit() only queues (dy, x, dW); the dgrad() that follows launches the queued item with its H.
        pending_ad = []

        def flush_adapter_wgrads(dy_=None, hs=None):
            keep = []
            for item in pending_ad:
                if dy_ is None or item[0].data_ptr() == dy_.data_ptr():
                    run_off_chain(lambda item=item, hs=(hs if dy_ is not None else None):
                                  ad.wgrad(*item, accumulate=acc, hs=hs))
                else:
                    keep.append(item)
            pending_ad[:] = keep

        # Weight/bias gradients are off the critical path (nothing in backward reads them): they go to a second
        # stream so their blocks fill the CUs the single-round dgrad launches leave idle, and their prologue/epilogue
        # phases overlap the other stream's MFMA phases.  Buffers they read are either written once per backward (head,
        # embedders) or alternate between two sets by block parity (transformer blocks, below).
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.side_wgrad else None

        def run_off_chain(fn):
            if side is None:
                fn()
                return
            self._wait_stream(side, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()

        def wgrad(dy, x, key, shape2d, bias_key=None):
            def run():
                if ad is not None:        # frozen base: only the adapters' share (nothing at all for a non-target)
                    ad.wgrad(dy, x, G[key].view(shape2d), accumulate=acc)
                    return
                ops.linear_wgrad(dy, x, G[key].view(shape2d), accumulate=acc,
                                 bias_grad=None if bias_key is None else G[bias_key], colsum_ws=ws_col)   # bias gradient: same launch
            if side is None:
                run()
                return
            self._wait_stream(side, main)
            with torch.cuda.stream(side):
                run()

        # ---- output head
        if dpred.dtype != BF16 or not dpred.is_contiguous():
            dpred = dpred.to(BF16).contiguous()
        d_out_tok = ops.transpose(dpred.view(B, Cout, N), buf("d_out_tok", (B, N, Cout))).view(M, Cout)
        wgrad(d_out_tok, S.hf, "proj_out.weight", (Cout, D), "proj_out.bias")
        dhf = ops.linear_dgrad(d_out_tok, P["proj_out.weight"], out=buf("dh", (M, D)))
        dmodf = ops.zero_(buf("dmodf", (B, 2, D), f32))
        dtmod = ops.zero_(buf("dtmod", (B, 6 * D), f32))
        demb = ops.zero_(buf("demb", (B, D), f32))
        dxa, dxb = buf("dx_a", (M, D)), buf("dx_b", (M, D))
        dmodf2d = dmodf.view(B, 2 * D)
        dx = ops.ln_modulate_bwd(S.x_last, S.meanf, S.rstdf, S.modf.view(B, 2 * D)[:, D:2 * D], 2 * D, N, dhf, None, dxa,
                                 dmodf2d[:, 0:D], dmodf2d[:, D:2 * D], 2 * D, ws_ln)
        ops.modulation_bwd(dmodf, G["scale_shift_table"], demb, 0, accumulate_table=acc)
        denc = tbuf("denc", D)
        # ---- blocks, last to first.  A block's seven weight gradients are deferred to its end and go out as ONE grouped
        # GEMM launch on the side stream (~1240 full-K 256x256 tiles = 4.85 rounds of the 256 CUs, instead of seven
        # launches of 81..396 tiles, three of them split-K); the gradient buffers they read alternate between two sets
        # by block parity, so the main stream is already writing block i-1's while the side stream reads block i's.
        set_done = [None, None]
        for i in reversed(range(cfg.num_layers)):
            pre = f"transformer_blocks.{i}."
            A = S.blocks[i]
            par = i & 1
            if set_done[par] is not None:
                self._ev_wait(main, set_done[par])            # block i+2's weight gradients have read this set
                set_done[par] = None
            mod2d = A.mod.view(B, 6 * D)
            dmod = ops.zero_(buf(f"dmod.{par}", (B, 6, D), f32))
            dmod2d = dmod.view(B, 6 * D)
            deferred = []                                     # (dy, x, dW) of this block

            def off_chain(fn):
                """Work nobody downstream on the dependent chain reads: side stream, right behind its producer."""
                if side is None:
                    fn()
                    return
                self._wait_stream(side, main)
                with torch.cuda.stream(side):
                    fn()

            small = []                                        # the three D x D weight gradients: one grouped launch

            def emit(dy_, x_, gw_, bias=None, group=False, text=False):
                """``text``: the reduction runs over the (packed) text rows -- never grouped, its K is the plan's dynamic
                row count"""
                if ad is not None:        # frozen base: adapter gradients only, launched by the dgrad() of the same dy
                    pending_ad.append((dy_, x_, gw_))
                    return
                if group and self.group_small_wgrad:      # (also without a side stream: one launch instead of three
                    small.append((dy_, x_, gw_, bias))            #  split-K ones -- the serialized pass runs the step's kernels)
                    return
                if side is None:
                    deferred.append((dy_, x_, gw_, bias, text))
                    return

                def run():
                    with (text_scope() if text else contextlib.nullcontext()):
                        ops.linear_wgrad(dy_, x_, gw_, accumulate=acc, bias_grad=bias, colsum_ws=ws_col)
                off_chain(run)

            # x3 = x2 + gate_mlp * lin3
            dlin3 = buf(f"dlin3.{par}", (M, D))
            ops.gate_bwd(dx, A.lin3, mod2d[:, 5 * D:6 * D], 6 * D, N, dlin3, dmod2d[:, 5 * D:6 * D], 6 * D, ws_gate)
            emit(dlin3, A.y, G[pre + "ff.conv_point.weight"].view(D, Hc))
            dz = buf(f"dz.{par}", (M, 2 * Hc))
            if A.u is not None:
                # the GLU backward runs in the epilogue of the GEMM that produces dy (dy itself never reaches memory);
                # the depthwise kernel's first backward pass (recompute u, 156 us) disappears
                du_, dy = ops.linear_dgrad_glu(dlin3, P[pre + "ff.conv_point.weight"].view(D, Hc), A.u,
                                               buf("du", (M, 2 * Hc))), None
            else:
                du_, dy = None, dgrad(dlin3, P[pre + "ff.conv_point.weight"].view(D, Hc), out=buf("dy", (M, Hc)))
            ops.dwconv_glu_bwd(A.s, A.z, B, h, w, Hc, P[pre + "ff.conv_depth.weight"].view(2 * Hc, 9),
                               P[pre + "ff.conv_depth.bias"], dy, dz, G[pre + "ff.conv_depth.weight"].view(2 * Hc, 9),
                               G[pre + "ff.conv_depth.bias"], ws_dw, accumulate=acc,
                               dz_colsum=G[pre + "ff.conv_inverted.bias"], du=du_)        # bias gradient in the same pass
            emit(dz, A.h2, G[pre + "ff.conv_inverted.weight"].view(2 * Hc, D))
            dh2_ = dgrad(dz, P[pre + "ff.conv_inverted.weight"].view(2 * Hc, D), out=buf(f"dh2.{par}", (M, D)))
            other = dxb if dx is dxa else dxa
            # LayerNorm backward: dx on the chain; the shift/scale gradients (column statistics) feed only the
            # modulation tables and go to the side stream
            ln2 = (A.x2, A.mean2, A.rstd2, mod2d[:, 4 * D:5 * D], 6 * D, N, dh2_)
            dx2 = ops.ln_modulate_bwd(*ln2, dx, buf(f"dx2.{par}", (M, D)), dmod2d[:, 3 * D:4 * D], dmod2d[:, 4 * D:5 * D],
                                      6 * D, ws_ln, parts=1 if side is not None else 3)
            if side is not None:
                off_chain(lambda ln2=ln2, dmod2d=dmod2d: ops.ln_modulate_bwd(
                    *ln2, None, None, dmod2d[:, 3 * D:4 * D], dmod2d[:, 4 * D:5 * D], 6 * D, ws_ln, parts=2))
            # x2 = x1 + to_out(o2)
            emit(dx2, A.o2, G[pre + "attn2.to_out.0.weight"], G[pre + "attn2.to_out.0.bias"], group=True)
            do2 = dgrad(dx2, P[pre + "attn2.to_out.0.weight"], out=buf(f"do2.{par}", (M, D)))
            dq2 = buf(f"dq2.{par}", (M, D))
            dkv2 = tbuf(f"dkv2.{par}", 2 * D)
            delta = buf(f"delta.{par}", (B, H2, N), f32)
            sd = (A.q2, A.kv2[:, :D], A.kv2[:, D:], B, N, T, H2, dh2, scale2, S.key_bias, S.kv_len, A.o2, do2, A.lse, delta,
                  dq2, dkv2[:, :D], dkv2[:, D:])

            def dkv_part(parts, sd=sd, dkv2=dkv2):
                with text_scope():
                    if packed:
                        # the zero rows behind the last prompt (< 256 of them) belong to no image: dK/dV leaves them alone,
                        # the text-side GEMMs read them -- zero the last 256 rows first (the real ones among them are rewritten)
                        ops.zero_last_rows(dkv2, 256)
                    ops.sdpa_bwd(*sd, work=S.kv_work, parts=parts, kv_off=S.kv_off)
            # cross-attention backward: dQ (+delta) on the chain, dK/dV -- read only by the text-side gradients -- behind it
            if side is not None:
                with text_scope():
                    ops.sdpa_bwd(*sd, work=S.kv_work, parts=1, kv_off=S.kv_off)
                off_chain(lambda dkv_part=dkv_part: dkv_part(2))
            else:
                dkv_part(3)
            emit(dq2, A.x1, G[pre + "attn2.to_q.weight"], G[pre + "attn2.to_q.bias"], group=True)
            dx1 = dgrad(dq2, P[pre + "attn2.to_q.weight"], out=other, residual=dx2)    # dx1 = dx2 + dq2 Wq
            wkv, gkv = self._fused(pre + "attn2.to_k.weight", 2 * D, D)
            _, gbkv = self._fused(pre + "attn2.to_k.bias", 2 * D)
            emit(dkv2, S.encn, gkv, gbkv, text=packed)
            # x1 = x + gate_msa * lin1
            dlin1 = buf(f"dlin1.{par}", (M, D))
            ops.gate_bwd(dx1, A.lin1, mod2d[:, 2 * D:3 * D], 6 * D, N, dlin1, dmod2d[:, 2 * D:3 * D], 6 * D, ws_gate,
                         dbias=G[pre + "attn1.to_out.0.bias"], accumulate_bias=acc)       # bias gradient in the same pass
            emit(dlin1, A.attn, G[pre + "attn1.to_out.0.weight"], group=True)
            if small:
                # 2240 x 2240 x 8192 each: 81 tiles of 256 x 256 -- alone they need split-K (fp32 slabs + a reduce launch);
                # together 243 full-K tiles fill the 256 CUs in one round
                def small_grads(small=small):
                    ops.wgrad_grouped([(a, b, c, bias_) if bias_ is not None else (a, b, c)
                                       for a, b, c, bias_ in small], accumulate=acc)          # bias gradients: same launch
                off_chain(small_grads)
            dattn = dgrad(dlin1, P[pre + "attn1.to_out.0.weight"], out=buf("dh", (M, D)))
            dqkv = buf(f"dqkv.{par}", (M, 3 * D))
            if A.softmax1:
                ops.sdpa_bwd(A.qkv[:, :D], A.qkv[:, D:2 * D], A.qkv[:, 2 * D:], B, N, N, H1, D // H1, 1.0 / math.sqrt(D // H1),
                             None, None, A.attn, dattn, A.lse1, buf("delta1", (B, H1, N), f32), dqkv[:, :D],
                             dqkv[:, D:2 * D], dqkv[:, 2 * D:])
            else:
                ops.linear_attn_bwd(A.qkv, B, N, H1, D, 2 * D, dattn, dqkv, la_ws, state=A.la_state)
            wqkv, gqkv = self._fused(pre + "attn1.to_q.weight", 3 * D, D)
            emit(dqkv, A.h1, gqkv)
            dh1 = dgrad(dqkv, wqkv, out=buf(f"dh1.{par}", (M, D)))
            ln1 = (A.x_in, A.mean1, A.rstd1, mod2d[:, D:2 * D], 6 * D, N, dh1)
            dx = ops.ln_modulate_bwd(*ln1, dx1, dx, dmod2d[:, 0:D], dmod2d[:, D:2 * D], 6 * D, ws_ln,
                                     parts=1 if side is not None else 3)

            def table_grads(ln1=ln1, dmod=dmod, dmod2d=dmod2d, pre=pre):
                if side is not None:
                    ops.ln_modulate_bwd(*ln1, None, None, dmod2d[:, 0:D], dmod2d[:, D:2 * D], 6 * D, ws_ln, parts=2)
                ops.modulation_bwd(dmod, G[pre + "scale_shift_table"], dtmod, D, accumulate_table=acc)
            off_chain(table_grads)

            def block_grads(deferred=deferred, dx2=dx2, dq2=dq2, dkv2=dkv2, wkv=wkv, gbkv=gbkv, pre=pre,
                            first=(i == cfg.num_layers - 1)):
                if deferred:                                      # (one-stream schedule: the weight gradients at the block's end)
                    for dy_, x_, gw_, _, text_ in deferred:
                        with (text_scope() if text_ else contextlib.nullcontext()):
                            ops.linear_wgrad(dy_, x_, gw_, accumulate=acc)
                    for dy_, _, _, bias_, text_ in deferred:      # attn2.to_out / to_q (unless grouped above) / to_k|to_v
                        if bias_ is not None:
                            with (text_scope() if text_ else contextlib.nullcontext()):
                                ops.colsum(dy_, bias_, ws_col, accumulate=acc)
                # the text-side gradient chain (denc += dkv2 Wkv) only meets the main chain at the caption branch
                with text_scope():
                    dgrad(dkv2, wkv, out=denc, residual=None if first else denc)

            if ad is not None:
                flush_adapter_wgrads()        # anything queued whose dy had no dgrad() (none today; keeps the queue per block)
            if side is None:
                block_grads()
                if self.grad_ready is not None:
                    self._callback(self.grad_ready, i + 1)
            else:
                self._wait_stream(side, main)
                with torch.cuda.stream(side):
                    block_grads()
                    # bucket i+1 is complete once the side stream gets here (it has waited for the main stream's
                    # share: the fused bias / depthwise gradients); the DDP hook records its event on the CURRENT
                    # stream, so the all-reduce follows the side stream and the dependent chain never waits for it
                    if self.grad_ready is not None:
                        self._callback(self.grad_ready, i + 1)
                    set_done[par] = self._ev_record(side)
        # ---- embedders (small: back on the main stream)
        if side is not None:
            self._wait_stream(main, side)
            side = None
        wgrad(dx, S.x_tok, "patch_embed.proj.weight", (D, Cin), "patch_embed.proj.bias")
        # caption branch (every call in it runs over the text rows)
        dc2 = tbuf("dc2", D)
        ws_rms = buf("ws_rms", (int(ops._lib().yat_rmsnorm_bwd_workspace_bytes(Mt_cap, D)),), u8)
        with text_scope():
            ops.rmsnorm_bwd(S.c2, P["caption_norm.weight"], S.enc_rstd, denc, dc2, G["caption_norm.weight"], ws_rms,
                            accumulate_dw=acc)
            wgrad(dc2, S.c1, "caption_projection.linear_2.weight", (D, D), "caption_projection.linear_2.bias")
            dc1 = dgrad(dc2, P["caption_projection.linear_2.weight"], out=denc)
            dzc1 = ops.act_bwd(S.zc1, dc1, "gelu_tanh", dc2)
            wgrad(dzc1, S.enc2d, "caption_projection.linear_1.weight", (D, cfg.caption_channels),
                  "caption_projection.linear_1.bias")
        # timestep branch
        dtmod_b = ops.f32_to_bf16(dtmod, buf("dtmod_b", (B, 6 * D)))
        wgrad(dtmod_b, S.se, "time_embed.linear.weight", (6 * D, D), "time_embed.linear.bias")
        dse = ops.linear_dgrad(dtmod_b, P["time_embed.linear.weight"], out=buf("te_d1", (B, D)))
        demb_a = ops.act_bwd(S.embedded, dse, "silu", buf("te_d2", (B, D)))
        demb_b = ops.f32_to_bf16(demb, buf("te_d3", (B, D)))
        d_emb = ops.add_bf16(demb_a, demb_b, buf("te_d1", (B, D)))
        wgrad(d_emb, S.e1, "time_embed.emb.timestep_embedder.linear_2.weight", (D, D), "time_embed.emb.timestep_embedder.linear_2.bias")
        de1 = dgrad(d_emb, P["time_embed.emb.timestep_embedder.linear_2.weight"], out=buf("te_d2", (B, D)))
        dz1 = ops.act_bwd(S.z1, de1, "silu", buf("te_d3", (B, D)))
        wgrad(dz1, S.tproj, "time_embed.emb.timestep_embedder.linear_1.weight", (D, 256), "time_embed.emb.timestep_embedder.linear_1.bias")
        if self.grad_ready is not None:
            self._callback(self.grad_ready, 0)
        if ad is not None:
            ad.project()                  # d_delta_w (flat gradient slots of the frozen targets) -> adapter gradients

    # ------------------------------------------------------------------ checkpoint I/O (diffusers layout)
    def save_pretrained(self, path):
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.join_pending_update()
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        cfgd = asdict(self.cfg)
        cfgd.update({"_class_name": "SanaTransformer2DModel", "cross_attention_dim": self.cfg.cross_attention_dim,
                     "attention_bias": False, "dropout": 0.0, "norm_elementwise_affine": False,
                     "interpolation_scale": None})
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfgd, f, indent=2)

    @classmethod
    def from_pretrained(cls, path, device="cuda", **_):
        import json
        import os
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        known = {k: raw[k] for k in SanaConfig.__dataclass_fields__ if k in raw}
        model = cls(SanaConfig(**known), device=device)
        model.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return model
