"""SANA flow-matching training recipe on the HIP path (``SanaModel.optimize``, train_sana.py:163-219).

Reference op order, kept: pad text embeddings to 512 + mask (:168-180) -> noise in bf16 from the CPU
generator (:183) -> logit-normal timestep indices (:185-193) -> sigmas (:195-204) ->
noisy = (1-sigma) x + sigma n (:206-207) -> model (:210-215) -> target = n - x (:217) ->
MSE in fp32 (:218).

What changes on MI355X: the Python pad loop and its B small H2D copies become one pinned staging
buffer, ONE H2D copy and one ``yat_pad_mask`` launch; ``get_sigmas``' B device->host syncs disappear
(the index is known on the host); mix/target/loss(+dL/dpred) are two fused launches.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .scheduler import DDPMSchedule, FlowMatchSchedule

BF16 = torch.bfloat16


class _MseLoss(torch.autograd.Function):
    """mean((pred.float() - target.float())**2) with the gradient produced in the same launch."""

    @staticmethod
    def forward(ctx, pred, target, ws):
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        ops.mse_fwd_bwd(pred.contiguous(), target, loss, dpred, ws)
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.to(dpred.dtype), None, None


class _Stager:
    """One host->device copy per batch: everything a step needs from the host (cached latents, the CPU-drawn noise, the ragged
    text embeddings, their offsets, timesteps, sigmas, the attention work list) is packed into ONE pinned buffer and lands in
    one device buffer with a single asynchronous copy on the step's stream (the reference issues B + 5 small pageable copies
    per step, train_sana.py:178-193).  Two pinned buffers alternate: a buffer is rewritten only after the copy that read it
    has completed (an event, long since signalled two steps later)."""

    def __init__(self, dev):
        self.dev, self.pin, self.ev, self.k, self.land = dev, [None, None], [None, None], 0, None

    def begin(self, nbytes, capacity=0):
        """``capacity``: the most this caller will ever stage for the current shapes -- buffers are sized for it at once, so
        their addresses (which recorded launch plans hold) do not move when a longer caption arrives."""
        k = self.k
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        self.cap = max(getattr(self, "cap", 0), capacity, nbytes)
        if self.pin[k] is None or self.pin[k].numel() < self.cap:
            self.pin[k] = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
        return self.pin[k]

    def commit(self, nbytes):
        k = self.k
        if self.land is None or self.land.numel() < self.cap:
            self.land = torch.empty(self.cap, dtype=torch.uint8, device=self.dev)
        self.land[:nbytes].copy_(self.pin[k][:nbytes], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.ev[k] = ev
        self.k ^= 1
        return self.land


def _pack_rows(embeddings, dst):
    """The ragged embeddings, one after the other, into ``dst`` ([sum of rows, C], bf16 -- pinned staging memory): one row-block
    copy each.  (``torch.cat(..., out=dst)`` does the same 20 x slower: 3.7 ms instead of 0.17 ms for eight prompts on the
    container's CPU, 9.5 ms per step in the trainer's host profile.)"""
    o = 0
    for e in embeddings:
        n = e.shape[0]
        dst[o:o + n].copy_(e)
        o += n
    return dst


def _report_loss(model, loss):
    """The device path has no ``accelerator.backward(loss)`` between its loss and its backward: a data-parallel wrapper that
    carries the logged loss with the gradients (yat_amd/ddp.py ``on_loss``) is told here."""
    hook = getattr(model, "loss_ready", None)
    if hook is not None:
        hook(loss)


def _layout(sizes):
    """16-byte aligned offsets of consecutive byte segments -> (offsets, total)."""
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 15) & ~15
    return offs, o


class SanaRecipe:
    def __init__(self, model, scheduler: FlowMatchSchedule | None = None, pad_to: int = 512, device="cuda"):
        self.model = model
        self.scheduler = scheduler or FlowMatchSchedule()
        self.pad_to = pad_to
        self.dev = torch.device(device)
        self._mse_ws = torch.empty(256, dtype=torch.float32, device=self.dev)
        self._pin = None
        self.device_rng = None     # torch.Generator(device) for the throughput mode

    # ---- text embeddings: ragged list -> padded [B, T, C] + mask/bias/kv_len on device (one H2D)
    def pad_embeddings(self, embeddings):
        B, T = len(embeddings), self.pad_to
        C = embeddings[0].shape[1]
        lens = [int(e.shape[0]) for e in embeddings]
        if max(lens) > T:
            raise ValueError(f"embedding longer than pad length {T}")
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        if embeddings[0].is_cuda:
            src = torch.cat([e.to(BF16) for e in embeddings])
        else:
            total = offs[-1]
            if self._pin is None or self._pin.numel() < total * C:
                self._pin = torch.empty(max(total * C, B * T * C), dtype=BF16).pin_memory()
            stage = self._pin[: total * C].view(total, C)
            _pack_rows(embeddings, stage)
            src = stage.to(self.dev, non_blocking=True)
        offsets = torch.tensor(offs, dtype=torch.int32).to(self.dev, non_blocking=True)
        enc = torch.empty(B, T, C, dtype=BF16, device=self.dev)
        mask = torch.empty(B, T, dtype=torch.int64, device=self.dev)
        bias = torch.empty(B, T, dtype=torch.float32, device=self.dev)
        kvl = torch.empty(B, dtype=torch.int32, device=self.dev)
        ops.pad_mask(src, offsets, B, T, C, enc, mask, bias, kvl)
        self.kv_work = ops.kv_work_list(lens, T, self.dev)     # lengths are host data: compact dK/dV work list
        return enc, mask, bias, kvl

    def draw(self, shape, generator):
        """noise (bf16, drawn first), then timestep indices -- the reference's draw order.
        ``generator`` None/CPU -> host draw exactly like the reference; a device generator draws on the GPU."""
        B = shape[0]
        if generator is not None and generator.device.type == "cuda":
            noise = torch.randn(shape, generator=generator, device=self.dev, dtype=BF16)
            cpu_gen = getattr(self, "_cpu_gen", None)
            idx, t, sig = self.scheduler.sample(B, cpu_gen)
        else:
            noise = torch.randn(shape, generator=generator, device="cpu", dtype=BF16).to(self.dev, non_blocking=True)
            idx, t, sig = self.scheduler.sample(B, generator)
        return noise, t.to(self.dev, non_blocking=True), sig.to(self.dev, non_blocking=True)

    def optimize(self, latents, embeddings, generator=None, return_pred=False):
        """-> loss (0-dim fp32 tensor attached to the HIP autograd node)."""
        enc, mask, bias, kvl = self.pad_embeddings(embeddings)
        latents = latents.to(device=self.dev, dtype=BF16).contiguous()
        noise, timesteps, sigmas = self.draw(latents.shape, generator)
        noisy, target = ops.flow_mix(latents, noise, sigmas)
        self.model.next_kv_work = self.kv_work
        pred = self.model(noisy, encoder_hidden_states=enc, timestep=timesteps, encoder_attention_mask=mask).sample
        loss = _MseLoss.apply(pred, target, self._mse_ws)
        return (loss, pred, target) if return_pred else loss

    # ---- packed text rows (no padding rows through the text-side GEMMs; SanaTransformer2DModelHIP.forward_impl)
    TEXT_ROW_PAD = 256          # the packed matrix ends in fewer than this many zero rows (one GEMM tile of rows)

    def packs_text(self, lens):
        """May this batch take the packed text layout?  The model must accept it and every prompt needs a row (a prompt of
        length 0 attends uniformly to the padding rows in the reference -- only the padded layout has them).  Adapters ride
        along: their products follow the row count of whatever activation they are handed.  ``YAT_TEXT_PACK=0`` switches it
        off."""
        if os.environ.get("YAT_TEXT_PACK", "1") == "0" or not getattr(self.model, "packed_text", False):
            return False
        return min(lens) >= 1

    def packed_rows(self, rows):
        return -(-rows // self.TEXT_ROW_PAD) * self.TEXT_ROW_PAD

    def packed_enc(self, B, T, C, rows):
        """[packed_rows(rows), C] view of one persistent buffer sized for B * T rows (its address never moves)."""
        cap = self.packed_rows(B * T)
        buf = getattr(self, "_packed_buf", None)
        if buf is None or buf.shape != (cap, C):
            buf = self._packed_buf = torch.empty(cap, C, dtype=BF16, device=self.dev)
        return buf[:self.packed_rows(rows)]

    def train_step_device(self, latents, enc, mask_bias_kvl, noise, timesteps, sigmas, loss_out, kv_work=None, gscale=1.0,
                          kv_off=None):
        """Graph-friendly straight-line step on device-resident inputs: forward, loss+dL/dpred, backward.
        Used by bench.py and the trainer fast path (no autograd objects, no allocation besides pred).
        ``kv_off``: ``enc`` is the packed text matrix (``ops.pack_mask``), see ``forward_impl``."""
        bias, kvl = mask_bias_kvl
        noisy, target = ops.flow_mix(latents, noise, sigmas, self._noisy(latents), self._target(latents))
        dev_path = hasattr(self.model, "forward_device")
        packed = {} if kv_off is None else {"kv_off": kv_off}
        if dev_path:
            pred = self.model.forward_device(noisy, enc, timesteps, bias, kvl, kv_work=kv_work, **packed)
        else:
            pred = self.model.forward_impl(noisy, enc, timesteps, None, key_bias=bias, kv_len=kvl, kv_work=kv_work, **packed)
        dpred = self._dpred(pred)
        ops.mse_fwd_bwd(pred, target, loss_out, dpred, self._mse_ws, gscale=gscale)
        _report_loss(self.model, loss_out)
        if dev_path:
            self.model.backward_device(dpred)
        else:
            self.model.backward_impl(dpred)
        return loss_out

    def optimize_device(self, latents, embeddings, generator=None, gscale=1.0):
        """The trainer's step (train_sana.py:163-219 + the backward of common/trainer.py:344) on the allocation-free path:
        same draws in the same order as ``optimize`` (noise first, then the timestep indices, both from ``generator`` on the
        CPU as the reference draws them), but the host side is one packed pinned buffer and ONE H2D copy, and forward, loss,
        dL/dpred (scaled by ``gscale`` = 1 / gradient_accumulation_steps) and backward are straight-line C-ABI launches with
        no autograd objects.  -> loss (0-dim fp32 device tensor); the gradients are already in the flat gradient buffer."""
        if latents.is_cuda or embeddings[0].is_cuda:
            raise ValueError("optimize_device stages host batches (the sampler yields CPU tensors)")
        B, T = len(embeddings), self.pad_to
        C = embeddings[0].shape[1]
        lens = [int(e.shape[0]) for e in embeddings]
        if max(lens) > T:
            raise ValueError(f"embedding longer than pad length {T}")
        rows = sum(lens)
        pairs = [(b, t) for b, L in enumerate(lens) for t in range((L + 63) // 64)]          # dK/dV work list (ops.kv_work_list)
        nlat = latents.numel()
        max_pairs = B * ((T + 63) // 64)
        # fixed-size segments first (their device addresses then depend on the bucket shape only: a recorded launch plan
        # stays valid from batch to batch); the ragged text rows go last
        (o_lat, o_noise, o_off, o_t, o_sig, o_work, o_emb), total = _layout(
            [2 * nlat, 2 * nlat, 4 * (B + 1), 4 * B, 2 * B, 8 * max_pairs, 2 * rows * C])
        st = self._stager = getattr(self, "_stager", None) or _Stager(self.dev)
        pin = st.begin(total, capacity=total + 2 * (B * T - rows) * C)

        def seg(o, n, dtype):
            return pin[o:o + n].view(dtype)
        seg(o_lat, 2 * nlat, BF16).view(latents.shape).copy_(latents)
        t, sig = self._draw_cached(latents.shape, B, generator, seg(o_noise, 2 * nlat, BF16).view(latents.shape))      # :183-204
        _pack_rows(embeddings, seg(o_emb, 2 * rows * C, BF16).view(rows, C))
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        seg(o_off, 4 * (B + 1), torch.int32).copy_(torch.tensor(offs, dtype=torch.int32))
        seg(o_t, 4 * B, torch.float32).copy_(t)
        seg(o_sig, 2 * B, BF16).copy_(sig)
        seg(o_work, 8 * len(pairs), torch.int32).copy_(torch.tensor(pairs, dtype=torch.int32).flatten())
        land = st.commit(total)

        def dseg(o, n, dtype):
            return land[o:o + n].view(dtype)
        lat_d = dseg(o_lat, 2 * nlat, BF16).view(latents.shape)
        noise_d = dseg(o_noise, 2 * nlat, BF16).view(latents.shape)
        fixed = getattr(self, "_fixed", None)
        if fixed is None or fixed[0].shape != (B, T, C):
            fixed = self._fixed = (torch.empty(B, T, C, dtype=BF16, device=self.dev),
                                   torch.empty(B, T, dtype=torch.int64, device=self.dev),
                                   torch.empty(B, T, dtype=torch.float32, device=self.dev),
                                   torch.empty(B, dtype=torch.int32, device=self.dev),
                                   torch.zeros(1, dtype=torch.float32, device=self.dev))
        enc, mask, bias, kvl, loss_out = fixed
        src_d, off_d, kv_off = dseg(o_emb, 2 * rows * C, BF16).view(rows, C), dseg(o_off, 4 * (B + 1), torch.int32), None
        if self.packs_text(lens):
            enc, kv_off = self.packed_enc(B, T, C, rows), off_d[:B]
            ops.pack_mask(src_d, off_d, B, T, C, enc, mask, bias, kvl)                                                   # :168-180,
        else:                                                                                    # minus the padding rows
            ops.pad_mask(src_d, off_d, B, T, C, enc, mask, bias, kvl)                                                    # :168-180
        self.train_step_device(lat_d, enc, (bias, kvl), noise_d, dseg(o_t, 4 * B, torch.float32), dseg(o_sig, 2 * B, BF16),
                               loss_out, kv_work=dseg(o_work, 8 * len(pairs), torch.int32).view(len(pairs), 2), gscale=gscale,
                               kv_off=kv_off)
        return loss_out[0].clone()

    def _draw_cached(self, shape, B, generator, noise_out):
        """The step's host draws -- bf16 noise of ``shape`` into ``noise_out``, then the logit-normal timestep indices (-> t,
        sigma) -- from ``generator``, in the reference's order (train_sana.py:183-204).  The reference hands every step a
        FRESH ``torch.Generator()`` (common/trainer.py:325), and a default-constructed CPU generator always starts from the
        same state (seed 67280421310721): its trainer draws the same noise and the same timesteps at every step of a bucket.
        Drawing 262 k bf16 normals on the CPU is the most expensive thing the step's host thread does (7 - 14 ms: a scalar
        Box-Muller loop), so the draw is remembered per (shape, generator state): a generator that arrives in a state seen
        before gets the remembered values and is left in the remembered end state -- exactly what drawing again would do.  Any
        other state (seeded generators of the tests, the exploration steps' advancing generator) draws as before."""
        if generator is None:
            torch.randn(shape, dtype=BF16, out=noise_out)
            _, t, sig = self.scheduler.sample(B, None)
            return t, sig
        cache = self.__dict__.setdefault("_draw_cache", {})
        key = (tuple(shape), B)
        state = generator.get_state()
        hit = cache.get(key)
        if hit is not None and torch.equal(hit[0], state):
            noise_out.copy_(hit[1])
            generator.set_state(hit[4])
            return hit[2], hit[3]
        torch.randn(shape, generator=generator, dtype=BF16, out=noise_out)
        _, t, sig = self.scheduler.sample(B, generator)
        cache[key] = (state, noise_out.clone(), t.clone(), sig.clone(), generator.get_state())
        return t, sig

    def _scratch(self, name, like):
        """One persistent buffer per (name, shape): a bucket that comes back finds its buffers at the same addresses."""
        cache = self.__dict__.setdefault("_scratch_cache", {})
        key = (name, tuple(like.shape), like.dtype)
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.empty_like(like)
        return t

    def _noisy(self, like):
        return self._scratch("_noisy_buf", like)

    def _target(self, like):
        return self._scratch("_target_buf", like)

    def _dpred(self, like):
        return self._scratch("_dpred_buf", like)


class PixArtRecipe(SanaRecipe):
    """``PixartSigmaTrainer.optimize`` (train_pixart_sigma.py:151-185) on the HIP path: pad the T5 embeddings to 300 rows +
    mask (:158-168) -> noise in bf16 (:170) -> logit-normal index -> ``scheduler.timesteps[index]`` (:172-174) ->
    ``add_noise`` (:176) -> model (:178-182) -> ``.chunk(2, 1)[0]`` against the noise, MSE evaluated in bf16 (:183-184).
    The pad / mask staging is SanaRecipe's; mix and loss(+dL/dpred) are one launch each."""

    def __init__(self, model, scheduler: DDPMSchedule | None = None, pad_to: int = 300, device="cuda"):
        super().__init__(model, scheduler or DDPMSchedule(), pad_to=pad_to, device=device)

    def draw(self, shape, generator=None, noise=None):
        """The reference draws the noise on the device from the global RNG (randn_tensor without a generator, :170) and the
        timestep indices on the host from the global CPU RNG (:172); a generator, when given, replaces the global streams
        (device generator -> noise, CPU generator -> both)."""
        if noise is None:
            if generator is not None and generator.device.type != "cuda":
                noise = torch.randn(shape, generator=generator, device="cpu", dtype=BF16).to(self.dev, non_blocking=True)
            else:
                noise = torch.randn(shape, generator=generator, device=self.dev, dtype=BF16)
        cpu_gen = generator if (generator is not None and generator.device.type == "cpu") else None
        t, a, c = self.scheduler.sample(shape[0], cpu_gen)
        return noise.to(self.dev), t.to(self.dev, non_blocking=True), a.to(self.dev, non_blocking=True), \
            c.to(self.dev, non_blocking=True)

    def optimize(self, latents, embeddings, generator=None, return_pred=False, noise=None):
        enc, mask, bias, kvl = self.pad_embeddings(embeddings)
        latents = latents.to(device=self.dev, dtype=BF16).contiguous()
        noise, timesteps, a, c = self.draw(latents.shape, generator, noise)
        noisy = ops.ddpm_add_noise(latents, noise, a, c)
        self.model.next_kv_work = self.kv_work
        out = self.model(noisy, encoder_hidden_states=enc, timestep=timesteps, encoder_attention_mask=mask).sample
        loss = _MseBf16Chunk.apply(out, noise, self._mse_ws)
        return (loss, out, noise) if return_pred else loss

    def train_step_device(self, latents, enc, mask_bias_kvl, noise, timesteps, coef_a, coef_c, loss_out, kv_work=None,
                          gscale=1.0):
        """Straight-line step on device-resident inputs (scripts/bench_pixart.py, ``optimize_device``): mix, forward, loss +
        dL/dpred (scaled by ``gscale``), backward."""
        bias, kvl = mask_bias_kvl
        noisy = ops.ddpm_add_noise(latents, noise, coef_a, coef_c, self._noisy(latents))
        out = self.model.forward_device(noisy, enc, timesteps, bias, kvl, kv_work=kv_work)      # (launch plans: yat_amd/flat.py)
        dpred = self._dpred(out)
        ops.mse_bf16_chunk(out, noise, loss_out, dpred, self._mse_ws, gscale=gscale)
        _report_loss(self.model, loss_out)
        self.model.backward_device(dpred)
        return loss_out

    def optimize_device(self, latents, embeddings, generator=None, gscale=1.0):
        """The trainer's step (train_pixart_sigma.py:151-185 + the backward of common/trainer.py:344) on the allocation-free
        path, as ``SanaRecipe.optimize_device``: the host side of the step -- cached latents, the ragged T5 rows, their
        offsets, the timesteps and the two add_noise coefficients the host looks up, the attention work list -- is ONE pinned
        buffer and ONE H2D copy; the noise is drawn on the device from the global RNG as the reference draws it (:170; a CPU
        generator, when given, draws it on the host in the reference's order instead); pad / mask, add_noise, forward, bf16
        loss + dL/dpred and backward are straight-line launches replayed from a launch plan.  -> loss (0-dim bf16 device
        tensor, as the reference's ``MSELoss`` on bf16 operands returns); gradients are in the flat gradient buffer."""
        if latents.is_cuda or embeddings[0].is_cuda:
            raise ValueError("optimize_device stages host batches (the sampler yields CPU tensors)")
        B, T = len(embeddings), self.pad_to
        C = embeddings[0].shape[1]
        lens = [int(e.shape[0]) for e in embeddings]
        if max(lens) > T:
            raise ValueError(f"embedding longer than pad length {T}")
        rows = sum(lens)
        pairs = [(b, t) for b, L in enumerate(lens) for t in range(((L if L > 0 else T) + 63) // 64)]     # ops.kv_work_list
        nlat = latents.numel()
        max_pairs = B * ((T + 63) // 64)
        cpu_gen = generator if (generator is not None and generator.device.type == "cpu") else None
        host_noise = cpu_gen is not None
        (o_lat, o_noise, o_off, o_t, o_a, o_c, o_work, o_emb), total = _layout(
            [2 * nlat, 2 * nlat if host_noise else 0, 4 * (B + 1), 4 * B, 2 * B, 2 * B, 8 * max_pairs, 2 * rows * C])
        st = self._stager = getattr(self, "_stager", None) or _Stager(self.dev)
        pin = st.begin(total, capacity=total + 2 * (B * T - rows) * C)

        def seg(o, n, dtype):
            return pin[o:o + n].view(dtype)
        seg(o_lat, 2 * nlat, BF16).view(latents.shape).copy_(latents)
        if host_noise:                                                                                           # :170
            torch.randn(latents.shape, generator=cpu_gen, dtype=BF16, out=seg(o_noise, 2 * nlat, BF16).view(latents.shape))
        t, a, c = self.scheduler.sample(B, cpu_gen)                                                              # :172-174
        if rows:
            _pack_rows(embeddings, seg(o_emb, 2 * rows * C, BF16).view(rows, C))
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        seg(o_off, 4 * (B + 1), torch.int32).copy_(torch.tensor(offs, dtype=torch.int32))
        seg(o_t, 4 * B, torch.float32).copy_(t)                     # int64 timestep -> the float the embedder takes (exact)
        seg(o_a, 2 * B, BF16).copy_(a)
        seg(o_c, 2 * B, BF16).copy_(c)
        seg(o_work, 8 * len(pairs), torch.int32).copy_(torch.tensor(pairs, dtype=torch.int32).flatten())
        land = st.commit(total)

        def dseg(o, n, dtype):
            return land[o:o + n].view(dtype)
        lat_d = dseg(o_lat, 2 * nlat, BF16).view(latents.shape)
        if host_noise:
            noise_d = dseg(o_noise, 2 * nlat, BF16).view(latents.shape)
        else:
            noise_d = self._scratch("_noise_buf", lat_d)
            dev_gen = generator if (generator is not None and generator.device.type == "cuda") else None
            torch.randn(latents.shape, generator=dev_gen, device=self.dev, dtype=BF16, out=noise_d)              # :170
        fixed = getattr(self, "_fixed", None)
        if fixed is None or fixed[0].shape != (B, T, C):
            fixed = self._fixed = (torch.empty(B, T, C, dtype=BF16, device=self.dev),
                                   torch.empty(B, T, dtype=torch.int64, device=self.dev),
                                   torch.empty(B, T, dtype=torch.float32, device=self.dev),
                                   torch.empty(B, dtype=torch.int32, device=self.dev),
                                   torch.zeros(1, dtype=torch.float32, device=self.dev))
        enc, mask, bias, kvl, loss_out = fixed
        ops.pad_mask(dseg(o_emb, 2 * rows * C, BF16).view(rows, C), dseg(o_off, 4 * (B + 1), torch.int32), B, T, C, enc, mask,
                     bias, kvl)                                                                                  # :158-168
        self.train_step_device(lat_d, enc, (bias, kvl), noise_d, dseg(o_t, 4 * B, torch.float32), dseg(o_a, 2 * B, BF16),
                               dseg(o_c, 2 * B, BF16), loss_out,
                               kv_work=dseg(o_work, 8 * len(pairs), torch.int32).view(len(pairs), 2), gscale=gscale)
        return loss_out[0].to(BF16)


class _MseBf16Chunk(torch.autograd.Function):
    """MSELoss()(out.chunk(2, 1)[0], noise) in bf16, with the gradient (zero on the dropped half) from the same launch."""

    @staticmethod
    def forward(ctx, out, target, ws):
        loss = torch.zeros(1, dtype=torch.float32, device=out.device)
        dpred = torch.empty_like(out)
        ops.mse_bf16_chunk(out.contiguous(), target, loss, dpred, ws)
        ctx.save_for_backward(dpred)
        return loss[0].to(BF16)

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.to(dpred.dtype), None, None


class SD3Recipe:
    """``SD35Trainer.optimize`` (train_sd35.py:165-194) on the HIP path: noise in the latents' dtype from the global RNG on the
    device (:180, ``randn_tensor`` without a generator) -> logit-normal u on the CPU (:182) -> indices (:183) ->
    ``scheduler.timesteps[indices]`` (:184) -> ``scheduler.scale_noise`` (:185) = sigma n + (1 - sigma) x in bf16 [RECALL] ->
    MMDiT with ``pooled_projections`` (:188-191) -> target = noise - latents (:192) -> ``MSELoss()`` evaluated in bf16
    (:193).  Mix / target are one launch (``yat_flow_mix``: the same three bf16 roundings, the sum commutes), loss and
    dL/dpred one launch (``yat_mse_bf16_chunk`` over the whole tensor).  A generator, when given, replaces the global streams
    (device generator -> noise, CPU generator -> noise and u) so tests can pin the draws."""

    def __init__(self, model, scheduler: FlowMatchSchedule | None = None, device="cuda"):
        self.model = model
        self.scheduler = scheduler or FlowMatchSchedule(shift=3.0)
        self.dev = torch.device(device)
        self._mse_ws = torch.empty(256, dtype=torch.float32, device=self.dev)

    def draw(self, shape, generator=None, noise=None):
        if noise is None:
            if generator is not None and generator.device.type != "cuda":
                noise = torch.randn(shape, generator=generator, device="cpu", dtype=BF16).to(self.dev, non_blocking=True)
            else:
                noise = torch.randn(shape, generator=generator, device=self.dev, dtype=BF16)
        cpu_gen = generator if (generator is not None and generator.device.type == "cpu") else None
        _, t, sig = self.scheduler.sample(shape[0], cpu_gen)
        return noise.to(self.dev), t.to(self.dev, non_blocking=True), sig.to(self.dev, non_blocking=True)

    @staticmethod
    def stack_embeddings(embeddings):
        """The sampler hands a list of per-sample (prompt_embeds [T, J], pooled [P]) pairs (shard members ``emb.pt`` /
        ``pooled.pt``); SD3 prompts are fixed-length (77 CLIP + 256 T5 tokens), so they stack without padding."""
        if isinstance(embeddings, (tuple, list)) and len(embeddings) == 2 and torch.is_tensor(embeddings[0]) \
                and embeddings[0].dim() == 3:
            return embeddings[0], embeddings[1]
        return torch.stack([e[0] for e in embeddings]), torch.stack([e[1].reshape(-1) for e in embeddings])

    def optimize(self, latents, embeddings, generator=None, return_pred=False, noise=None):
        prompt, pooled = self.stack_embeddings(embeddings)
        latents = latents.to(device=self.dev, dtype=BF16).contiguous()
        noise, timesteps, sigmas = self.draw(latents.shape, generator, noise)
        noisy, target = ops.flow_mix(latents, noise, sigmas)
        pred = self.model(noisy, encoder_hidden_states=prompt.to(self.dev, BF16), pooled_projections=pooled.to(self.dev, BF16),
                          timestep=timesteps).sample
        loss = _MseBf16Chunk.apply(pred, target, self._mse_ws)
        return (loss, pred, target) if return_pred else loss

    def train_step_device(self, latents, prompt, pooled, noise, timesteps, sigmas, loss_out, gscale=1.0):
        """Straight-line step on device-resident inputs (scripts/bench_sd35.py, ``optimize_device``): mix, forward, loss +
        dL/dpred (scaled by ``gscale``), backward."""
        noisy, target = ops.flow_mix(latents, noise, sigmas, self._scratch("_noisy_buf", latents), self._scratch("_target_buf", latents))
        pred = self.model.forward_device(noisy, prompt, pooled, timesteps)                        # (launch plans: yat_amd/flat.py)
        dpred = self._scratch("_dpred_buf", pred)
        ops.mse_bf16_chunk(pred, target, loss_out, dpred, self._mse_ws, gscale=gscale)
        _report_loss(self.model, loss_out)
        self.model.backward_device(dpred)
        return loss_out

    def optimize_device(self, latents, embeddings, generator=None, gscale=1.0):
        """The trainer's step (train_sd35.py:165-194 + the backward of common/trainer.py:344) on the allocation-free path: cached
        latents, the fixed-length prompt embeddings, the pooled projections, timesteps and sigmas travel in ONE pinned buffer and
        ONE H2D copy; the noise is drawn on the device from the global RNG as the reference draws it (:180; a CPU generator,
        when given, draws it on the host instead); mix, forward, bf16 loss + dL/dpred and backward replay a launch plan.
        -> loss (0-dim bf16 device tensor); gradients are in the flat gradient buffer."""
        prompt, pooled = self.stack_embeddings(embeddings)
        if latents.is_cuda or prompt.is_cuda:
            raise ValueError("optimize_device stages host batches (the sampler yields CPU tensors)")
        B = latents.shape[0]
        nlat, npr, npo = latents.numel(), prompt.numel(), pooled.numel()
        cpu_gen = generator if (generator is not None and generator.device.type == "cpu") else None
        host_noise = cpu_gen is not None
        (o_lat, o_noise, o_pr, o_po, o_t, o_sig), total = _layout([2 * nlat, 2 * nlat if host_noise else 0, 2 * npr, 2 * npo,
                                                                   4 * B, 2 * B])
        st = self._stager = getattr(self, "_stager", None) or _Stager(self.dev)
        pin = st.begin(total)

        def seg(o, n, dtype):
            return pin[o:o + n].view(dtype)
        seg(o_lat, 2 * nlat, BF16).view(latents.shape).copy_(latents)
        if host_noise:
            torch.randn(latents.shape, generator=cpu_gen, dtype=BF16, out=seg(o_noise, 2 * nlat, BF16).view(latents.shape))
        _, t, sig = self.scheduler.sample(B, cpu_gen)                                                            # :182-184
        seg(o_pr, 2 * npr, BF16).view(prompt.shape).copy_(prompt)
        seg(o_po, 2 * npo, BF16).view(pooled.shape).copy_(pooled)
        seg(o_t, 4 * B, torch.float32).copy_(t)
        seg(o_sig, 2 * B, BF16).copy_(sig)
        land = st.commit(total)

        def dseg(o, n, dtype):
            return land[o:o + n].view(dtype)
        lat_d = dseg(o_lat, 2 * nlat, BF16).view(latents.shape)
        if host_noise:
            noise_d = dseg(o_noise, 2 * nlat, BF16).view(latents.shape)
        else:
            noise_d = self._scratch("_noise_buf", lat_d)
            dev_gen = generator if (generator is not None and generator.device.type == "cuda") else None
            torch.randn(latents.shape, generator=dev_gen, device=self.dev, dtype=BF16, out=noise_d)              # :180
        loss_out = self.__dict__.setdefault("_loss_out", torch.zeros(1, dtype=torch.float32, device=self.dev))
        self.train_step_device(lat_d, dseg(o_pr, 2 * npr, BF16).view(prompt.shape), dseg(o_po, 2 * npo, BF16).view(pooled.shape),
                               noise_d, dseg(o_t, 4 * B, torch.float32), dseg(o_sig, 2 * B, BF16), loss_out, gscale=gscale)
        return loss_out[0].to(BF16)

    _scratch = SanaRecipe._scratch
