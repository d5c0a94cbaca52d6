"""Fused clip + AdamW (+EMA) over the model's flat parameter buffer (C ABI: yat_gradnorm_clip, yat_adamw_step).

Mirrors the optimizer section of the reference step loop (common/trainer.py:246-248,347-356):
``clip_grad_norm_(max_norm=1.0)`` -> ``AdamW.step()`` -> ``EMAModel.step`` -> ``zero_grad()``, with torch's
defaults (betas 0.9/0.999, eps 1e-8) and the states in the parameter dtype (bf16).  Everything stays on the
device: the clip coefficient is a device scalar consumed by the AdamW kernel, so the step never syncs.
"""
from __future__ import annotations

import os

import torch

from . import ops

BF16 = torch.bfloat16


class FlatAdamW:
    def __init__(self, model, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, use_ema=False,
                 ema_decay=0.999, overlap_update=False):
        self.model = model
        dev = model.flat_param.device
        self.param_groups = [dict(lr=lr, initial_lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]
        self.max_grad_norm = max_grad_norm
        self.exp_avg = torch.zeros_like(model.flat_param)
        self.exp_avg_sq = torch.zeros_like(model.flat_param)
        self.step_count = 0
        self.seg_start = model.seg_start.to(dev)
        nseg = self.seg_start.numel() - 1
        self._ws = torch.empty(ops.gradnorm_workspace_bytes(model.numel_flat, nseg), dtype=torch.uint8, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.clip_coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.ema_shadow = model.flat_param.clone() if use_ema else None
        self.ema_decay = ema_decay
        self.ema_steps = 0
        # overlap_update: the update runs bucket by bucket (forward order) on its own stream and hands the model one
        # event per bucket; the next forward waits per bucket, so this HBM-bound pass hides under the forward GEMMs.
        # Anything else that reads parameters first calls model.join_pending_update().
        self.overlap_update = overlap_update and dev.type == "cuda"
        # (yat_adamw_step's 48-VGPR background form -- a few persistent workgroups under the forward instead of the
        # full-width kernel -- was measured and rejected in round 3; the update always uses the full-width kernel)
        self.background_blocks = 0
        self._stream = None

    def _ema_decay_now(self):
        """[RECALL] diffusers EMAModel.get_decay (use_ema_warmup=False): min(decay, (1+s)/(10+s)), 0 on the first call."""
        step = max(0, self.ema_steps - 1)
        if step <= 0:
            return 0.0
        return max(min((1 + step) / (10 + step), self.ema_decay), 0.0)

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        m = self.model
        coef = None
        if self.max_grad_norm is not None:
            ops.gradnorm_clip(m.flat_grad, self.seg_start, float(self.max_grad_norm), self.grad_norm, self.clip_coef,
                              self._ws)
            coef = self.clip_coef
        ema_decay = 0.0
        if self.ema_shadow is not None:
            self.ema_steps += 1
            ema_decay = self._ema_decay_now()

        bg = self.background_blocks if self.overlap_update else 0

        def update(lo, hi, step_count=None):
            # no gradient clear: every backward overwrites the whole flat gradient (accumulate_grads=False on the
            # first micro-step), like the reference's zero_grad(set_to_none=True) which writes nothing either
            ops.adamw_step(m.flat_param[lo:hi], m.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], coef,
                           g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                           self.step_count if step_count is None else step_count,
                           zero_grad=False, ema_shadow=None if self.ema_shadow is None else self.ema_shadow[lo:hi],
                           ema_decay=ema_decay, background=bg)

        ranges = getattr(m, "update_ranges", None)
        if ranges is not None:
            # an adapter set with module dropout: parameters whose gradient is None in the reference (adapter dropped for
            # the whole accumulation window) are skipped like torch.optim.AdamW skips them, each range at its own step
            m.join_pending_update()
            for lo, hi, sc in ranges():
                update(lo, hi, sc)
            return
        if not self.overlap_update:
            update(0, m.numel_flat)
            return
        m.join_pending_update()
        if self._stream is None:
            from .flat import compute_stream
            self._stream = compute_stream(m.flat_param.device, "opt")
        main = torch.cuda.current_stream()
        self._stream.wait_stream(main)
        # one persistent event per bucket, re-recorded every step: the next forward's waits are then the same calls on
        # the same objects step after step (a recorded launch plan can hold them)
        if getattr(self, "_events", None) is None or len(self._events) != len(m.bucket_bounds):
            self._events = [torch.cuda.Event() for _ in m.bucket_bounds]
        with torch.cuda.stream(self._stream):
            for (lo, hi), ev in zip(m.bucket_bounds, self._events):
                update(lo, hi)
                ev.record(self._stream)
        m.param_events = self._events

    def zero_grad(self, set_to_none=False):
        # the gradient clear is fused into step(); explicit calls (e.g. before the first step) still work
        self.model.flat_grad.zero_()
