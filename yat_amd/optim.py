"""Fused clip + AdamW (+EMA) over the model's flat parameter buffer (C ABI: yat_gradnorm_pieces_*, yat_adamw_step).

Mirrors the optimizer section of the reference step loop (common/trainer.py:246-248,347-356):
``clip_grad_norm_(max_norm=1.0)`` -> ``AdamW.step()`` -> ``EMAModel.step`` -> ``zero_grad()``, with torch's
defaults (betas 0.9/0.999, eps 1e-8) and the states in the parameter dtype (bf16).  Everything stays on the
device: the clip coefficient is a device scalar consumed by the AdamW kernel, so the step never syncs.

Sharded step (round 6; SURVEY.md section 5, yat_amd/ddp.py ``HipDDP(shard_optimizer=True)``): in a data-parallel job of N ranks
rank r holds the reduced gradients of slice r of every bucket only (reduce-scatter instead of all-reduce), updates those
parameters (1 / N of the AdamW traffic) and the buckets' parameters come back by all-gather in forward order under the next
forward.  The update is elementwise and the gradient norm is computed over PIECES -- tensors cut at the eighths of their
bucket, ``norm_pieces`` -- whose sums do not depend on who computed them, so the sharded step is BIT-IDENTICAL to the
replicated one (same reduced gradients in, same parameters out) for N in {1, 2, 4, 8}.
"""
from __future__ import annotations

import os

import torch

from . import ops

BF16 = torch.bfloat16


NORM_CHUNK = 1 << 18        # elements per partial sum of the gradient norm (csrc/optim.hip NORM_CHUNK)
NORM_PARTS = 8              # a bucket's pieces never straddle an eighth of it: shards of 1, 2, 4 or 8 ranks own whole pieces


def norm_pieces(seg_start, bucket_bounds, parts=NORM_PARTS):
    """The units the gradient norm is summed over.  ``seg_start``: the tensors' offsets in the flat buffer (+ the total);
    ``bucket_bounds``: the data-parallel buckets.  Every tensor is cut where an eighth of its bucket ends (only buckets whose
    length is a multiple of 8 x 16 bytes are cut: the models' own layouts, yat_amd/flat.py SHARD_ALIGN) ->
    (piece_start [np + 1], tensor_first_piece [nt + 1], chunk_base [np + 1], max chunks of a piece,
     piece_part [np]: (bucket, eighth) of each piece, or (bucket, -1) in a bucket that is not cut)."""
    seg = [int(x) for x in seg_start]
    cuts, part_of = set(), []
    for bi, (lo, hi) in enumerate(bucket_bounds):
        L = hi - lo
        cuts.add(lo)                     # (bucket starts are tensor starts in every model; a cut there costs nothing)
        if L > 0 and L % (8 * parts) == 0:
            cuts.update(lo + k * (L // parts) for k in range(1, parts))
    cuts = sorted(cuts)
    piece_start, tensor_first = [], []
    import bisect
    for t in range(len(seg) - 1):
        s0, s1 = seg[t], seg[t + 1]
        tensor_first.append(len(piece_start))
        if s1 <= s0:
            continue
        inner = cuts[bisect.bisect_right(cuts, s0):bisect.bisect_left(cuts, s1)]
        piece_start.extend([s0] + inner)
    tensor_first.append(len(piece_start))
    piece_start.append(seg[-1])
    lows = [lo for lo, _ in bucket_bounds]
    for p0 in piece_start[:-1]:
        bi = max(0, bisect.bisect_right(lows, p0) - 1)
        lo, hi = bucket_bounds[bi]
        L = hi - lo
        part_of.append((bi, (p0 - lo) // (L // parts)) if (L > 0 and L % (8 * parts) == 0) else (bi, -1))
    chunk_base, mx = [0], 1
    for a, b in zip(piece_start[:-1], piece_start[1:]):
        n = (b - a + NORM_CHUNK - 1) // NORM_CHUNK
        chunk_base.append(chunk_base[-1] + n)
        mx = max(mx, n)
    return piece_start, tensor_first, chunk_base, mx, part_of


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
        # gradient norm over pieces (tensors cut at the eighths of their bucket): the sums a sharded step can split over ranks
        ps, tf, cb, self._max_chunks, self._piece_part = norm_pieces(model.seg_start.tolist(), list(model.bucket_bounds))
        self._piece_start = torch.tensor(ps, dtype=torch.int64, device=dev)
        self._tensor_first = torch.tensor(tf, dtype=torch.int32, device=dev)
        self._chunk_base = torch.tensor(cb, dtype=torch.int32, device=dev)
        self._partial = torch.zeros(cb[-1], dtype=torch.float32, device=dev)
        self._owned, self._owned_for = None, None
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
        shard = self._shard()
        if self.max_grad_norm is not None:
            ops.gradnorm_pieces_partial(m.flat_grad, self._piece_start, self._chunk_base, self._max_chunks,
                                        None if shard is None else self._owned_mask(shard), self._partial)
            if shard is not None:
                # every slot has exactly one non-zero contributor (the rank that owns the piece): the sum over ranks is exact
                shard.ddp.allreduce_bulk(self._partial, mean=False)
            ops.gradnorm_pieces_finish(self._tensor_first, self._chunk_base, self._partial, float(self.max_grad_norm),
                                       self.grad_norm, self.clip_coef)
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
        if shard is not None:
            if ranges is not None:
                raise RuntimeError("a sharded optimizer step over an adapter set with module dropout is not built")
            self._sharded_update(shard, update)
            return
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

    # ------------------------------------------------------------------ sharded step
    def _shard(self):
        """The model's shard descriptor (set by ``HipDDP(shard_optimizer=True)``: .ddp, .rank, .world), or None."""
        return getattr(self.model, "shard", None)

    def _owned_mask(self, shard):
        key = (shard.rank, shard.world)
        if self._owned_for != key:
            per = NORM_PARTS // shard.world
            own = [1 if (part >= 0 and shard.rank * per <= part < (shard.rank + 1) * per) else 0 for _, part in self._piece_part]
            if any(part < 0 for _, part in self._piece_part):
                raise RuntimeError("sharded optimizer step: a bucket of this model is not a whole number of 8 x 16-byte parts")
            self._owned = torch.tensor(own, dtype=torch.uint8, device=self.model.flat_param.device)
            self._owned_for = key
        return self._owned

    def _sharded_update(self, shard, update):
        """AdamW on this rank's slice of every bucket, forward order; each bucket's parameters all-gathered right behind its
        update on a stream of their own, so that the update of bucket i + 1 runs beside the gather of bucket i and the next
        forward waits per bucket (``param_events``) exactly as it does for the replicated update."""
        m, ddp, r, n = self.model, shard.ddp, shard.rank, shard.world
        m.join_pending_update()
        main = torch.cuda.current_stream()
        if not self.overlap_update:
            for lo, hi in m.bucket_bounds:
                s_ = (hi - lo) // n
                update(lo + r * s_, lo + (r + 1) * s_)
                ddp.allgather_bulk(m.flat_param[lo:hi])
            return
        from .flat import compute_stream
        if self._stream is None:
            self._stream = compute_stream(m.flat_param.device, "opt")
        if getattr(self, "_gather_stream", None) is None:
            self._gather_stream = compute_stream(m.flat_param.device, "opt")
        upd, gat = self._stream, self._gather_stream
        upd.wait_stream(main)
        nb = len(m.bucket_bounds)
        if getattr(self, "_events", None) is None or len(self._events) != nb:
            self._events = [torch.cuda.Event() for _ in range(nb)]
        if getattr(self, "_upd_events", None) is None or len(self._upd_events) != nb:
            self._upd_events = [torch.cuda.Event() for _ in range(nb)]
        for i, (lo, hi) in enumerate(m.bucket_bounds):
            s_ = (hi - lo) // n
            with torch.cuda.stream(upd):
                update(lo + r * s_, lo + (r + 1) * s_)
                self._upd_events[i].record(upd)
            with torch.cuda.stream(gat):
                gat.wait_event(self._upd_events[i])
                ddp.allgather_bulk(m.flat_param[lo:hi])
                self._events[i].record(gat)
        m.param_events = self._events

    def gather_ema(self):
        """Sharded step: every rank's EMA shadow is current on its own slices only; before anybody reads the whole shadow
        (validation / save, common/trainer.py:371-383) the slices are all-gathered -- this replaces the reference's mean over
        ranks of identical shadows."""
        shard = self._shard()
        if shard is None or self.ema_shadow is None:
            return False
        self.model.join_pending_update()
        for lo, hi in self.model.bucket_bounds:
            shard.ddp.allgather_bulk(self.ema_shadow[lo:hi])
        return True

    def zero_grad(self, set_to_none=False):
        # the gradient clear is fused into step(); explicit calls (e.g. before the first step) still work
        self.model.flat_grad.zero_()
