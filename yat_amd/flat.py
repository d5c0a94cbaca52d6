"""Flat-buffer parameter plumbing shared by the transformer classes of the HIP path (yat_amd/sana.py, yat_amd/pixart.py).

All parameters of a model live in ONE flat bf16 HBM buffer in forward-execution order, gradients in a second one (``p.grad``
are views of it): gradient norm + AdamW are single launches over the buffer, data-parallel buckets are contiguous slices
(``bucket_bounds``), and consecutive tensors (to_q | to_k | to_v) can be addressed as one fused matrix while the checkpoint
still sees the diffusers keys.  Activations kept for the backward live in a persistent arena (``_buf``).
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops

BF16 = torch.bfloat16


class _Node(nn.Module):
    """Anonymous container so parameter names can carry the diffusers dotted paths."""


def schedule(chains):
    """(weight gradients on a side stream?, forward chains) of a model's step.  ``YAT_SERIAL=1`` is the one diagnostic switch
    left of the per-model stream switches of rounds 1-4: every launch of the step on ONE stream, in program order -- the
    form whose per-kernel durations the roofline figures are taken from (bench.py's serialized pass, scripts/gpu_round.sh
    profserial)."""
    serial = os.environ.get("YAT_SERIAL", "0") != "0"
    return (not serial), (1 if serial else int(chains))


def isolate_streams():
    """Should the step's compute streams live on the high-priority level?  Only when a process group exists (data-parallel
    job, or the forced one-rank rehearsal): that is when the streams of the process group and of the copy engine crowd the
    normal level's hardware queues.  Alone, the step's four streams get a hardware queue each at the normal level anyway,
    and the trainer fed from shards was measured SLOWER with high-priority streams (84 -> 107 ms)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def compute_stream(device, role="chain"):
    """A HIP stream for the step's compute (``role``: "chain" = dependent chain / second forward chain, "side" =
    weight-gradient stream, "opt" = optimizer stream).
    The HIP runtime multiplexes streams onto a few hardware queues PER PRIORITY LEVEL, and two streams on one hardware queue
    run their kernels strictly one after the other.  In a data-parallel job the normal level is shared with every stream
    torch, the process group and the copy engine create -- the default stream, the weight-gradient stream and the optimizer
    stream were found on ONE hardware queue, 16 ms per step lost -- so there the four compute streams move to the high level,
    whose only users they are: four streams, four queues ("hardware queues")."""
    prio = _role_priority(role)
    if prio is not None:
        return _hip_stream(device, prio)
    return torch.cuda.Stream(device=device, priority=-1 if isolate_streams() else 0)


def _role_priority(role):
    """Diagnostic (round 6): ``YAT_PRIO="chain=-1,side=1,opt=1"`` gives the streams of a role an
    explicit HIP priority (-1 high, 0 normal, 1 low).  Unset in a product run."""
    spec = os.environ.get("YAT_PRIO", "")
    if not spec:
        return None
    table = dict(item.split("=") for item in spec.split(",") if item)
    return int(table[role]) if role in table else None


_HIP = None


def _hip_stream(device, priority):
    """torch.cuda.Stream accepts only the levels torch knows; the low level (+1) is created through the runtime itself."""
    import ctypes
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    with torch.cuda.device(device):
        s = ctypes.c_void_p()
        rc = _HIP.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(0), ctypes.c_int(priority))
        if rc != 0:
            raise RuntimeError(f"hipStreamCreateWithPriority({priority}) -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=device)


SHARD_ALIGN = 64     # elements: 8 parts x 16 bytes -- bucket starts of the flat layout (``_alloc_flat``)
GRAD_TAIL = 8        # 16 bytes: the flat gradient buffer stays a whole number of 16-byte vectors


# diagnostic hook (scripts/step_ablation.py): called on every new arena buffer; None in a product run
ARENA_INIT = None


class FlatParamModule(nn.Module):
    def _alloc_flat(self, specs, device, bucket_first=None):
        """``specs``: [(diffusers key, shape)] in forward-execution order.  ``bucket_first(key)`` -> True for the tensor a
        data-parallel bucket starts with: those offsets and the total are multiples of SHARD_ALIGN elements, so that every
        bucket splits into 8 equal 16-byte-aligned parts -- what a sharded optimizer step (reduce-scatter, AdamW on 1 / N of
        every bucket, all-gather: yat_amd/ddp.py, yat_amd/optim.py) and the partition-invariant gradient norm
        (``FlatAdamW.norm_pieces``) need for N in {1, 2, 4, 8}.  The padding (< 64 elements per bucket, zero parameters with zero
        gradients) belongs to the tensor in front of it as far as the optimizer is concerned and to nobody otherwise."""
        self.dev = torch.device(device)
        offs, off = [], 0
        for name, shape in specs:
            if bucket_first is not None and bucket_first(name):
                off = (off + SHARD_ALIGN - 1) // SHARD_ALIGN * SHARD_ALIGN
            offs.append(off)
            off += (math.prod(shape) + 7) // 8 * 8            # 16-byte aligned segment starts
        if bucket_first is not None:
            off = (off + SHARD_ALIGN - 1) // SHARD_ALIGN * SHARD_ALIGN
        self.numel_flat = off
        self.flat_param = torch.zeros(off, dtype=BF16, device=self.dev)
        # GRAD_TAIL spare elements behind the gradients: the data-parallel wrapper lets the step's loss ride in the first of them
        # with the top bucket's all-reduce (yat_amd/ddp.py ``on_loss``); the optimizer and the norm never see them
        self._grad_store = torch.zeros(off + GRAD_TAIL, dtype=BF16, device=self.dev)
        self.flat_grad = self._grad_store[:off]
        self.grad_tail = self._grad_store[off:]
        self.seg_start = torch.tensor(offs + [off], dtype=torch.int64)     # host copy (true tensor extents)
        self._seg_numel = [math.prod(s) for _, s in specs]
        self.P, self.G = {}, {}
        for (name, shape), o in zip(specs, offs):
            n = math.prod(shape)
            param = nn.Parameter(self.flat_param[o:o + n].view(shape))
            param.grad = self.flat_grad[o:o + n].view(shape)
            self._register(name, param)
            self.P[name], self.G[name] = param.data, param.grad
        self._offset = dict(zip([n for n, _ in specs], offs))
        self.grad_ready = None            # callable(bucket_index) set by HipDDP
        self.loss_ready = None            # callable(loss tensor) set by HipDDP: the device path reports its loss before the backward
        self.adapters = None              # yat_amd.lokr.LoKrAdapters when the config asks for PEFT adapters
        self.param_events = None          # set by FlatAdamW(overlap_update=True): one event per bucket
        self.accumulate_grads = False     # True on non-first micro-steps of gradient accumulation
        self._arena, self._chains, self._side = {}, {}, None
        self._consts, self._plans = {}, {}
        # launch plans: replay a recorded step instead of re-deriving ~600 launches in Python (see ``planned``)
        self.use_plans = os.environ.get("YAT_LAUNCH_PLANS", "1") != "0"
        self.plan_dynamic = {}            # per-step integers of the recorded calls (ops.Recorder.dynamic), set by the caller
        self._saved = None
        self._anchor = torch.zeros((), device=self.dev, requires_grad=True)
        self.gradient_checkpointing = False
        return offs, off

    # ------------------------------------------------------------------ nn.Module plumbing
    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            if not hasattr(mod, part):
                mod.add_module(part, _Node())
            mod = getattr(mod, part)
        mod.register_parameter(parts[-1], param)

    def _block_buckets(self, specs, offs, total, num_layers, first_key="scale_shift_table"):
        """bucket 0 = embedders; bucket i+1 = block i (the last one also holds the output head)."""
        names = [n for n, _ in specs]
        starts = [0]
        for i in range(num_layers):
            starts.append(offs[names.index(f"transformer_blocks.{i}.{first_key}")])
        return [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)] + [(starts[-1], total)]

    @property
    def dtype(self):
        return BF16

    @property
    def device(self):
        return self.dev

    def enable_gradient_checkpointing(self):
        """Accepted for drop-in compatibility (train_sana.py:63, train_pixart_sigma.py:34); all activations fit in 288 GB
        HBM, so nothing is recomputed."""
        self.gradient_checkpointing = True

    def _apply(self, fn, *a, **k):
        # parameters are views of flat device buffers; .to()/.cuda()/.bfloat16() must not re-materialise them
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        missing = [k for k in self.P if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self.P]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        with torch.no_grad():
            for k, v in state_dict.items():
                if k in self.P:
                    self.P[k].copy_(v.to(device=self.dev, dtype=BF16).view(self.P[k].shape))
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    # ------------------------------------------------------------------ arena
    def _buf(self, name, shape, dtype=BF16):
        n = math.prod(shape)
        t = self._arena.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            if t is not None:
                self._plans.clear()       # a buffer moves: every recorded plan may hold its old address
            t = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            if ARENA_INIT is not None:                                          # (scripts/step_ablation.py only)
                ARENA_INIT(t)
            self._arena[name] = t
        return t[:n].view(shape)

    def _const(self, name, shape, dtype, value):
        """A device constant (zeros / a fill value), created once per (name, shape): no per-step fill launch."""
        key = (name, tuple(shape), dtype, value)
        t = self._consts.get(key)
        if t is None:
            t = self._consts[key] = torch.full(tuple(shape), value, dtype=dtype, device=self.dev)
        return t

    # ------------------------------------------------------------------ launch plans
    # The forward / backward of a training step are pure functions of (shapes, buffer addresses, schedule flags): every
    # activation lives in the arena, so a second step on the same bucket issues the SAME C calls with the SAME arguments.
    # ``planned`` records them once (ops.Recorder: C-ABI calls through ops._lib(), stream / event operations through the
    # helpers below) and afterwards replays the flat list: host cost per step drops from ~25 ms of Python (tensor slicing,
    # stride arithmetic, struct building, stream look-ups for ~600 launches) to one tight loop.  Not a graph capture: the
    # launches, streams and events are exactly those of the recorded run, and anything dynamic stays dynamic -- the input
    # buffers' contents, and the integers registered in ``plan_dynamic`` (the length of the attention work list).
    def planned(self, kind, key, fn):
        if not self.use_plans or self.adapters is not None or ops.GEMM_TIMER is not None or ops.RECORDER is not None:
            return fn()
        key = (kind, torch.cuda.current_stream().cuda_stream) + tuple(key)
        plan = self._plans.get(key)
        if plan is not None:
            self.plan_replays = getattr(self, "plan_replays", 0) + 1
            out = plan.replay(self.plan_dynamic)
            if kind == "fwd":
                self._saved = plan.saved
            return out
        if len(self._plans) >= 128:
            self._plans.clear()
        rec = ops.Recorder(ops._l.load())
        ops.RECORDER = rec
        try:
            result = fn()
        finally:
            ops.RECORDER = None
        from .plan import LaunchPlan
        self._plans[key] = LaunchPlan(rec.entries, rec.dynamic, result=result, saved=self._saved)
        return result

    def _require_device(self, **tensors):
        """Inputs of a planned forward must be what the kernels read, in place: a silent ``.to()`` / ``.contiguous()`` copy
        would be made once, at recording time, and every replay would read that stale copy.  name=(tensor, dtype)."""
        for name, (t, dtype) in tensors.items():
            if t is None:
                continue
            on_dev = t.device.type == self.dev.type and (self.dev.index is None or t.device.index == self.dev.index)
            if not on_dev or t.dtype != dtype or not t.is_contiguous():
                raise ValueError(f"forward_device: `{name}` must be a contiguous {dtype} tensor on {self.dev} "
                                 f"(got {t.dtype} on {t.device}, contiguous={t.is_contiguous()})")

    # stream / event operations of the model code go through these, so that a recorder sees them
    def _ev_record(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        if ops.RECORDER is not None:
            ops.RECORDER.entries.append(["ev_record", ev, stream])
        return ev

    def _ev_wait(self, stream, ev):
        stream.wait_event(ev)
        if ops.RECORDER is not None:
            ops.RECORDER.entries.append(["wait_event", stream, ev])

    def _wait_stream(self, waiter, waited):
        waiter.wait_stream(waited)
        if ops.RECORDER is not None:
            ops.RECORDER.entries.append(["wait_stream", waiter, waited])

    def _callback(self, fn, *args):
        """A host callback in launch order (the data-parallel hook): runs now and on every replay, under the torch stream
        that is current now."""
        fn(*args)
        if ops.RECORDER is not None:
            st = torch.cuda.current_stream()

            def again(st=st, fn=fn, args=args):
                with torch.cuda.stream(st):
                    fn(*args)
            ops.RECORDER.entries.append(["py", again])

    def _fused(self, first_key, rows_total, cols=None):
        """Contiguous view spanning consecutive parameter tensors (e.g. to_q|to_k|to_v -> [3D, D])."""
        o = self._offset[first_key]
        if cols is None:
            return self.flat_param[o:o + rows_total], self.flat_grad[o:o + rows_total]
        n = rows_total * cols
        return self.flat_param[o:o + n].view(rows_total, cols), self.flat_grad[o:o + n].view(rows_total, cols)

    # ------------------------------------------------------------------ streams
    def join_pending_update(self):
        """Make the current stream wait for an optimizer update still running on the optimizer's stream."""
        pev, self.param_events = self.param_events, None
        if pev is not None:
            cur = torch.cuda.current_stream()
            for ev in pev:
                cur.wait_event(ev)

    def _chain_stream(self, c):
        if c not in self._chains:
            self._chains[c] = compute_stream(self.flat_param.device)
        return self._chains[c]

    def _side_stream(self):
        if self._side is None:
            self._side = compute_stream(self.flat_param.device, "side")
        return self._side
