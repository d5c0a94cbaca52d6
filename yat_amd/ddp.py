"""Data-parallel gradient reduction over RCCL/xGMI, replacing Accelerate's DDP wrap
(common/trainer.py:31-37,253; fired inside accelerator.backward at :344).

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  The model's gradients
live in ONE flat buffer laid out in forward order, cut into contiguous buckets (embedders | one
per transformer block, ~156 MB each at SANA-1.6B -- large enough that every per-peer chunk of a
direct reduce-scatter/all-gather stays on the xGMI bandwidth plateau, unlike torch's 25 MiB
default).  The backward pass calls ``bucket_ready(i)`` the moment bucket i's last gradient kernel
is enqueued; the all-reduce is issued on a side HIP stream behind an event, so communication of
block i overlaps the backward compute of blocks < i.  The optimizer waits on the side stream.

No find_unused_parameters graph walk, no per-parameter hooks, no gradient copies.
Gradient accumulation (accelerator.accumulate / no_sync, trainer.py:317): with ``sync=False`` the
callback does nothing and gradients keep accumulating in place.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HipDDP:
    def __init__(self, model, process_group=None, average=True, force=False):
        self.model = model
        self.force = force            # run the collectives even in a one-rank group (single-GPU rehearsal of the N>1 path)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.sync = True
        self.on_gpu = model.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=model.flat_grad.device) if self.on_gpu else None
        self._works = []
        self.bytes_reduced = 0
        model.grad_ready = self.bucket_ready

    def broadcast_parameters(self, src=0):
        """accelerator.prepare -> DDP's rank0 -> all parameter broadcast (trainer.py:253)."""
        if self.world > 1 or self.force:
            dist.broadcast(self.model.flat_param, src=src, group=self.pg)

    def bucket_ready(self, i):
        if (self.world == 1 and not self.force) or not self.sync:
            return
        lo, hi = self.model.bucket_bounds[i]
        chunk = self.model.flat_grad[lo:hi]
        op = dist.ReduceOp.AVG if (self.average and self.on_gpu) else dist.ReduceOp.SUM
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append(dist.all_reduce(chunk, op=op, group=self.pg, async_op=True))
        else:  # gloo path used by the CPU multi-process tests
            w = dist.all_reduce(chunk, op=op, group=self.pg, async_op=True)
            self._works.append((w, chunk))
        self.bytes_reduced += chunk.numel() * chunk.element_size()

    def wait(self):
        """Called before the optimizer: the compute stream waits for every outstanding bucket."""
        if self.on_gpu:
            for w in self._works:
                w.wait()                         # makes the current stream wait; does not block the host
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w, chunk in self._works:
                w.wait()
                if self.average:
                    chunk.div_(self.world)
        self._works.clear()

    def all_reduce_scalar_mean(self, t):
        """accelerator.gather(avg_loss).mean() (trainer.py:359) as one tiny all-reduce."""
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            t /= self.world
        return t
