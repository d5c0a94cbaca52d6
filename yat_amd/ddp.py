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

import ctypes
import os

import torch
import torch.distributed as dist


# What the reference exports before it builds its process group (utils/set_nccl_vars.py:4-9, imported by its launch scripts;
# common/trainer.py:27-28 sets the first two again) and what this build does with each: NONE is inherited.  The values were
# chosen for the author's two PCIe RTX 4070s; on an MI355X node every one of them is wrong or pointless.
# tests/test_host_logic.py holds this table to the variables the reference module really sets (tests/golden/nccl_vars.json).
REFERENCE_NCCL_ENV_NOT_INHERITED = {
    "NCCL_P2P_DISABLE": "xGMI IS the peer-to-peer path: with it off RCCL stages every gradient bucket through host memory",
    "NCCL_IB_DISABLE": "one node, no InfiniBand in the data path; left to the site's own environment",
    "NCCL_SOCKET_IFNAME": "'eth0' need not exist; the bootstrap runs over loopback (MASTER_ADDR 127.0.0.1)",
    "NCCL_BLOCKING_WAIT": "would make every collective block the host thread: the bucket all-reduces are enqueued on a side "
                          "stream and overlap the backward (async_op=True), which blocking waits defeat",
    "NCCL_ASYNC_ERROR_HANDLING": "a deprecated alias; torch's own default error handling for the process group stays in force",
    "NCCL_DEBUG": "INFO makes RCCL print on stdout, where bench.py's one JSON line lives",
}


def default_transport():
    """Which transport a data-parallel job uses when nothing is said.  ``YAT_COMM`` = torch | native wins.  Otherwise
    ``torch``: the gradient buckets go through torch.distributed's process group (RCCL on a GPU) -- the path every multi-GPU
    PyTorch job on this image takes.  The library's own communicator (``native``: one RCCL communicator per process, the
    launcher-level group on gloo) was the faster configuration in every forced ONE-rank rehearsal (79.3 ms per step against
    85.5 ms through torch's group, 77.0 ms plain; profiles/r04_d_*), but RCCL refuses two ranks on one device and no
    multi-GPU box was available to the builder, so it has never run with N > 1 ranks: it stays opt-in
    (``YAT_COMM=native``, ``bench.py --transport native``) until such a run is recorded under profiles/ (round-4 advisor)."""
    return forced_transport() or "torch"


def forced_transport():
    """``YAT_COMM`` = torch | native, or None."""
    return os.environ.get("YAT_COMM") or None


def forced_backend():
    """``YAT_DIST_BACKEND`` (gloo: several ranks sharing one GPU in a rehearsal, where RCCL cannot be used at all), or None."""
    return os.environ.get("YAT_DIST_BACKEND") or None


def group_backend(on_gpu=True):
    """Backend of the launcher-level process group: ``YAT_DIST_BACKEND`` if given; gloo when the gradients travel through the
    native transport (the group is then rendezvous / barrier / consensus only -- no second RCCL communicator and its streams
    beside the library's: that pairing cost the step 23 ms in round 3); nccl (= RCCL) otherwise on a GPU."""
    env = forced_backend()
    if env:
        return env
    if not on_gpu:
        return "gloo"
    return "gloo" if default_transport() == "native" else "nccl"


# RCCL channels of a data-parallel job: RCCL's own defaults unless somebody asks (round-5 advisor).  A channel is a persistent
# workgroup of the collective kernel, and a gemm256 workgroup owns its CU outright (144 KiB of LDS, 2 x 250 VGPRs per SIMD:
# DESIGN.md section 5), so the two never share a CU: while a bucket's all-reduce runs, every channel takes one CU away from the
# backward.  On paper (ESTIMATES -- nothing below was ever measured on N > 1 GPUs; no such box was available to the builder):
#   channels   bus GB/s   ring time of 3.21 GB   CU share while it runs   cost to the backward   last bucket (149 MB) exposed
#      16        ~145          ~39 ms                  6 %                    ~2.4 ms                  ~1.8 ms
#      24        ~215          ~26 ms                  9 %                    ~2.4 ms                  ~1.2 ms
#      64        ~300          ~19 ms                 25 %                    ~4.7 ms                  ~0.9 ms
# i.e. a cap around 24 might give the backward back ~2 ms of CU time -- or expose the last buckets if a channel moves less than
# assumed.  The reference leaves RCCL alone, and so does this package by default: the cap is OPT-IN (``YAT_RCCL_CHANNELS=N``,
# ``bench.py --rccl-channels N``) until an N > 1 sweep is recorded under profiles/.
RCCL_CHANNEL_CAP = 24            # the value the table suggests trying first; nothing applies it by itself


def apply_channel_policy(world, cap=None):
    """Called before any RCCL communicator exists; returns the NCCL_MAX_NCHANNELS in force (None = RCCL's default).
    ``cap``: None = ``YAT_RCCL_CHANNELS`` if set, else leave RCCL alone; 0 = leave RCCL alone; N = ask for at most N.
    A site's own NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS are never overridden: a request that contradicts them raises
    instead of rewriting them silently."""
    if world <= 1:
        return None
    if cap is None:
        cap = int(os.environ.get("YAT_RCCL_CHANNELS", "0") or 0)
    site_max = os.environ.get("NCCL_MAX_NCHANNELS")
    site_min = os.environ.get("NCCL_MIN_NCHANNELS")
    if cap <= 0:
        return int(site_max) if site_max else None
    if site_max and int(site_max) != cap:
        raise ValueError(f"RCCL channel cap {cap} requested, but the site exports NCCL_MAX_NCHANNELS={site_max}: "
                         f"unset one of them")
    if site_min and int(site_min) > cap:
        raise ValueError(f"RCCL channel cap {cap} requested, but the site exports NCCL_MIN_NCHANNELS={site_min} > {cap}: "
                         f"unset one of them")
    os.environ["NCCL_MAX_NCHANNELS"] = str(cap)
    return cap


def agree(ok, process_group=None, device=None):
    """True iff ``ok`` holds on every rank of the group: one MIN all-reduce of a flag -- on the host for a gloo group, on
    ``device`` (default: the current GPU) for an nccl one."""
    if not (dist.is_initialized() and dist.get_world_size(process_group) > 1):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(process_group) == "nccl":
        flag = flag.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
    return bool(int(flag.item()))


def negotiate_native(process_group=None, device=None, factory=None):
    """Every rank builds the library's communicator, then all agree on the outcome: ``(NativeComm, None)`` when every rank has
    one, else ``(None, first local error or None)`` on EVERY rank, a communicator that was built on this rank destroyed
    again.  ``factory``: NativeComm.get (tests pass a stub)."""
    err, native = None, None
    try:
        native = (factory or NativeComm.get)(process_group)
    except Exception as e:              # noqa: BLE001 -- reported by the caller, after the ranks have agreed
        err = e
    if agree(err is None, process_group, device):
        return native, None
    if native is not None:
        native.destroy()
    return None, err


class NativeComm:
    """The C-ABI transport (include/yat_hip.h, communication section): one RCCL communicator owned by libyat_hip.so.
    Rendezvous: rank 0 draws the 128-byte id and ships it over the launcher's process group (any backend) -- or nowhere in
    a one-rank job, which needs no process group at all."""

    _instance = None

    @classmethod
    def get(cls, process_group=None):
        if cls._instance is None:
            cls._instance = cls(process_group)
            cls._instance._group = process_group
        elif cls._instance._group is not process_group and (
                dist.is_initialized() and (dist.get_rank(process_group), dist.get_world_size(process_group))
                != (cls._instance.rank, cls._instance.world) and dist.get_world_size(process_group) > 1):
            # the library owns ONE communicator (csrc/comm.hip): a second HipDDP over another set of ranks would silently
            # reduce over the first group's ranks
            raise RuntimeError("NativeComm was built for another process group (rank "
                               f"{cls._instance.rank} of {cls._instance.world}); the C-ABI transport holds one communicator")
        return cls._instance

    def __init__(self, process_group=None, lib=None):
        from . import lib as _l
        self._l, self.lib = _l, (lib if lib is not None else _l.load())
        multi = dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.rank = dist.get_rank(process_group) if multi else 0
        self.world = dist.get_world_size(process_group) if multi else 1
        # Rendezvous that cannot strand a rank (round-4 advisor): yat_comm_init is collective, so nobody may enter it
        # unless EVERY rank can.  (1) every rank binds RCCL, rank 0 also draws the id -- failures are kept, not raised;
        # (2) rank 0 ALWAYS broadcasts, the id or None; (3) one MIN all-reduce over "I could and I hold an id"; only
        # then (4) the collective init.  A failure on any rank before (4) raises on all of them, after the same two
        # collectives on every rank.
        err, ident = None, [None]
        buf = ctypes.create_string_buffer(128)
        try:
            _l.check(self.lib.yat_comm_available(), "yat_comm_available")
            if self.rank == 0:
                _l.check(self.lib.yat_comm_unique_id(buf), "yat_comm_unique_id")
                ident = [bytes(buf.raw)]
        except Exception as e:          # noqa: BLE001 -- raised below, after the ranks have agreed
            err = e
        if multi:
            dist.broadcast_object_list(ident, src=0, group=process_group)
            if not agree(err is None and ident[0] is not None, process_group):
                raise err if err is not None else RuntimeError("NativeComm: another rank cannot build the communicator")
        elif err is not None:
            raise err
        _l.check(self.lib.yat_comm_init(self.rank, self.world, ident[0]), "yat_comm_init")

    def broadcast(self, t, root=0):
        self._l.check(self.lib.yat_comm_broadcast(t.data_ptr(), t.numel() * t.element_size(), root,
                                                  torch.cuda.current_stream().cuda_stream), "yat_comm_broadcast")

    def allreduce_async(self, t, bucket_id, producer_stream, comm_stream):
        self._l.check(self.lib.yat_bucket_allreduce_async(t.data_ptr(), t.numel() * t.element_size(), bucket_id,
                                                          producer_stream.cuda_stream, comm_stream.cuda_stream),
                      "yat_bucket_allreduce_async")

    def allreduce(self, t, mean=True, stream=None):
        """In-place all-reduce of a bf16 / fp32 device tensor on ``stream`` (default: the current one), stream-ordered."""
        code = {torch.bfloat16: 0, torch.float32: 1}.get(t.dtype)
        if code is None or not t.is_contiguous():
            raise ValueError("NativeComm.allreduce: contiguous bf16 or fp32 tensors")
        st = stream if stream is not None else torch.cuda.current_stream()
        self._l.check(self.lib.yat_comm_allreduce(t.data_ptr(), t.numel(), code, 0 if mean else 1, st.cuda_stream),
                      "yat_comm_allreduce")

    def reduce_scatter_async(self, t, bucket_id, producer_stream, comm_stream):
        """In-place reduce-scatter(mean) of a bucket: this rank's slice of ``t`` receives the mean (sharded optimizer step)."""
        self._l.check(self.lib.yat_bucket_reduce_scatter_async(t.data_ptr(), t.numel() * t.element_size(), bucket_id,
                                                               producer_stream.cuda_stream, comm_stream.cuda_stream),
                      "yat_bucket_reduce_scatter_async")

    def allgather(self, t, stream=None):
        """In-place all-gather of a contiguous device tensor on ``stream`` (default: the current one): slice ``rank`` is sent."""
        if not t.is_contiguous():
            raise ValueError("NativeComm.allgather: contiguous tensors")
        st = stream if stream is not None else torch.cuda.current_stream()
        self._l.check(self.lib.yat_comm_allgather(t.data_ptr(), t.numel() * t.element_size(), st.cuda_stream), "yat_comm_allgather")

    def wait(self, stream, bucket_id=-1):
        self._l.check(self.lib.yat_comm_wait(bucket_id, stream.cuda_stream), "yat_comm_wait")

    def destroy(self):
        self._l.check(self.lib.yat_comm_destroy(), "yat_comm_destroy")
        type(self)._instance = None


class HipDDP:
    """``transport``: "torch" = the gradient buckets go through torch.distributed's process group (RCCL on a GPU) -- the
    default of a multi-rank job (``default_transport``); "native" = the library's own communicator through the C ABI
    (``yat_comm_*``; opt-in with ``YAT_COMM=native``, the launcher-level group is then built over gloo:
    ``group_backend``).  Same buckets, same streams, same arithmetic (RCCL mean) either way.  ``allreduce_bulk`` sends any
    other device tensor (the EMA mean before validation) through whichever transport the buckets use.

    ``shard_optimizer`` (round 6; SURVEY.md section 5's design point; default: ``YAT_SHARD_OPTIMIZER=1``): every bucket is
    REDUCE-SCATTERED instead of all-reduced -- rank r ends up with the mean of slice r of the bucket, the other slices keep its
    local values -- ``FlatAdamW`` updates those slices only (1 / N of the 16 bytes per parameter the replicated update reads
    and writes on every rank) and all-gathers each bucket's parameters in forward order under the next forward
    (yat_amd/optim.py ``_sharded_update``); the clip norm is summed over pieces each rank owns and combined by one small
    all-reduce.  Same bytes on the wire as the all-reduce, same reduced gradients in -> the same parameters out, bit for bit
    (tests/test_ddp_gpu.py, tests/test_ddp_gloo.py).  Needs every bucket to be a whole number of N x 16-byte slices with
    N in {1, 2, 4, 8}: the models' flat layouts are built for it (yat_amd/flat.py SHARD_ALIGN); an adapter set's single small
    bucket usually is not, and has nothing to gain.  The step's logged loss then comes from the reference's own fp32
    gather (no spare gradient element rides a reduce-scatter)."""

    def __init__(self, model, process_group=None, average=True, force=False, transport=None, coalesce=1, shard_optimizer=None):
        self.model = model
        self.force = force            # run the collectives even in a one-rank group (single-GPU rehearsal of the N>1 path)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.sync = True
        self.on_gpu = model.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=model.flat_grad.device) if self.on_gpu else None
        if transport is None:
            transport = default_transport()
            if force and self.on_gpu and not dist.is_initialized() and forced_transport() is None:
                transport = "native"      # a forced one-rank rehearsal without any process group: only the library can run it
        if transport not in ("torch", "native"):
            raise ValueError(f"transport {transport!r}: torch | native")
        self.native = None
        if transport == "native" and self.on_gpu and (self.world > 1 or force):
            if not average:
                raise ValueError("the native transport reduces to the mean (DDP semantics)")
            # Every rank must end up on the same transport: build the communicator, agree on the outcome over the process
            # group, and if ANY rank failed (no librccl, a communicator error) all of them fall back to torch.distributed's
            # RCCL group -- created here, collectively, when the launcher-level group is gloo.
            self.native, err = negotiate_native(process_group, model.flat_grad.device)
            if self.native is None:
                import warnings
                warnings.warn(f"HipDDP: the native transport is not available on every rank ({err!r}); falling back to "
                              f"torch.distributed's RCCL group")
                if not dist.is_initialized():
                    raise err
                if dist.get_backend(process_group) != "nccl":
                    self.pg = dist.new_group(backend="nccl")          # (collective: every rank is here)
        self._works = []
        self.bytes_reduced = 0
        # diagnostics (bench.py's `comm` object; off in the timed region).  ``dryrun``: the whole machinery -- hooks, events,
        # streams, the optimizer's wait -- without the collective itself (step time with vs without = what the collective
        # costs the step, overlap and interference included).  ``timing``: HIP events around every bucket's collective on
        # the communication stream and around the compute stream's wait for it.
        self.dryrun = os.environ.get("YAT_DDP_DRYRUN", "0") != "0"
        self.timing = False
        self.timed_buckets = []          # (bucket index, bytes, start event, end event) per collective
        self.timed_waits = []            # (before, after) events of the compute stream's wait in ``wait()``
        self.buckets_reduced = 0
        # coalesce = k: k consecutive buckets (they complete in reverse order and are adjacent in the flat gradient buffer)
        # go out as one collective -- fewer, larger messages; 1 = one per transformer block (bench.py --coalesce)
        self.coalesce = max(1, int(coalesce))
        self._pending = None
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("YAT_SHARD_OPTIMIZER", "0") != "0"
        self.shard = None
        model.shard = None
        if shard_optimizer and (self.world > 1 or force):
            from types import SimpleNamespace
            rank = dist.get_rank(self.pg) if (dist.is_initialized() and self.world > 1) else 0
            if 8 % self.world:
                raise ValueError(f"shard_optimizer: {self.world} ranks (1, 2, 4 or 8: a bucket is cut into eighths)")
            bad = [i for i, (lo, hi) in enumerate(model.bucket_bounds) if (hi - lo) % 64 or hi <= lo]
            if bad:
                raise ValueError(f"shard_optimizer: buckets {bad[:4]} of this {type(model).__name__} are not whole numbers of "
                                 f"8 x 16-byte parts; use the replicated optimizer step")
            if self.coalesce > 1:
                raise ValueError("shard_optimizer: buckets are scattered one by one (coalesce=1)")
            if not average:
                raise ValueError("shard_optimizer reduces to the mean (DDP semantics)")
            self.shard = model.shard = SimpleNamespace(ddp=self, rank=rank, world=self.world)
        model.grad_ready = self.bucket_ready
        # The logged loss (``accelerator.gather(avg_loss).mean()``, common/trainer.py:359) without a collective of its own:
        # a caller that wants it calls ``track_loss(running sum)`` before the micro-step; the model reports the step's loss
        # when it exists (``on_loss``: before the backward), the sum is written into the spare element behind the gradients
        # and travels with the top bucket -- the first one the backward completes -- and ``wait()`` leaves the mean over
        # ranks in ``carried_loss``.  Without ``track_loss`` (bench.py) nothing of this runs.
        tail = getattr(model, "grad_tail", None)
        self._tail = tail if (tail is not None and tail.numel() >= 2 and hasattr(model, "_grad_store")
                              and self.shard is None) else None
        self._loss_acc, self._track, self._tail_armed, self._tail_sent = None, False, False, False
        self._loss_offset = None             # previous step's carried mean (device scalar, the same on every rank)
        # which value train/loss shows under data parallel: the carried one (default: bf16 head + remainder of the difference
        # to the previous mean, ~1e-4 relative at eight ranks) or, with YAT_LOSS_GATHER=1, exactly the reference's fp32
        # gather(avg_loss).mean() at the price of its per-step collective
        self.loss_gather = os.environ.get("YAT_LOSS_GATHER", "0") != "0"
        self.carried_loss = None
        if self._tail is not None:
            model.loss_ready = self.on_loss

    def broadcast_parameters(self, src=0):
        """accelerator.prepare -> DDP's rank0 -> all parameter broadcast (trainer.py:253)."""
        if self.world > 1 or self.force:
            if self.native is not None:
                self.native.broadcast(self.model.flat_param, src)
            else:
                dist.broadcast(self.model.flat_param, src=src, group=self.pg)

    def allreduce_bulk(self, t, mean=True):
        """In-place all-reduce of a flat device / host tensor outside the bucket schedule -- the EMA shadow before validation
        (common/trainer.py:374-377) -- through the transport the gradients use: the library's communicator when it is the
        transport (a gloo rendezvous group would stage GBs through the host), else this object's process group."""
        if self.world == 1 and not self.force:
            return t
        if self.native is not None and t.is_cuda:
            self.native.allreduce(t, mean=mean)
            return t
        if dist.is_initialized() and dist.get_world_size(self.pg) > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            if mean:
                t /= dist.get_world_size(self.pg)
        return t

    def allgather_bulk(self, t):
        """In-place all-gather of a contiguous flat tensor whose slice ``rank`` (of ``world`` equal 16-byte-aligned slices) this
        rank holds -- a bucket's updated parameters, or its EMA shadow -- stream-ordered on the CURRENT stream: the collective
        starts after what that stream has enqueued and the stream continues after it."""
        n = self.world
        if t.numel() * t.element_size() % (16 * n) or not t.is_contiguous():
            raise ValueError("allgather_bulk: a contiguous tensor of world x 16-byte slices")
        if self.world == 1 and not self.force:
            return t
        if self.native is not None and t.is_cuda:
            self.native.allgather(t)
            return t
        if not (dist.is_initialized() and dist.get_world_size(self.pg) >= 1):
            return t
        rank, per = dist.get_rank(self.pg), t.numel() // n
        own = t[rank * per:(rank + 1) * per]
        if t.is_cuda and dist.get_backend(self.pg) == "nccl":
            dist.all_gather_into_tensor(t, own, group=self.pg, async_op=True).wait()     # (the current stream waits, not the host)
        else:
            parts = [torch.empty_like(own) for _ in range(n)]                            # gloo: CPU tests / one-GPU rehearsal
            dist.all_gather(parts, own.clone(), group=self.pg)
            for r_, part in enumerate(parts):
                if r_ != rank:
                    t[r_ * per:(r_ + 1) * per].copy_(part)
        return t

    def track_loss(self, running_sum=None):
        """Arm the piggyback for the coming micro-step; ``running_sum``: the window's earlier micro-step losses (device
        scalar) that ``gather(avg_loss)`` would have included, or None.  ``YAT_LOSS_GATHER=1`` keeps the reference's own
        fp32 collective instead (common/trainer.py:359): nothing is armed and ``HipAccelerator.mean_loss`` gathers."""
        if self.loss_gather:
            return
        self._loss_acc, self._track = running_sum, True

    def on_loss(self, loss):
        if not self._track or self._tail is None or (self.world == 1 and not self.force):
            return
        x = loss.detach().float().reshape(())
        if self._loss_acc is not None:
            x = x + self._loss_acc.detach().float().reshape(()).to(x.device)
        # The slots have the gradient buffer's dtype (bf16: 8 bits) and the reduction rounds in it; the reference's gathered
        # loss is fp32 (train_sana.py:216).  So what travels is the DIFFERENCE to the previous step's mean (identical on
        # every rank: it came out of the same all-reduce), split into a bf16 head and a bf16 remainder: the rounding of the
        # reduction then scales with how far the ranks' losses are from last step's mean, not with the loss itself.
        if self._loss_offset is not None:
            x = x - self._loss_offset.to(x.device)
        hi = x.to(self._tail.dtype)
        lo = torch.where(torch.isfinite(hi.float()), x - hi.float(), torch.zeros_like(x)).to(self._tail.dtype)   # (inf - inf)
        self._tail[0:2].copy_(torch.stack([hi, lo]))
        self._tail_armed = True

    def bucket_ready(self, i):
        if (self.world == 1 and not self.force) or not self.sync:
            return
        lo, hi = self.model.bucket_bounds[i]
        if self.coalesce > 1:
            if self._pending is not None and self._pending[0] == hi:          # adjacent below the pending range: extend it
                self._pending = (lo, self._pending[1], self._pending[2] + 1)
            else:
                self._pending = (lo, hi, 1) if self._pending is None else self._pending
                if self._pending[0] != lo and self._pending[1] != hi:         # not adjacent (never with the models' order)
                    raise RuntimeError("HipDDP: buckets completed out of order")
            if self._pending[2] < self.coalesce and i != 0:
                return
            (lo, hi, _), self._pending = self._pending, None
        chunk = self.model.flat_grad[lo:hi]
        if self._tail_armed and hi == self.model.flat_grad.numel():       # the top bucket carries the loss element(s)
            chunk = self.model._grad_store[lo:hi + self._tail.numel()]
            self._tail_armed, self._tail_sent = False, True
        self.bytes_reduced += chunk.numel() * chunk.element_size()
        self.buckets_reduced += 1
        rccl = self.on_gpu and (self.native is not None or dist.get_backend(self.pg) == "nccl")
        own = None
        if self.shard is not None:                       # reduce-scatter: this rank's slice of the bucket receives the mean
            per = (hi - lo) // self.shard.world
            own = chunk[self.shard.rank * per:(self.shard.rank + 1) * per]

        def scatter_over_gloo():
            # gloo has no reduce-scatter: sum a COPY and keep the own slice only, so that -- as with RCCL's in-place form --
            # every other slice of the bucket still holds this rank's local values (nothing may read them as reduced)
            tmp = chunk.clone()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.pg)
            mine = tmp[self.shard.rank * per:(self.shard.rank + 1) * per]
            own.copy_(mine / self.world if self.average else mine)
        if self.dryrun and rccl and not self.timing and self.native is None:
            return                  # torch transport, collective off: its live form below touches no stream of ours either
        if self.on_gpu and (self.timing or (self.dryrun and rccl)):
            # diagnostic form, both transports: the producer's event and the communication stream's wait for it are issued
            # here, so that the start event sits between that wait and the collective (inside the library / process group
            # the two are one call) and the end event behind the collective's completion on the communication stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                e0 = e1 = None
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                if self.dryrun and rccl:
                    pass
                elif self.native is not None:
                    (self.native.allreduce_async if own is None else self.native.reduce_scatter_async)(
                        chunk, i, self.comm_stream, self.comm_stream)                            # (runs ON that stream)
                elif rccl and own is not None:
                    dist.reduce_scatter_tensor(own, chunk, op=dist.ReduceOp.AVG, group=self.pg, async_op=True).wait()
                elif rccl:
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM, group=self.pg,
                                    async_op=True).wait()          # the communication stream waits; the host does not
                elif own is not None:
                    scatter_over_gloo()
                else:
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg)
                    if self.average:
                        chunk.div_(self.world)
                if self.timing:
                    e1.record(self.comm_stream)
                    self.timed_buckets.append((i, chunk.numel() * chunk.element_size(), e0, e1))
            return
        if self.native is not None:
            # event on the stream that finished the bucket -> all-reduce on the communication stream, all inside the library
            (self.native.allreduce_async if own is None else self.native.reduce_scatter_async)(
                chunk, i, torch.cuda.current_stream(), self.comm_stream)
            return
        op = dist.ReduceOp.AVG if (self.average and rccl) else dist.ReduceOp.SUM
        if own is not None and rccl:
            self._works.append(dist.reduce_scatter_tensor(own, chunk, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
        elif own is not None and self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                scatter_over_gloo()
        elif own is not None:
            scatter_over_gloo()                          # (CPU multi-process tests: synchronous)
        elif rccl:
            # torch's RCCL group runs every collective on a stream of its own and makes THAT stream wait for the stream the
            # call is issued from -- here the one that has just finished the bucket -- so no stream of ours sits in between.
            # (Through round 4 the call went out under ``comm_stream`` behind an event: one more stream for the runtime to
            # fold onto its few hardware queues, and the forced one-rank step ran 86.3 ms against 79.7 ms through the
            # library's communicator, which has exactly this shape; profiles/r05_c_ddp_forced_one_rank_*.)
            self._works.append(dist.all_reduce(chunk, op=op, group=self.pg, async_op=True))
        elif self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                # one-GPU rehearsal over gloo (device tensors staged through the host): sum, then the mean
                dist.all_reduce(chunk, op=op, group=self.pg)
                if self.average:
                    chunk.div_(self.world)
        else:  # gloo path used by the CPU multi-process tests
            w = dist.all_reduce(chunk, op=op, group=self.pg, async_op=True)
            self._works.append((w, chunk))

    def wait(self):
        """Called before the optimizer: the compute stream waits for every outstanding bucket."""
        if self.on_gpu and (self.timing or self.dryrun):
            cur = torch.cuda.current_stream()
            if self.timing:
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record(cur)
            cur.wait_stream(self.comm_stream)
            if self.native is not None and not self.dryrun:
                self.native.wait(cur)                        # (clears the library's per-bucket pending flags)
            if self.timing:
                w1.record(cur)
                self.timed_waits.append((w0, w1))
            self._works.clear()
            self._harvest_loss()
            return
        if self.native is not None:
            self.native.wait(torch.cuda.current_stream())
        elif self.on_gpu:
            for w in self._works:
                w.wait()                         # makes the current stream wait; does not block the host
            if not (dist.is_initialized() and dist.get_backend(self.pg) == "nccl"):
                torch.cuda.current_stream().wait_stream(self.comm_stream)     # (the gloo rehearsal's staging copies)
        else:
            for w, chunk in self._works:
                w.wait()
                if self.average:
                    chunk.div_(self.world)
        self._works.clear()
        self._harvest_loss()

    def _harvest_loss(self):
        self._track, self._loss_acc = False, None
        if not self._tail_sent:
            self._tail_armed = False
            return
        self._tail_sent = False
        v = self._tail[0].float() + self._tail[1].float()       # (a new tensor: the slots are cleared below)
        if not self.average:
            v = v / self.world
        if self._loss_offset is not None:
            v = v + self._loss_offset.to(v.device)
        self.carried_loss = v
        if self.dryrun:
            # the tail was not reduced: v is this rank's own value, and an offset taken from it would differ between ranks
            self._loss_offset = None
        else:
            # A non-finite loss must not poison every later step (inf - inf = nan in the difference that travels): the offset
            # only ever follows finite means -- the reference's gather(avg_loss).mean() recovers on the next step, so does
            # this.  On the device, no host sync; every rank sees the same reduced v and takes the same branch.
            prev = self._loss_offset.to(v.device) if self._loss_offset is not None else torch.zeros_like(v)
            self._loss_offset = torch.where(torch.isfinite(v), v, prev).detach().clone()
        self._tail.zero_()

    def all_reduce_scalar_mean(self, t):
        """accelerator.gather(avg_loss).mean() (trainer.py:359) as one tiny all-reduce."""
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            t /= self.world
        return t
