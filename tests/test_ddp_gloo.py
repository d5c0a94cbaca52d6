"""N > 1 path on CPU: world_size-2 gloo processes exercising the bucketed gradient reduction, the parameter
broadcast, and the bucket sampler's one-exchange-per-batch consensus."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FlatModel:
    """The attributes HipDDP needs from SanaTransformer2DModelHIP, on CPU."""

    def __init__(self, n, bounds):
        self.flat_param = torch.zeros(n, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, dtype=torch.float32)
        self.bucket_bounds = bounds
        self.grad_ready = None


def _ddp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yat_amd.ddp import HipDDP
    from yat_amd.common.trainer import HipAccelerator
    m = _FlatModel(100, [(0, 30), (30, 70), (70, 100)])
    m.flat_param += rank + 1
    ddp = HipDDP(m)
    ddp.broadcast_parameters()
    assert torch.all(m.flat_param == 1)                       # rank 0's weights everywhere
    m.flat_grad[:] = torch.arange(100, dtype=torch.float32) * (rank + 1)
    for i in (2, 1, 0):                                       # backward order: last bucket first
        m.grad_ready(i)
    ddp.wait()
    assert torch.allclose(m.flat_grad, torch.arange(100, dtype=torch.float32) * 1.5)       # mean over 2 ranks
    # no_sync (gradient accumulation): nothing is reduced
    ddp.sync = False
    m.flat_grad[:] = rank
    m.grad_ready(0)
    ddp.wait()
    assert torch.all(m.flat_grad == rank)
    t = ddp.all_reduce_scalar_mean(torch.tensor([float(rank)]))
    assert t.item() == 0.5
    assert ddp.carried_loss is None                           # a model without spare gradient elements: nothing rides along

    # the REAL model's flat layout and bucket order (built on the CPU: no kernel runs here): buckets complete last block
    # first, embedders last, exactly as backward_impl reports them; every bucket is reduced once, the tail of the last
    # bucket (output head) included, and a second "step" reuses the same machinery
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    real = SanaTransformer2DModelHIP(SanaConfig(num_layers=3, num_attention_heads=2, num_cross_attention_heads=2,
                                                cross_attention_head_dim=32, cross_attention_dim=64, caption_channels=96,
                                                in_channels=8, out_channels=8, sample_size=4), device="cpu")
    assert real.bucket_bounds[0][0] == 0 and real.bucket_bounds[-1][1] == real.numel_flat and len(real.bucket_bounds) == 4
    assert all(a[1] == b[0] for a, b in zip(real.bucket_bounds, real.bucket_bounds[1:]))
    real.flat_param = real.flat_param.float()
    real.flat_grad = real.flat_grad.float()                   # gloo has no bf16 all-reduce; the layout is what is under test
    real.flat_param += rank
    ddp_real = HipDDP(real)
    ddp_real.broadcast_parameters()
    assert torch.all(real.flat_param == 0)
    for step in range(2):
        real.flat_grad[:] = torch.arange(real.numel_flat, dtype=torch.float32) * (rank + 1 + step)
        for i in (3, 2, 1, 0):
            real.grad_ready(i)
        ddp_real.wait()
        assert torch.allclose(real.flat_grad, torch.arange(real.numel_flat, dtype=torch.float32) * (1.5 + step))
    assert ddp_real.bytes_reduced == 2 * real.numel_flat * 4

    # the logged loss rides behind the top bucket (trainer.py:359's gather(avg_loss).mean() without its own collective):
    # armed per micro-step by track_loss, reported by the model before its backward, harvested by wait()
    from yat_amd.flat import GRAD_TAIL
    store = torch.zeros(real.numel_flat + GRAD_TAIL)
    real._grad_store, real.flat_grad, real.grad_tail = store, store[:real.numel_flat], store[real.numel_flat:]
    ddp_l = HipDDP(real)
    assert real.loss_ready is not None
    for step, (acc, loss) in enumerate([(None, 0.25 + rank), (torch.tensor(1.0 + rank), 0.5)]):
        real.flat_grad[:] = float(rank + 1)
        ddp_l.track_loss(acc)
        real.loss_ready(torch.tensor([loss]))                 # the device path's report (recipe._report_loss)
        for i in (3, 2, 1, 0):
            real.grad_ready(i)
        ddp_l.wait()
        want = [0.75, 2.0][step]                              # mean over ranks of (running sum + loss)
        assert abs(ddp_l.carried_loss.item() - want) < 1e-6 and torch.all(real.flat_grad == 1.5)
        assert torch.all(real.grad_tail == 0)
        ddp_l.carried_loss = None
    # one rank's loss spikes to inf for ONE step: that step's logged mean is inf (as the reference's gather would show), and
    # the next steps' are finite again -- the offset that travels only ever follows finite means (round-3 advisor finding)
    for loss, want in ((float("inf") if rank == 1 else 0.5, float("inf")), (0.25 + rank, 0.75), (1.0, 1.0)):
        real.flat_grad[:] = 1.0
        ddp_l.track_loss(None)
        real.loss_ready(torch.tensor([loss]))
        for i in (3, 2, 1, 0):
            real.grad_ready(i)
        ddp_l.wait()
        got = ddp_l.carried_loss.item()
        assert (got == want) if want == float("inf") else abs(got - want) < 1e-6, (got, want)
        ddp_l.carried_loss = None
    assert torch.isfinite(ddp_l._loss_offset).all()
    # a diagnostic pass that does not reduce (bench.py's dry pass) must not leave a per-rank offset behind
    ddp_l.dryrun = True
    real.flat_grad[:] = 1.0
    ddp_l.track_loss(None)
    real.loss_ready(torch.tensor([5.0 + rank]))
    for i in (3, 2, 1, 0):
        real.grad_ready(i)
    ddp_l.wait()
    ddp_l.dryrun, ddp_l.carried_loss = False, None
    assert ddp_l._loss_offset is None
    # YAT_LOSS_GATHER=1: the reference's own fp32 gather -- nothing is armed, mean_loss gathers
    ddp_l.loss_gather = True
    real.flat_grad[:] = 1.0
    ddp_l.track_loss(None)
    real.loss_ready(torch.tensor([2.0 + rank]))
    for i in (3, 2, 1, 0):
        real.grad_ready(i)
    ddp_l.wait()
    assert ddp_l.carried_loss is None and torch.all(real.grad_tail == 0)
    acc_g = HipAccelerator(1, device="cpu")
    acc_g.ddp = ddp_l
    assert acc_g.mean_loss(torch.tensor(2.0 + rank)).item() == 2.5
    ddp_l.loss_gather = False
    real.flat_grad[:] = 1.0                                   # not armed (bench.py): plain buckets, nothing carried
    real.loss_ready(torch.tensor([3.0]))
    for i in (3, 2, 1, 0):
        real.grad_ready(i)
    ddp_l.wait()
    assert ddp_l.carried_loss is None and torch.all(real.grad_tail == 0)
    acc2 = HipAccelerator(1, device="cpu")
    acc2.ddp = ddp_l
    ddp_l.carried_loss = torch.tensor(0.5)
    assert acc2.mean_loss(torch.tensor(9.0)).item() == 0.5 and ddp_l.carried_loss is None
    assert acc2.mean_loss(torch.tensor(float(rank))).item() == 0.5          # nothing carried: the gather of the reference

    # PEFT adapters under data parallel: the adapter set is the "model" HipDDP sees; its single bucket is reduced when
    # project() (the last thing the backward does) reports the gradients complete
    from yat_amd.lora import LoRAAdapters
    base = type("Base", (), {})()
    base.flat_param = torch.zeros(16 * 8 + 24 * 8, dtype=torch.bfloat16)
    base.flat_grad = torch.zeros_like(base.flat_param)
    base.P = {"blocks.0.to_q.weight": base.flat_param[:128].view(16, 8), "blocks.0.other.weight": base.flat_param[128:].view(24, 8)}
    ad = LoRAAdapters(base, ["to_q"], r=2, alpha=2.0)
    assert [e["module"] for e in ad.entries] == ["blocks.0.to_q"] and ad.num_parameters() == 2 * (8 + 16)
    ddp_ad = HipDDP(ad)
    ddp_ad.broadcast_parameters()                             # rank 0's kaiming draw everywhere
    gathered = [torch.zeros_like(ad.flat_param, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(gathered, ad.flat_param.float())
    assert torch.equal(gathered[0], gathered[1])
    ad.flat_grad[:] = float(rank + 1)
    ad.project()
    ddp_ad.wait()
    assert torch.all(ad.flat_grad.float() == 1.5)

    # bucket sampler consensus: both ranks must yield the same ratio at every step
    from tests.test_host_logic import _make_shards
    from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN
    from yat_amd.common.bucket_sampler import BucketSampler
    import pathlib
    paths = _make_shards(pathlib.Path(tmp) / f"r{rank}", 2, 30, seed=rank)
    model = type("M", (), {"aspect_ratios": ASPECT_RATIO_1024_BIN})()
    acc = HipAccelerator(1, device="cpu")
    it = iter(BucketSampler([], acc, batch_size=3, model=model, seed=5, local_paths=paths))
    for _ in range(8):
        b = next(it)
        r = torch.tensor([b.ratio])
        both = [torch.zeros(1), torch.zeros(1)]
        dist.all_gather(both, r)
        assert both[0].item() == both[1].item()
    dist.destroy_process_group()


def test_world2_gloo(tmp_path):
    for r in range(2):
        (tmp_path / f"r{r}").mkdir()
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _shard_worker(rank, world, port, tmp):
    """Sharded optimizer step (yat_amd/ddp.py shard_optimizer, yat_amd/optim.py _sharded_update) on the REAL model's flat layout,
    CPU + gloo: what each collective leaves where, who owns which norm pieces, and that 'update the own slice, all-gather the
    bucket' reproduces the replicated update on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytest
    from yat_amd.ddp import HipDDP
    from yat_amd.optim import NORM_PARTS, norm_pieces
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    real = SanaTransformer2DModelHIP(SanaConfig(num_layers=3, num_attention_heads=2, num_cross_attention_heads=2,
                                                cross_attention_head_dim=32, cross_attention_dim=64, caption_channels=96,
                                                in_channels=8, out_channels=8, sample_size=4), device="cpu")
    n = real.numel_flat
    # the layout: every bucket a whole number of 8 x 16-byte parts (yat_amd/flat.py SHARD_ALIGN), tensors still 16-byte aligned
    assert all((hi - lo) % 64 == 0 and hi > lo for lo, hi in real.bucket_bounds) and n % 64 == 0
    assert all(int(o) % 8 == 0 for o in real.seg_start)
    real.flat_param, real.flat_grad = real.flat_param.float(), real.flat_grad.float()       # (gloo has no bf16 reduction)
    ddp = HipDDP(real, shard_optimizer=True)
    assert real.shard is ddp.shard and (ddp.shard.rank, ddp.shard.world) == (rank, world) and real.loss_ready is None
    local = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank
    for step in range(2):
        real.flat_grad[:] = local
        for i in (3, 2, 1, 0):
            real.grad_ready(i)
        ddp.wait()
        mean = torch.arange(n, dtype=torch.float32) * 1.5 + 0.5
        for lo, hi in real.bucket_bounds:
            per = (hi - lo) // world
            for r in range(world):
                sl = slice(lo + r * per, lo + (r + 1) * per)
                # own slice: the mean over ranks; every other slice: still this rank's local values (RCCL's in-place form)
                assert torch.equal(real.flat_grad[sl], mean[sl] if r == rank else local[sl]), (step, lo, r)
    assert ddp.bytes_reduced == 2 * n * 4
    # norm pieces: never across an eighth of a bucket; the pieces a rank owns tile exactly its slices; all ranks together
    # tile the whole buffer once
    ps, tf, cb, mx, part = norm_pieces(real.seg_start.tolist(), real.bucket_bounds)
    assert ps[0] == 0 and ps[-1] == n and all(b > a for a, b in zip(ps, ps[1:])) and tf[-1] == len(ps) - 1
    per8 = NORM_PARTS // world
    owned = torch.zeros(n, dtype=torch.int32)
    for (a, b), (bi, k) in zip(zip(ps, ps[1:]), part):
        lo, hi = real.bucket_bounds[bi]
        e = (hi - lo) // NORM_PARTS
        assert k >= 0 and lo + k * e <= a and b <= lo + (k + 1) * e, "a piece straddles an eighth of its bucket"
        if rank * per8 <= k < (rank + 1) * per8:
            owned[a:b] += 1
    want = torch.zeros(n, dtype=torch.int32)
    for lo, hi in real.bucket_bounds:
        per = (hi - lo) // world
        want[lo + rank * per:lo + (rank + 1) * per] = 1
    assert torch.equal(owned, want)
    both = [torch.zeros_like(owned) for _ in range(world)]
    dist.all_gather(both, owned)
    assert torch.all(both[0] + both[1] == 1)
    # the update: every rank applies f to its slices only, the buckets are all-gathered -> f applied everywhere, on every rank
    real.flat_param[:] = torch.arange(n, dtype=torch.float32)
    f = lambda p, g_: p * 0.5 - g_                            # noqa: E731  (any elementwise update)
    for lo, hi in real.bucket_bounds:
        per = (hi - lo) // world
        sl = slice(lo + rank * per, lo + (rank + 1) * per)
        real.flat_param[sl] = f(real.flat_param[sl], real.flat_grad[sl])
        ddp.allgather_bulk(real.flat_param[lo:hi])
    assert torch.equal(real.flat_param, f(torch.arange(n, dtype=torch.float32), mean))
    # what cannot be sharded says so: 3 ranks, ragged buckets, coalesced buckets
    with pytest.raises(ValueError, match="not whole numbers"):
        HipDDP(type("M", (), {"flat_param": torch.zeros(100), "flat_grad": torch.zeros(100), "bucket_bounds": [(0, 100)],
                              "grad_ready": None})(), shard_optimizer=True)
    with pytest.raises(ValueError, match="coalesce"):
        HipDDP(real, shard_optimizer=True, coalesce=2)
    # the default stays the replicated step, and rebuilding the wrapper takes the shard descriptor off the model again
    assert HipDDP(real).shard is None and real.shard is None
    dist.destroy_process_group()


def test_world2_gloo_sharded_optimizer_collectives(tmp_path):
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


class _FakeCommLib:
    """The four entry points NativeComm's rendezvous touches; ``fail_on``: {rank: entry point that returns YAT_ENOCOMM}."""

    def __init__(self, rank, fail_on):
        self.calls, self._bad = [], fail_on.get(rank)

    def _rc(self, name):
        self.calls.append(name)
        return -2 if self._bad == name else 0

    def yat_comm_available(self):
        return self._rc("available")

    def yat_comm_unique_id(self, buf):
        return self._rc("unique_id")

    def yat_comm_init(self, rank, world, ident):
        assert ident is not None and len(ident) == 128
        return self._rc("init")

    def yat_comm_destroy(self):
        return self._rc("destroy")

    def yat_comm_last_error(self):
        return b"stub"


def _rendezvous_worker(rank, world, port):
    """Round-4 advisor: a rank that cannot build the native communicator must take every OTHER rank to the fallback with it,
    without a hang (rank 0 used to raise before the id broadcast its peers were blocked in) and without anybody entering the
    collective yat_comm_init alone."""
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from yat_amd import ddp
    for fail_on in ({0: "available"}, {1: "available"}, {0: "unique_id"}):
        fake = _FakeCommLib(rank, fail_on)
        try:
            ddp.NativeComm(None, lib=fake)
            raised = False
        except Exception:       # noqa: BLE001
            raised = True
        assert raised, f"rank {rank} built a communicator although {fail_on} failed"
        assert "init" not in fake.calls                       # nobody entered the collective
    fake = _FakeCommLib(rank, {})
    comm = ddp.NativeComm(None, lib=fake)                     # the healthy case: rank 0 drew the id, both initialised
    assert (comm.rank, comm.world) == (rank, 2) and fake.calls[-1] == "init"
    assert ("unique_id" in fake.calls) == (rank == 0)

    # HipDDP's agreement on the outcome (negotiate_native): a communicator that fails to build on ONE rank, in yat_comm_init
    # itself or anywhere else, leaves no rank on the native transport, and the one that was built is destroyed again
    class _Stub:
        destroyed = False

        def destroy(self):
            type(self).destroyed = True

    for bad_rank in (0, 1):
        _Stub.destroyed = False

        def factory(pg, bad_rank=bad_rank):
            if rank == bad_rank:
                raise RuntimeError("no librccl here")
            return _Stub()
        native, err = ddp.negotiate_native(None, None, factory)
        assert native is None
        assert (err is not None) == (rank == bad_rank) and _Stub.destroyed == (rank != bad_rank)
    native, err = ddp.negotiate_native(None, None, lambda pg: _Stub())
    assert isinstance(native, _Stub) and err is None
    assert ddp.agree(True) and not ddp.agree(rank == 0)
    dist.destroy_process_group()


def test_native_rendezvous_failure_on_one_rank_reaches_every_rank():
    mp.spawn(_rendezvous_worker, args=(2, _free_port()), nprocs=2, join=True)
