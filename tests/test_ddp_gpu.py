"""The data-parallel machinery on ONE GPU: a forced one-rank RCCL group drives the real model's bucket order, the side-stream
`grad_ready` hooks, the communication stream and the optimizer's wait -- everything of the N > 1 path except a second rank
(common/trainer.py:31-37,253,344 is what it replaces).  A one-rank mean is the identity, so gradients and parameters must be
bit-identical to the plain step.  Both transports: torch.distributed's process group and the library's own communicator
behind the C ABI (yat_comm_* / yat_bucket_allreduce_async, include/yat_hip.h)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def _model():
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    cfg = SanaConfig(num_layers=3, num_attention_heads=4, attention_head_dim=32, num_cross_attention_heads=2,
                     cross_attention_head_dim=64, cross_attention_dim=128, caption_channels=96, in_channels=8, out_channels=8,
                     sample_size=8)
    return SanaTransformer2DModelHIP(cfg, device=DEV).init_synthetic(3), cfg


# Every model family of BASELINE's "DP = 8" configs puts its own `grad_ready` call sites under the data-parallel hook
# (common/trainer.py:212-241,253: accelerator.prepare wraps whatever model -- or PEFT-wrapped model -- the recipe built):
#   sana    config 2   yat_amd/sana.py     per-block hooks on the weight-gradient stream + embedders
#   pixart  config 3   yat_amd/pixart.py   (same structure, its own call sites)
#   sd3     config 4   yat_amd/sd3.py      MMDiT: hooks behind the modulation-gradient chain of each block
#   lokr    config 5   yat_amd/lokr.py     one small bucket after project(), B = 32 per GPU
#   lora               yat_amd/lora.py     (the reference's other adapter branch, :214-219)
KINDS = ["sana", "pixart", "sd3", "lokr", "lora"]
ADAPTER_TARGETS = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]


def _case(kind, seed=2):
    """-> (base model, trained object (model or adapter set), step(i) -> loss with gradients in trained.flat_grad)."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(1000 + seed)           # the adapters' own init draws from the global CPU RNG (peft's kaiming_uniform_)
    if kind in ("sana", "lokr", "lora"):
        from yat_amd.recipe import SanaRecipe
        model, cfg = _model()
        trained = model
        B = 4
        if kind == "lokr":
            from yat_amd.lokr import LoKrAdapters
            trained, B = LoKrAdapters(model, ADAPTER_TARGETS, r=2, alpha=4.0, module_dropout=0.0), 32      # config 5: B = 32
        elif kind == "lora":
            from yat_amd.lora import LoRAAdapters
            trained, B = LoRAAdapters(model, ADAPTER_TARGETS, r=2, alpha=4.0), 32
        if kind == "lokr":                   # adapters start as the identity: move the zero factor off it (w1 / lora_B)
            for e in trained.entries:
                w1, _, _ = trained._views(e, trained.flat_param)
                w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
        elif kind == "lora":
            for e in trained.entries:
                _, bt = trained._views(e, trained.flat_param)
                bt[:2].copy_((torch.randn(2, e["out"], generator=g) * 0.05).to(BF))
        recipe = SanaRecipe(model, pad_to=32, device=DEV)
        latents = (torch.randn(B, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
        embs = [torch.randn(int(L), cfg.caption_channels, generator=g).to(BF)
                for L in torch.randint(1, 33, (B,), generator=g).tolist()]

        def step(i):
            loss = recipe.optimize(latents, embs, torch.Generator().manual_seed(10 + i))
            loss.backward()
            return loss
        return model, trained, step
    if kind == "pixart":
        from yat_amd.pixart import PixArtConfig, PixArtTransformer2DModelHIP
        from yat_amd.recipe import PixArtRecipe
        cfg = PixArtConfig(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=3,
                           cross_attention_dim=48, sample_size=8, patch_size=2, caption_channels=64)
        model = PixArtTransformer2DModelHIP(cfg, device=DEV).init_synthetic(3)
        recipe = PixArtRecipe(model, pad_to=16, device=DEV)
        latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
        embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (5, 16, 9, 1)]

        def step(i):
            loss = recipe.optimize(latents, embs, torch.Generator().manual_seed(10 + i))
            loss.backward()
            return loss
        return model, model, step
    if kind == "sd3":
        from yat_amd.recipe import SD3Recipe
        from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP
        cfg = SD3Config(sample_size=16, patch_size=2, in_channels=8, out_channels=8, num_layers=3, attention_head_dim=64,
                        num_attention_heads=2, joint_attention_dim=96, caption_projection_dim=128, pooled_projection_dim=64,
                        pos_embed_max_size=24, dual_attention_layers=(0, 1))
        model = SD3Transformer2DModelHIP(cfg, device=DEV).init_synthetic(3)
        recipe = SD3Recipe(model, device=DEV)
        latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
        emb = (torch.randn(4, 10, cfg.joint_attention_dim, generator=g).to(BF),
               torch.randn(4, cfg.pooled_projection_dim, generator=g).to(BF))

        def step(i):
            loss = recipe.optimize(latents, emb, torch.Generator().manual_seed(10 + i))
            loss.backward()
            return loss
        return model, model, step
    raise ValueError(kind)


def _two_steps(transport, kind="sana", shard=False, use_ema=False):
    from yat_amd.ddp import HipDDP
    from yat_amd.optim import FlatAdamW
    model, trained, step = _case(kind)
    fired = []
    ddp = HipDDP(trained, force=True, transport=transport, shard_optimizer=shard) if transport else None
    assert ddp is None or (ddp.shard is not None) == shard
    if ddp:
        ddp.broadcast_parameters()
        hook = trained.grad_ready
        trained.grad_ready = lambda i: (fired.append(i), hook(i))[1]          # which call sites fired under the group
    opt = FlatAdamW(trained, lr=1e-3, weight_decay=0.01, overlap_update=True, use_ema=use_ema)
    grads = []
    for s in range(2):
        step(s)
        if ddp:
            ddp.wait()
        grads.append(trained.flat_grad.clone())
        opt.step()
    trained.join_pending_update()
    if use_ema:
        assert opt.gather_ema() == shard
        _two_steps.ema = opt.ema_shadow.clone()
    torch.cuda.synchronize()
    assert torch.isfinite(grads[0].float()).all() and grads[0].float().abs().max().item() > 0
    if ddp:
        assert sorted(set(fired)) == list(range(len(trained.bucket_bounds))), (kind, fired)     # every bucket's hook fired
        assert len(fired) == 2 * len(trained.bucket_bounds)
    return grads, trained.flat_param.clone(), (ddp.bytes_reduced if ddp else 0), trained.numel_flat


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", KINDS)
def test_forced_ddp_torch_transport_is_bit_identical(one_rank_group, kind):
    g0, p0, _, n = _two_steps(None, kind)
    g1, p1, reduced, _ = _two_steps("torch", kind)
    assert reduced == 2 * 2 * n, (reduced, n)                     # every bucket of both steps went through RCCL
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("kind", KINDS[1:])
def test_forced_ddp_native_transport_all_families(kind):
    """The library's own communicator under the hooks of PixArt-Sigma, the MMDiT and the adapter sets (configs 3 - 5)."""
    from yat_amd.ddp import NativeComm
    g0, p0, _, n = _two_steps(None, kind)
    try:
        g1, p1, reduced, _ = _two_steps("native", kind)
    finally:
        if NativeComm._instance is not None:
            NativeComm.get().destroy()
    assert reduced == 2 * 2 * n, (reduced, n)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)


def test_forced_ddp_native_transport_is_bit_identical():
    """yat_comm_unique_id -> yat_comm_init(0, 1, id) -> yat_comm_broadcast -> per bucket yat_bucket_allreduce_async on the
    communication stream -> yat_comm_wait(-1, compute stream) -> yat_comm_destroy, all through the C ABI."""
    from yat_amd import lib as L
    from yat_amd.ddp import NativeComm
    lib = L.load()
    assert lib.yat_comm_world() == 0 and lib.yat_comm_rank() == -1
    assert lib.yat_comm_wait(-1, None) == -2                       # YAT_ENOCOMM before init
    g0, p0, _, n = _two_steps(None)
    g1, p1, reduced, _ = _two_steps("native")
    assert lib.yat_comm_world() == 1 and lib.yat_comm_rank() == 0
    assert reduced == 2 * 2 * n, (reduced, n)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)
    # argument checks of the transport
    x = torch.ones(64, dtype=BF, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.yat_bucket_allreduce_async(x.data_ptr(), 128, 256, s, s) == -1      # bucket id out of range
    assert lib.yat_bucket_allreduce_async(x.data_ptr(), 127, 0, s, s) == -1        # odd byte count (bf16 elements)
    assert lib.yat_comm_broadcast(x.data_ptr(), 128, 1, s) == -1                   # root outside the group
    assert lib.yat_bucket_allreduce_async(x.data_ptr(), 128, 7, s, s) == 0
    assert lib.yat_comm_wait(7, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, torch.ones_like(x))
    # every other bulk collective rides the same communicator (round-4 review: the EMA mean before validation,
    # common/trainer.py:374-377, went through the gloo rendezvous group): yat_comm_allreduce, fp32 and bf16, mean and sum --
    # over one rank both are the identity -- and HipDDP.allreduce_bulk picks it whenever the native transport is in use
    from yat_amd.ddp import HipDDP
    ema = torch.randn(1 << 20, device=DEV)
    ema_ref = ema.clone()
    assert lib.yat_comm_allreduce(ema.data_ptr(), ema.numel(), 1, 0, s) == 0
    h = torch.randn(4096, device=DEV).to(BF)
    h_ref = h.clone()
    NativeComm.get().allreduce(h, mean=False)
    _, trained, _ = _case("sana")
    ddp = HipDDP(trained, force=True, transport="native")
    assert ddp.native is NativeComm.get()
    shadow = trained.flat_param.float()
    shadow_ref = shadow.clone()
    assert ddp.allreduce_bulk(shadow, mean=True) is shadow
    torch.cuda.synchronize()
    assert torch.equal(ema, ema_ref) and torch.equal(h, h_ref) and torch.equal(shadow, shadow_ref)
    assert lib.yat_comm_allreduce(ema.data_ptr(), 0, 1, 0, s) == -1                # empty
    assert lib.yat_comm_allreduce(ema.data_ptr(), 8, 2, 0, s) == -1                # unknown dtype
    assert lib.yat_comm_allreduce(ema.data_ptr(), 8, 1, 2, s) == -1                # unknown reduction
    with pytest.raises(ValueError):
        NativeComm.get().allreduce(torch.ones(8, dtype=torch.float16, device=DEV))
    NativeComm.get().destroy()
    assert lib.yat_comm_world() == 0
    assert lib.yat_comm_allreduce(ema.data_ptr(), 8, 1, 0, s) == -2                # YAT_ENOCOMM after destroy


@pytest.mark.parametrize("kind", KINDS[:3])
def test_forced_ddp_sharded_optimizer_torch_transport(one_rank_group, kind):
    """HipDDP(shard_optimizer=True) (round-5 review item 4; common/trainer.py:246-253,347-348 is what it replaces): every
    bucket reduce-scattered, the clip norm over owned pieces + one small all-reduce, AdamW (+ EMA) on this rank's slice of
    every bucket, the bucket's parameters all-gathered on their own stream under `param_events` -- with one forced rank the
    whole machinery runs and must leave the plain step's gradients, parameters and EMA shadow, bit for bit."""
    g0, p0, _, n = _two_steps(None, kind, use_ema=True)
    e0 = _two_steps.ema
    g1, p1, reduced, _ = _two_steps("torch", kind, shard=True, use_ema=True)
    assert reduced == 2 * 2 * n
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1) and torch.equal(e0, _two_steps.ema)


def test_forced_ddp_sharded_optimizer_native_transport():
    """The same through the library's communicator: yat_bucket_reduce_scatter_async / yat_comm_allgather / yat_comm_allreduce."""
    from yat_amd import lib as L
    from yat_amd.ddp import NativeComm
    g0, p0, _, n = _two_steps(None, "sana", use_ema=True)
    e0 = _two_steps.ema
    try:
        g1, p1, reduced, _ = _two_steps("native", "sana", shard=True, use_ema=True)
        assert reduced == 2 * 2 * n
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)
        assert torch.equal(p0, p1) and torch.equal(e0, _two_steps.ema)
        lib = L.load()
        x = torch.arange(64, dtype=torch.float32, device=DEV).to(BF)
        s = torch.cuda.current_stream().cuda_stream
        assert lib.yat_bucket_reduce_scatter_async(x.data_ptr(), 120, 0, s, s) == -1     # not a whole number of 16-byte slices
        assert lib.yat_comm_allgather(x.data_ptr(), 120, s) == -1
        assert lib.yat_bucket_reduce_scatter_async(x.data_ptr(), 128, 3, s, s) == 0 and lib.yat_comm_wait(3, s) == 0
        assert lib.yat_comm_allgather(x.data_ptr(), 128, s) == 0
        torch.cuda.synchronize()
        assert torch.equal(x, torch.arange(64, dtype=torch.float32, device=DEV).to(BF))    # one rank: both are the identity
    finally:
        if NativeComm._instance is not None:
            NativeComm.get().destroy()
    assert L.load().yat_comm_allgather(x.data_ptr(), 128, s) == -2                          # YAT_ENOCOMM after destroy


def test_adapter_sets_refuse_the_sharded_step(one_rank_group):
    """An adapter set's single small bucket is not a whole number of 8 x 16-byte parts (and has nothing to gain): it says so."""
    from yat_amd.ddp import HipDDP
    _, trained, _ = _case("lokr")
    if all((hi - lo) % 64 == 0 for lo, hi in trained.bucket_bounds):
        pytest.skip("this adapter set happens to be shardable")
    with pytest.raises(ValueError, match="replicated optimizer"):
        HipDDP(trained, force=True, transport="torch", shard_optimizer=True)


def test_native_transport_failure_falls_back_to_the_process_group(one_rank_group, monkeypatch):
    """HipDDP builds the library's communicator, the ranks agree on the outcome, and if it failed anywhere every rank uses
    torch.distributed's RCCL group instead (yat_amd/ddp.py): same gradients and parameters as the plain step, a warning says so."""
    import warnings
    from yat_amd import ddp as D

    def boom(cls, process_group=None):
        raise RuntimeError("no librccl on this rank")
    monkeypatch.setattr(D.NativeComm, "get", classmethod(boom))
    g0, p0, _, n = _two_steps(None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        g1, p1, reduced, _ = _two_steps("native")
    assert any("falling back" in str(x.message) for x in w)
    assert reduced == 2 * 2 * n
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)


def _world2_worker(rank, world, port, out_dir, shard=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      YAT_DIST_BACKEND="gloo")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yat_amd.ddp import HipDDP
    from yat_amd.optim import FlatAdamW
    from yat_amd.recipe import SanaRecipe
    model, cfg = _model()
    if rank == 1:                                  # broadcast must overwrite this
        model.flat_param.add_(1.0)
    ddp = HipDDP(model, shard_optimizer=shard)
    assert (ddp.shard is not None) == shard
    ddp.broadcast_parameters()
    opt = FlatAdamW(model, lr=1e-3, weight_decay=0.01, overlap_update=True, use_ema=shard is not None)
    recipe = SanaRecipe(model, pad_to=32, device=DEV)
    g = torch.Generator().manual_seed(20 + rank)   # every rank its own batch
    grads = []
    for s in range(3):                             # step 0 records the launch plans' first versions, 1-2 replay (DDP hooks inside)
        latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
        embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (5, 32, 17, 9)]
        recipe.optimize_device(latents, embs, torch.Generator().manual_seed(10 + s))
        ddp.wait()
        torch.cuda.synchronize()
        grads.append(model.flat_grad.clone().cpu())
        opt.step()
    model.join_pending_update()
    opt.gather_ema()
    torch.cuda.synchronize()
    torch.save(dict(grads=grads, param=model.flat_param.cpu(), replays=getattr(model, "plan_replays", 0),
                    ema=opt.ema_shadow.cpu(), norm=opt.grad_norm.cpu(), coef=opt.clip_coef.cpu()),
               os.path.join(out_dir, f"rank{rank}{'_shard' if shard else ''}.pt"))
    dist.destroy_process_group()


def test_world2_over_gloo_matches_two_single_rank_backwards(tmp_path):
    """Two processes on this one GPU (gloo transport: RCCL refuses two ranks on one device), each training its own batch on
    the device path with launch plans on: after the bucketed mean every rank must hold the SAME gradient -- the mean of the two
    single-rank gradients -- and, after the optimizer, the same parameters.  The single-rank gradients are recomputed here."""
    import socket
    import torch.multiprocessing as mp
    from yat_amd.recipe import SanaRecipe
    from yat_amd.optim import FlatAdamW
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_world2_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    assert r0["replays"] >= 2 and r1["replays"] >= 2, "the launch plans were never replayed"
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b), "ranks disagree on the reduced gradient"
    assert torch.equal(r0["param"], r1["param"])
    # reference for step 0: the two local gradients from identical initial weights, averaged
    local = []
    for rank in (0, 1):
        model, cfg = _model()
        recipe = SanaRecipe(model, pad_to=32, device=DEV)
        g = torch.Generator().manual_seed(20 + rank)
        latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
        embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (5, 32, 17, 9)]
        recipe.optimize_device(latents, embs, torch.Generator().manual_seed(10))
        torch.cuda.synchronize()
        local.append(model.flat_grad.float().cpu())
    mean = (local[0] + local[1]) / 2
    err = ((r0["grads"][0].float() - mean).norm() / mean.norm()).item()
    print(f"[parity] world-2 mean gradient vs the two single-rank gradients: rel_l2={err:.3e}")
    assert err <= 6e-3


def test_world2_over_gloo_sharded_step_equals_the_replicated_step(tmp_path):
    """Two ranks on this one GPU, three steps each way: the replicated step (all-reduce, AdamW over everything on both ranks) and
    the sharded step (reduce-scatter, clip norm from owned pieces + one all-reduce of the partial sums, AdamW + EMA on half of
    every bucket, all-gather) end with the SAME clip coefficient, parameters and EMA shadow on both ranks, bit for bit (two-term
    sums commute, so the reduce-scatter's means are the all-reduce's; with more ranks the collective's own summation order
    decides and only 'same reduced gradients -> same parameters' is promised)."""
    import socket
    import torch.multiprocessing as mp
    out = {}
    for shard in (False, True):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_world2_worker, args=(2, port, str(tmp_path), shard), nprocs=2, join=True)
        out[shard] = [torch.load(tmp_path / f"rank{r}{'_shard' if shard else ''}.pt") for r in (0, 1)]
    rep, sh = out[False], out[True]
    assert sh[0]["replays"] >= 2, "the launch plans were never replayed under the sharded step"
    for r in (0, 1):
        assert torch.equal(sh[r]["norm"], rep[r]["norm"]) and torch.equal(sh[r]["coef"], rep[r]["coef"])
        assert torch.equal(sh[r]["param"], rep[0]["param"]), f"rank {r}: sharded parameters differ from the replicated step's"
        assert torch.equal(sh[r]["ema"], rep[0]["ema"])
    # and the sharded gradients are reduced only where a rank owns them: rank 0's first half of a bucket is the mean
    g_rep, g0, g1 = rep[0]["grads"][2], sh[0]["grads"][2], sh[1]["grads"][2]
    from yat_amd.sana import SanaConfig  # noqa: F401  (layout only)
    model, _ = _model()
    for lo, hi in model.bucket_bounds:
        per = (hi - lo) // 2
        assert torch.equal(g0[lo:lo + per], g_rep[lo:lo + per]) and torch.equal(g1[lo + per:hi], g_rep[lo + per:hi])
    assert not torch.equal(g0, g_rep)


def _world2_adapter_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      YAT_DIST_BACKEND="gloo")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yat_amd.ddp import HipDDP
    from yat_amd.optim import FlatAdamW
    model, trained, step = _case("lokr", seed=20 + rank)      # every rank its own batch AND its own adapter draw ...
    ddp = HipDDP(trained)
    ddp.broadcast_parameters()                                # ... which rank 0's must overwrite (trainer.py:253)
    start = trained.flat_param.clone().cpu()
    opt = FlatAdamW(trained, lr=1e-3, weight_decay=0.01, overlap_update=True)
    grads = []
    for s in range(2):
        step(s)
        ddp.wait()
        torch.cuda.synchronize()
        grads.append(trained.flat_grad.clone().cpu())
        opt.step()
    trained.join_pending_update()
    torch.cuda.synchronize()
    torch.save(dict(grads=grads, start=start, param=trained.flat_param.cpu(), reduced=ddp.bytes_reduced),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_world2_over_gloo_adapter_step(tmp_path):
    """BASELINE config 5's data-parallel clause with a real second rank: a LoKr adapter set (B = 32 per rank) under HipDDP,
    two processes on this GPU over gloo.  The adapter parameters are rank 0's after the broadcast, the single small bucket is
    reduced once per step (after ``project()``), both ranks hold the same gradient -- the mean of the two single-rank
    gradients, recomputed here from the broadcast parameters -- and the same parameters after the optimizer."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_world2_adapter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    assert torch.equal(r0["start"], r1["start"]), "broadcast did not overwrite rank 1's adapter parameters"
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b), "ranks disagree on the reduced adapter gradient"
    assert torch.equal(r0["param"], r1["param"]) and not torch.equal(r0["param"], r0["start"])
    assert r0["reduced"] == 2 * 2 * r0["start"].numel()        # one bucket of numel bf16 elements per step
    local = []
    for rank in (0, 1):
        _, trained, step = _case("lokr", seed=20 + rank)
        with torch.no_grad():
            trained.flat_param.copy_(r0["start"].to(DEV))
        step(0)
        torch.cuda.synchronize()
        local.append(trained.flat_grad.float().cpu())
    mean = (local[0] + local[1]) / 2
    err = ((r0["grads"][0].float() - mean).norm() / mean.norm()).item()
    print(f"[parity] world-2 LoKr adapter gradient vs the mean of the two single-rank gradients: rel_l2={err:.3e}")
    assert err <= 6e-3


@pytest.mark.parametrize("transport", ["torch", "native"])
def test_ddp_diagnostic_modes_do_not_change_results(one_rank_group, transport):
    """bench.py's `comm` object is measured with HipDDP.timing (HIP events around every bucket's collective and around the
    optimizer's wait) and HipDDP.dryrun (everything but the collective): under a one-rank group both must leave gradients and
    parameters bit-identical to the plain step, time every bucket once per step and clear their bookkeeping."""
    from yat_amd.ddp import HipDDP, NativeComm
    from yat_amd.optim import FlatAdamW
    g0, p0, _, n = _two_steps(None, "sana")
    model, trained, step = _case("sana")
    ddp = HipDDP(trained, force=True, transport=transport)
    try:
        ddp.broadcast_parameters()
        opt = FlatAdamW(trained, lr=1e-3, weight_decay=0.01, overlap_update=True)
        grads = []
        for s, (timing, dry) in enumerate([(True, False), (False, True)]):
            ddp.timing, ddp.dryrun = timing, dry
            step(s)
            ddp.wait()
            grads.append(trained.flat_grad.clone())
            opt.step()
        trained.join_pending_update()
        torch.cuda.synchronize()
        nb = len(trained.bucket_bounds)
        assert len(ddp.timed_buckets) == nb and len(ddp.timed_waits) == 1
        assert sum(b for _, b, _, _ in ddp.timed_buckets) == 2 * n
        assert all(e0.elapsed_time(e1) >= 0.0 for _, _, e0, e1 in ddp.timed_buckets)
        assert ddp.timed_waits[0][0].elapsed_time(ddp.timed_waits[0][1]) >= 0.0
        for a, b in zip(g0, grads):
            assert torch.equal(a, b)
        assert torch.equal(p0, trained.flat_param)
    finally:
        if NativeComm._instance is not None:
            NativeComm.get().destroy()


@pytest.mark.parametrize("transport", ["torch", "native"])
def test_logged_loss_rides_with_the_top_bucket(one_rank_group, transport):
    """common/trainer.py:359 gathers the averaged loss with a collective of its own every step; here the device path reports
    its loss before the backward (recipe._report_loss -> HipDDP.on_loss) and it travels in the spare element behind the
    gradients with the first bucket the backward completes.  Armed per micro-step (track_loss); unarmed steps (bench.py) are
    byte-for-byte the plain buckets."""
    from yat_amd.ddp import HipDDP, NativeComm
    from yat_amd.flat import GRAD_TAIL
    from yat_amd.recipe import SanaRecipe
    model, cfg = _model()
    recipe = SanaRecipe(model, pad_to=32, device=DEV)
    g = torch.Generator().manual_seed(5)
    latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (7, 32, 1, 19)]
    ddp = HipDDP(model, force=True, transport=transport)
    try:
        n = model.numel_flat
        recipe.optimize_device(latents, embs, torch.Generator().manual_seed(1))
        ddp.wait()
        torch.cuda.synchronize()
        assert ddp.carried_loss is None and ddp.bytes_reduced == 2 * n
        plain = model.flat_grad.clone()
        ddp.track_loss(torch.tensor(0.5, device=DEV))                  # the window's earlier micro-steps
        loss = recipe.optimize_device(latents, embs, torch.Generator().manual_seed(1))
        ddp.wait()
        torch.cuda.synchronize()
        want = loss.float().item() + 0.5
        # two bf16 slots (head + remainder of the difference to the previous step's mean): 16 bits of the value survive
        assert abs(ddp.carried_loss.item() - want) <= 2e-5 * abs(want), (ddp.carried_loss.item(), want)
        assert torch.equal(model.flat_grad, plain)                     # same draws, same gradients: the passenger disturbs nothing
        assert ddp.bytes_reduced == 2 * (2 * n + GRAD_TAIL) and torch.all(model.grad_tail == 0)
        ddp.carried_loss = None
        recipe.optimize_device(latents, embs, torch.Generator().manual_seed(1))      # armed for ONE micro-step only
        ddp.wait()
        torch.cuda.synchronize()
        assert ddp.carried_loss is None
        ddp.track_loss(torch.tensor(0.25, device=DEV))                 # next armed step: the offset is the last carried mean
        loss = recipe.optimize_device(latents, embs, torch.Generator().manual_seed(1))
        ddp.wait()
        torch.cuda.synchronize()
        want = loss.float().item() + 0.25
        assert abs(ddp.carried_loss.item() - want) <= 2e-6 * abs(want), (ddp.carried_loss.item(), want)
    finally:
        if NativeComm._instance is not None:
            NativeComm.get().destroy()
