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


def _two_steps(transport):
    from yat_amd.ddp import HipDDP
    from yat_amd.optim import FlatAdamW
    from yat_amd.recipe import SanaRecipe
    model, cfg = _model()
    ddp = HipDDP(model, force=True, transport=transport) if transport else None
    if ddp:
        ddp.broadcast_parameters()
    opt = FlatAdamW(model, lr=1e-3, weight_decay=0.01, overlap_update=True)
    recipe = SanaRecipe(model, pad_to=32, device=DEV)
    g = torch.Generator().manual_seed(2)
    latents = (torch.randn(4, cfg.in_channels, 8, 12, generator=g) * 0.5).to(BF)
    embs = [torch.randn(L, cfg.caption_channels, generator=g).to(BF) for L in (5, 32, 17, 9)]
    grads = []
    for s in range(2):
        loss = recipe.optimize(latents, embs, torch.Generator().manual_seed(10 + s))
        loss.backward()
        if ddp:
            ddp.wait()
        grads.append(model.flat_grad.clone())
        opt.step()
    model.join_pending_update()
    torch.cuda.synchronize()
    return grads, model.flat_param.clone(), (ddp.bytes_reduced if ddp else 0), model.numel_flat


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


def test_forced_ddp_torch_transport_is_bit_identical(one_rank_group):
    g0, p0, _, n = _two_steps(None)
    g1, p1, reduced, _ = _two_steps("torch")
    assert reduced == 2 * 2 * n, (reduced, n)                     # every bucket of both steps went through RCCL
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
    NativeComm.get().destroy()
    assert lib.yat_comm_world() == 0


def _world2_worker(rank, world, port, out_dir):
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
    ddp = HipDDP(model)
    ddp.broadcast_parameters()
    opt = FlatAdamW(model, lr=1e-3, weight_decay=0.01, overlap_update=True)
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
    torch.cuda.synchronize()
    torch.save(dict(grads=grads, param=model.flat_param.cpu(), replays=getattr(model, "plan_replays", 0)),
               os.path.join(out_dir, f"rank{rank}.pt"))
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
