"""Pin of the data-parallel / gradient-accumulation semantics against the reference's OWN dependency (CPU, world 2, gloo).

The reference wraps its model with HuggingFace Accelerate (common/trainer.py:31-37 ``Accelerator(gradient_accumulation_steps)``,
:253 ``prepare``, :317 ``accumulate``, :344 ``backward``, :346-347 ``sync_gradients`` / ``clip_grad_norm_``, :359 ``gather``).
``accelerate`` IS installed in this container (1.14.0), and what it wraps -- torch ``DistributedDataParallel`` over gloo -- runs on
the CPU.  So, unlike the model math, this row can be held to the real thing: two processes train the same tiny MLP on different
data for two accumulation windows, once through ``accelerate.Accelerator`` + DDP exactly as the reference's loop drives them,
once through this build's ``HipAccelerator`` + ``HipDDP`` over a flat gradient buffer with two buckets; compared at every
micro-step: the ``sync_gradients`` cadence, the gradients after the reduction (mean over ranks of the accumulated, 1/k-scaled
micro-step gradients), and the gathered loss mean.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

accelerate = pytest.importorskip("accelerate")


class _Flat:
    """Two Linear layers whose parameters are views of ONE flat buffer (the layout yat_amd/flat.py gives the HIP models):
    gradients accumulate into views of ``flat_grad``; two buckets (layer 2 completes first in the backward)."""

    def __init__(self, ref: nn.Module):
        ps = list(ref.parameters())
        n = sum(p.numel() for p in ps)
        self.flat_param, self.flat_grad = torch.zeros(n), torch.zeros(n)
        self.params, o = [], 0
        for p in ps:
            v = nn.Parameter(self.flat_param[o:o + p.numel()].view_as(p))
            with torch.no_grad():
                v.copy_(p)
            v.grad = self.flat_grad[o:o + p.numel()].view_as(p)
            self.params.append(v)
            o += p.numel()
        cut = ps[0].numel() + ps[1].numel()
        self.bucket_bounds = [(0, cut), (cut, n)]
        self.grad_ready = None
        self.accumulate_grads = False

    def __call__(self, x):
        w1, b1, w2, b2 = self.params
        return torch.tanh(x @ w1.T + b1) @ w2.T + b2


def _data(rank, micro):
    g = torch.Generator().manual_seed(100 * rank + micro)
    return torch.randn(6, 5, generator=g), torch.randn(6, 3, generator=g)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      ACCELERATE_USE_CPU="true")
    from accelerate import Accelerator
    K = 2
    # ---- A: the reference's stack
    acc = Accelerator(cpu=True, gradient_accumulation_steps=K)
    assert acc.num_processes == world and dist.is_initialized() and dist.get_backend() == "gloo"
    torch.manual_seed(7)
    ref = nn.Sequential(nn.Linear(5, 8), nn.Tanh(), nn.Linear(8, 3))
    init = [p.detach().clone() for p in ref.parameters()]
    opt = torch.optim.SGD(ref.parameters(), lr=0.0)                    # the update itself is pinned elsewhere (clip + AdamW)
    model, opt = acc.prepare(ref, opt)
    rec_a = []
    for micro in range(2 * K):
        x, y = _data(rank, micro)
        with acc.accumulate(model):
            loss = nn.functional.mse_loss(model(x), y)
            acc.backward(loss)
            grads = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
            rec_a.append((bool(acc.sync_gradients), grads, acc.gather(loss.detach()).mean().item()))
            opt.step()
            opt.zero_grad()
    # ---- B: this build's accelerator + bucketed reduction over flat buffers, same weights, same data
    from yat_amd.common.trainer import HipAccelerator
    hacc = HipAccelerator(K, device="cpu")
    seed = nn.Sequential(nn.Linear(5, 8), nn.Tanh(), nn.Linear(8, 3))
    with torch.no_grad():
        for p, v in zip(seed.parameters(), init):
            p.copy_(v)
    flat = _Flat(seed)
    hacc.prepare(flat)
    assert hacc.ddp is not None and hacc.ddp.world == world
    rec_b = []
    for micro in range(2 * K):
        x, y = _data(rank, micro)
        with hacc.accumulate(flat):
            if not flat.accumulate_grads:
                flat.flat_grad.zero_()                                  # a backward that does not accumulate overwrites
            loss = nn.functional.mse_loss(flat(x), y)
            (loss / K).backward()                                       # the recipes' device path: backward inside optimize(),
            for i in (1, 0):                                            # buckets reported last layer first ...
                flat.grad_ready(i)
            loss.yat_backward_done = True
            hacc.backward(loss)                                         # ... and accelerator.backward only waits for them
            rec_b.append((bool(hacc.sync_gradients), flat.flat_grad.clone(), hacc.gather(loss.detach()).mean().item()))
    torch.save(dict(a=rec_a, b=rec_b), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulate_and_reduce_match_accelerate_and_torch_ddp(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in (0, 1)]
    for k in (0, 1):
        a, b = r[k]["a"], r[k]["b"]
        assert [x[0] for x in a] == [False, True, False, True] == [x[0] for x in b]      # sync_gradients cadence (:346)
        for micro, ((sa, ga, la), (sb, gb, lb)) in enumerate(zip(a, b)):
            assert abs(la - lb) <= 1e-6 * max(1.0, abs(la)), (k, micro, la, lb)           # gather(loss).mean() (:359)
            assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-7), (k, micro, (ga - gb).abs().max())
    # at the sync points both ranks hold the same, averaged gradient -- in both stacks
    for micro in (1, 3):
        assert torch.allclose(r[0]["a"][micro][1], r[1]["a"][micro][1], atol=1e-7)
        assert torch.equal(r[0]["b"][micro][1], r[1]["b"][micro][1])
