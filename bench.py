#!/usr/bin/env python3
"""Headline benchmark: SANA-1.6B 1024 px bf16 training step, images/sec (whole job), on N MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (no launcher around it: starts its own ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = the whole hot path of common/trainer.py:312-356 x train_sana.py:163-219 on one batch of synthetic
cached features: pad/mask of the ragged text embeddings, noise + logit-normal timestep draw, flow-matching
mix, SANA-1.6B forward, fp32 MSE, backward, (bucketed RCCL gradient all-reduce overlapped with backward when
N > 1), global-norm clip and AdamW.  B = 8 images per GPU (BASELINE config 2), weak scaling.  Inputs (latents and
the concatenated text embeddings) are resident in HBM before the timed region; weights are random-init of the
1.6 B architecture (no checkpoints offline).

The JSON line carries, besides the driver's contract:
  roofline      dominant kernel = the MFMA GEMM family.  achieved = algorithmic FLOPs of every GEMM launch in the
                timed region (2*M*N*K each) / the summed HIP-event durations of those launches, recorded live on the
                launch stream; peak = 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).
  mfma_util_step  whole-step figure: algorithmic training FLOPs (6 x MAC, SURVEY.md 8d: 9.285 TFLOP/image at
                N=1024,T=512) x img/s/GPU / 2.5e15.
  comm          (N > 1, or YAT_DDP_FORCE=1 on one GPU) the data-parallel exchange: transport, buckets and bytes reduced per
                step, per-bucket collective time from HIP events on the communication stream, the step with the collective
                switched off on the same inputs (exposed time = the difference), how long the optimizer waited, overlap
                fraction, the RCCL channel cap in force (--rccl-channels N sets NCCL_MAX_NCHANNELS); with --lokr / --lora the
                latency of the adapter set's single small all-reduce.
  cpu_baseline  the CPU oracle (torch restatement of the reference path; diffusers is absent offline) timed on this
                box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0
# latent-grid buckets (h, w) = ASPECT_RATIO_1024_BIN / 32, all ~1024 tokens
BUCKETS = [(32, 32), (16, 64), (24, 42), (44, 22)]


def train_flops_per_image(cfg, N, T):
    """Algorithmic training FLOPs of one image: 6 x forward MACs (fwd + dgrad + wgrad), no recompute.  T = text rows that
    are computed for the image: 512 when the padding rows go through the text side as in the reference, the prompt's own
    length (batch mean) when the text rows are packed -- the skipped rows are not counted as work."""
    D, Hc, Cc = cfg.inner_dim, cfg.ffn_hidden, cfg.caption_channels
    blk = (N * D * 3 * D + N * D * D            # qkv, attn1.out
           + 2 * 33 * 32 * N * cfg.num_attention_heads   # linear attention state + apply
           + N * D * D + T * D * 2 * D + N * D * D       # attn2 q, kv, out
           + 2 * N * T * D                                # QK^T and PV
           + N * D * 2 * Hc + 9 * 2 * Hc * N + N * Hc * D)   # conv_inverted, depthwise, conv_point
    head = (N * cfg.in_channels * D + 256 * D + D * D + D * 6 * D + T * Cc * D + T * D * D + N * D * cfg.out_channels)
    return 6.0 * (cfg.num_layers * blk + head)


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may actually use (yat_amd/common/host.py: min(affinity mask, cgroup cpu quota))."""
    from yat_amd.common.host import usable_cores as f
    return f()


def cpu_baseline(state_dict, cfg_layers=20):
    """the CPU restatement of the reference step (oracle/, stock torch ops + torch.optim.AdamW, bf16
    parameters, no checkpointing) timed for real on this box's host cores -- the FULL SANA-1.6B stack (20 blocks), B=1,
    N=1024, T=512, forward + backward + clip + AdamW, 1 warm-up + 2 timed steps, all usable cores.  The weights are the HIP
    model's (one D2H copy; drawing 1.6 B normals on the CPU would take longer than the measurement)."""
    from oracle.sana_ref import SanaConfig as RefCfg, SanaTransformerRef
    from oracle.recipe_ref import FlowMatchSchedule, optimize_ref, clip_and_adamw_step
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host cores (os.cpu_count()={os.cpu_count()})")
    with torch.device("meta"):
        m = SanaTransformerRef(RefCfg(num_layers=cfg_layers))
    m = m.to(torch.bfloat16).to_empty(device="cpu")
    m.load_state_dict(state_dict)
    del state_dict
    g = torch.Generator().manual_seed(1234)
    lat = (torch.randn(1, 32, 32, 32, generator=g) * 0.5).to(torch.bfloat16)
    embs = [torch.randn(160, 2304, generator=g).to(torch.bfloat16)]
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5)
    sched = FlowMatchSchedule()
    times = []
    for it in range(3):                          # 1 warm-up + 2 timed
        t0 = time.perf_counter()
        loss, _, _ = optimize_ref(m, sched, lat, embs, torch.Generator(), 512, torch.bfloat16)
        loss.backward()
        clip_and_adamw_step(list(m.parameters()), opt)
        times.append(time.perf_counter() - t0)
        log(f"cpu_baseline: step {it} ({'warm-up' if it == 0 else 'timed'}) {times[-1]:.2f}s loss={loss.item():.4f}")
        if it == 1 and times[0] + times[1] > 60.0:      # keep the default run within minutes on a slow host
            break
    timed = times[1:]
    t_step = sum(timed) / len(timed)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": 1.0 / t_step, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": (f"CPU restatement of the reference path (oracle/: stock torch ops + torch.optim.AdamW; the literal "
                       f"Accelerate+diffusers stack is absent offline), full SANA-1.6B ({cfg_layers} blocks), bf16, B=1, N=1024, "
                       f"T=512, fwd+bwd+clip+AdamW, 1 warm-up ({times[0]:.2f}s) + {len(timed)} timed steps "
                       f"({', '.join(f'{t:.2f}s' for t in timed)}), measured not extrapolated"),
            "cpu_model": cpu}


def run_from_shards(args, rank, world, local_rank):
    """``--data shards``: the benchmark IS the trainer.  Synthetic cached-feature shards (the reference's {ratio, latent.pt,
    emb.pt} tar format, same shapes and text-length distribution as the resident mode) are written to local disk, then
    ``train_sana.SanaModel(params).run()`` -- sampler with its decode thread, bucket consensus, one packed H2D per batch, the
    allocation-free device step, clip + AdamW, loss logging -- runs W + K steps; the clock brackets the last K.  Validation /
    checkpoint writes are switched off (they are not the step); everything else is the code path a user's training run takes."""
    import tempfile
    import torch.distributed as dist
    from train_sana import SanaModel
    from yat_amd.common.shards import write_shard
    from yat_amd.common.training_parameters_reader import TrainingParameters
    from yat_amd.sana import SanaConfig
    B = args.batch
    root = os.path.join(tempfile.gettempdir(), f"yat_bench_shards_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}")
    os.makedirs(root, exist_ok=True)
    ratios = {(32, 32): "1.0", (16, 64): "0.25", (24, 42): "0.57", (44, 22): "2.0"}      # ASPECT_RATIO_1024_BIN keys
    g = torch.Generator().manual_seed(1234 + rank)
    per_bucket = max(2 * B, 16)
    samples = []
    for i in range(per_bucket * len(BUCKETS)):
        h, w = BUCKETS[i % len(BUCKETS)]
        L = int(torch.randint(20, 301, (1,), generator=g))
        samples.append(dict(__key__=f"{rank:02d}{i:06d}", ratio=ratios[(h, w)],
                            latent=(torch.randn(32, h, w, generator=g) * 0.5).to(torch.bfloat16),
                            emb=torch.randn(L, 2304, generator=g).to(torch.bfloat16)))
    write_shard(os.path.join(root, f"shard-{rank:06d}.tar"), samples)
    del samples
    if world > 1:
        dist.barrier()
    paths = [os.path.join(root, f"shard-{r:06d}.tar") for r in range(world)]
    cfg_path = os.path.join(root, f"config_{rank}.yaml")
    with open(cfg_path, "w") as f:
        f.write("\n".join(["urls:", "  - unused", "local_shard_paths:", *[f"  - {p}" for p in paths], f"num_shards: {world}",
                           "dataset_seed: 7", f"batch_size: {B}", "learning_rate: 1e-5", f"steps: {args.warmup + args.steps}",
                           "num_steps_per_validation: 1000000", "validation_prompts:", "  - x", "bfloat16: true",
                           "aspect_ratio: 1024", "train_unconditional_prob: 0.0", ""]))
    params = TrainingParameters()
    params.read_yaml(cfg_path)
    os.environ.setdefault("YAT_TENSORBOARD", "0")
    cwd = os.getcwd()
    os.chdir(root)                                   # the trainer writes models/ and runs/ relative to the cwd
    trainer = SanaModel(params, config=SanaConfig(num_layers=args.layers))
    trainer._validate_and_save = lambda: None        # the step-0 validation + 3.2 GB checkpoint write are not the step
    clock = {}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace_every = int(os.environ.get("YAT_BENCH_TRACE_STEPS", "0"))      # diagnostic: wall time per N steps (adds a sync each time)

    def on_step(step):
        if trace_every and step > args.warmup and (step - args.warmup) % trace_every == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            log(f"trainer mode: steps {step - trace_every - args.warmup}..{step - args.warmup}: "
                f"{1e3 * (now - clock.get('t_trace', clock['t0'])) / trace_every:.1f} ms/step")
            clock["t_trace"] = now
        if step == args.warmup:
            barrier()
            clock["t0"] = time.perf_counter()
        elif step == args.warmup + args.steps:
            clock["t_issue"] = time.perf_counter()       # host done enqueueing the timed steps (it runs ahead of the GPU)
            barrier()
            clock["t1"] = time.perf_counter()
    log(f"rank {rank}/{world}: trainer mode, {args.warmup} + {args.steps} steps from {len(paths)} shard(s) in {root}")
    host = {}
    if os.environ.get("YAT_BENCH_HOST_SPLIT", "0") != "0":       # diagnostic: where the trainer's host thread spends its time
        def timed(name, fn):
            def wrapped(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    host[name] = host.get(name, 0.0) + time.perf_counter() - t
            return wrapped
        trainer.optimize = timed("optimize (stage + enqueue fwd/bwd)", trainer.optimize)
        from yat_amd import plan as _plan, recipe as _recipe
        from yat_amd.optim import FlatAdamW as _Opt
        from yat_amd.common.bucket_sampler import BucketSampler as _Sampler
        _Opt.step = timed("optimizer.step", _Opt.step)
        _plan.LaunchPlan.replay = timed("  of which plan replay", _plan.LaunchPlan.replay)
        _recipe._Stager.begin = timed("  of which wait for a staging buffer", _recipe._Stager.begin)
        samp_iter = _Sampler.__iter__

        def iter_timed(self_):
            it = samp_iter(self_)
            while True:
                t = time.perf_counter()
                try:
                    b = next(it)
                except StopIteration:
                    return
                finally:
                    host["sampler (next batch)"] = host.get("sampler (next batch)", 0.0) + time.perf_counter() - t
                yield b
        _Sampler.__iter__ = iter_timed
    trainer.run(on_step=on_step)
    if host:
        n = args.warmup + args.steps
        log("trainer mode host split (ms per step, main thread): " + "; ".join(f"{k} {1e3 * v / n:.1f}" for k, v in host.items()))
    os.chdir(cwd)
    elapsed = clock["t1"] - clock["t0"]
    log(f"trainer mode: host enqueue time {1e3 * (clock['t_issue'] - clock['t0']) / args.steps:.1f} ms/step, "
        f"step {1e3 * elapsed / args.steps:.1f} ms")
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=trainer.accelerator.device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    loss_val = float(trainer.loss_history[-1])
    if rank == 0:
        cfg = trainer.model.cfg
        # text rows computed per image: the prompts' own lengths (uniform 20..300 in the synthetic shards, mean 160) when
        # the recipe packs the text rows, the reference's 512 otherwise
        trows = 160.0 if os.environ.get("YAT_TEXT_PACK", "1") != "0" else 512
        flops = B * sum(train_flops_per_image(cfg, h * w, trows) for h, w in BUCKETS) / len(BUCKETS)
        emit_json({
            "metric": "images/sec (whole node) SANA-1.6B 1024px bf16 training step", "value": world * B * args.steps / elapsed,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic shards on local disk (trainer mode)",
            "config": {"workload": "train_sana.py end to end: SANA-1.6B (D=2240, 20 blocks) 1024px, bf16, full fine-tune, "
                                   f"{{ratio, latent.pt, emb.pt}} tar shards -> BucketSampler (decode thread) -> one packed H2D -> "
                                   f"device step -> clip + AdamW; aspect buckets {BUCKETS}, T=512",
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": 1024, "parallelism": f"dp{world}",
                       "num_layers": cfg.num_layers, "params": trainer.model.numel_flat},
            "loss": loss_val, "hbm_peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "mfma_util_step": flops / (elapsed / args.steps) / (PEAK_BF16_TFLOPS * 1e12)})
    import shutil
    if world > 1:
        dist.barrier()
    if rank == 0:
        shutil.rmtree(root, ignore_errors=True)


def emit_json(obj):
    """The ONE line on the real stdout (see ``main``: fd 1 is pointed at stderr for everything else)."""
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


_JSON_OUT = None


def self_launch(n):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the ranks ourselves.  The reference's
    counterpart is ``accelerate launch`` (README.md:62; the process group it builds: common/trainer.py:31-37).  This parent
    has made no GPU call (importing torch does not initialise HIP) and never will: it starts ``python -m
    torch.distributed.run`` as a fresh CHILD process -- never an exec of the current one -- on a free loopback port, lets the
    child inherit stdout / stderr (rank 0 prints the one JSON line there) and exits with the child's code."""
    import socket
    import subprocess
    port = os.environ.get("YAT_BENCH_MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on these hosts (RCCL needs it across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {n} without a launcher: starting {' '.join(cmd[1:9])} ...")
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE config 2: 8)")
    ap.add_argument("--layers", type=int, default=20, help="debug only; anything but 20 is not the headline config")
    ap.add_argument("--lokr", type=int, default=0, metavar="RANK",
                    help="BASELINE config 5 instead of config 2: LoKr adapters of this rank on the README target modules "
                         "(alpha = rank, module dropout 0.05), frozen base; not the headline line")
    ap.add_argument("--lora", type=int, default=0, metavar="RANK",
                    help="plain LoRA adapters (lora_algo: lora) of this rank on the README target modules, alpha = rank; "
                         "not the headline line")
    ap.add_argument("--data", choices=["resident", "shards"], default="resident",
                    help="resident (headline): features in HBM before the timed region; shards: the whole trainer "
                         "(train_sana.SanaModel.run) fed from synthetic shards on local disk -- a side measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-timer", action="store_true")
    ap.add_argument("--lokr-pre-add", action="store_true",
                    help="LoKr / LoRA: adapter term computed by launches of its own + the pre_add epilogue instead of the base GEMM's "
                         "second operand pair (A/B of the two forms)")
    ap.add_argument("--gemm-detail", default=None, help="write per-shape GEMM timings of the roofline pass to this file")
    ap.add_argument("--roofline-steps", type=int, default=4, help="steps of the serialized GEMM-timing pass")
    ap.add_argument("--rccl-channels", type=int, default=-1, metavar="N",
                    help="cap RCCL at N channels (NCCL_MAX_NCHANNELS=N before the communicator is built): fewer channels = "
                         "fewer persistent collective workgroups competing with the GEMM tiles for CUs, at lower link "
                         "bandwidth; 0 = RCCL's default; -1 = the policy of yat_amd/ddp.py (RCCL's default unless "
                         "YAT_RCCL_CHANNELS asks for a cap; a site's own NCCL_*_NCHANNELS are never rewritten).  The value in "
                         "force is reported in the `comm` object")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1 (or YAT_DDP_FORCE=1): reduce-scatter the gradient buckets, AdamW on 1 / N of every bucket, all-gather "
                         "the parameters under the next forward (yat_amd/ddp.py shard_optimizer; also YAT_SHARD_OPTIMIZER=1); "
                         "bit-identical to the replicated step")
    ap.add_argument("--comm-steps", type=int, default=6, help="steps of each pass of the data-parallel diagnostics")
    ap.add_argument("--transport", choices=["torch", "native"], default=None,
                    help="gradient all-reduce transport of an N > 1 job (yat_amd/ddp.py): torch = torch.distributed's RCCL group "
                         "(default), native = the library's own communicator (yat_comm_*) with the launcher group on gloo")
    ap.add_argument("--coalesce", type=int, default=1, help="consecutive gradient buckets per collective (1 = one per block)")
    ap.add_argument("--phases", default=None, metavar="FILE",
                    help="after the timed region, time the phases of 6 steps with HIP events and write them to FILE")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))      # (before anything touches the GPU or rewires stdout)
    # stdout carries exactly one JSON line.  Libraries do not know that: RCCL prints its version block on stdout when the
    # first communicator is built (seen with a forced one-rank group), which would land in front of the line the driver
    # parses.  Keep a private handle on the real stdout for the JSON line and point fd 1 at stderr for everyone else.
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    from yat_amd.common.host import cap_host_threads
    cap_host_threads()       # the host thread's small CPU ops on this rank's share of the usable CPUs, not on 256 OpenMP threads

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one process per GPU.  (Rehearsal of the N > 1 flow on a one-GPU box: YAT_DIST_BACKEND=gloo lets several ranks share
    # cuda:0 -- RCCL refuses two ranks on one device -- so everything but the transport is exercised.)
    # YAT_COMM=native: gradients go through the library's own RCCL communicator (yat_comm_*, csrc/comm.hip); the process group is
    # then only rendezvous / barrier / max-over-ranks and is built over gloo, so the process holds ONE RCCL communicator
    if args.transport:
        os.environ["YAT_COMM"] = args.transport
    from yat_amd.ddp import group_backend
    backend = group_backend()
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # YAT_DDP_FORCE=1: run the whole data-parallel machinery (RCCL group, bucket hooks on the side stream, comm stream,
    # optimizer wait) even with ONE rank -- the only way to exercise that code path on a single-GPU box.
    force_ddp = os.environ.get("YAT_DDP_FORCE", "0") != "0"
    from yat_amd.ddp import apply_channel_policy
    apply_channel_policy(max(world, 2 if force_ddp else 1), None if args.rccl_channels < 0 else args.rccl_channels)
    if world > 1 or force_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.data == "shards":
        run_from_shards(args, rank, world, local_rank)
        if world > 1:
            dist.destroy_process_group()
        return

    from yat_amd import ops
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    from yat_amd.recipe import SanaRecipe
    from yat_amd.optim import FlatAdamW
    from yat_amd.ddp import HipDDP

    cfg = SanaConfig(num_layers=args.layers)
    log(f"rank {rank}/{world}: building SANA ({args.layers} blocks) on {dev}")
    model = SanaTransformer2DModelHIP(cfg, device=dev).init_synthetic(seed=0)
    trained = model
    if args.lokr:
        from yat_amd.lokr import LoKrAdapters
        trained = LoKrAdapters(model, ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2",
                                       "proj"], r=args.lokr, alpha=float(args.lokr), module_dropout=0.05,
                               pair=not args.lokr_pre_add)
        log(f"LoKr rank {args.lokr}: {len(trained.entries)} adapted modules, {trained.num_parameters():,} trainable parameters")
    elif args.lora:
        from yat_amd.lora import LoRAAdapters
        trained = LoRAAdapters(model, ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2",
                                       "proj"], r=args.lora, alpha=float(args.lora), pair=not args.lokr_pre_add)
        log(f"LoRA rank {args.lora}: {len(trained.entries)} adapted modules, {trained.num_parameters():,} trainable parameters")
    from yat_amd.common.trainer import adapter_arithmetic
    if trained is not model:
        log(f"adapter arithmetic: {adapter_arithmetic(trained)}")
    opt = FlatAdamW(trained, lr=1e-5, weight_decay=0.0, max_grad_norm=1.0,
                    overlap_update=os.environ.get("YAT_SERIAL", "0") == "0")
    ddp = HipDDP(trained, force=force_ddp, coalesce=args.coalesce, shard_optimizer=True if args.shard_optimizer else None) \
        if (world > 1 or force_ddp) else None
    # small consensus / timing scalars of the launcher-level group: host tensors when it is gloo (no device round trip)
    ctl_dev = dev if (world > 1 or force_ddp) and dist.is_initialized() and dist.get_backend() == "nccl" else torch.device("cpu")

    def communicator_view():
        if ddp is None:
            return None
        if ddp.native is not None:
            from yat_amd import lib as _ylib
            L = _ylib.load()
            return {"owner": "libyat_hip.so (yat_comm_*: RCCL bound at run time)", "rank": int(L.yat_comm_rank()),
                    "world": int(L.yat_comm_world())}
        return {"owner": f"torch.distributed process group ({dist.get_backend(ddp.pg)})", "rank": dist.get_rank(ddp.pg),
                "world": dist.get_world_size(ddp.pg)}
    if ddp:
        ddp.broadcast_parameters()
    recipe = SanaRecipe(model, pad_to=512, device=dev)
    B, T, Cc = args.batch, 512, cfg.caption_channels

    # ---- synthetic cached features, resident in HBM (seed 1234 + rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    hg = torch.Generator().manual_seed(1234 + rank)
    batches = []
    for (h, w) in BUCKETS:
        lat = (torch.randn(B, cfg.in_channels, h, w, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        lens = torch.randint(20, 301, (B,), generator=hg).tolist()
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        src = torch.randn(offs[-1], Cc, generator=g, device=dev).to(torch.bfloat16)
        batches.append(dict(h=h, w=w, lat=lat, src=src, offsets=torch.tensor(offs, dtype=torch.int32, device=dev),
                            work=ops.kv_work_list(lens, T, dev), lens=lens, rows=offs[-1]))
    enc = torch.empty(B, T, Cc, dtype=torch.bfloat16, device=dev)
    mask = torch.empty(B, T, dtype=torch.int64, device=dev)
    bias = torch.empty(B, T, dtype=torch.float32, device=dev)
    kvl = torch.empty(B, dtype=torch.int32, device=dev)
    loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
    noise_gen = torch.Generator(device=dev).manual_seed(99 + rank)   # throughput mode: advancing device stream
    ts_gen = torch.Generator().manual_seed(77 + rank)

    OPT_TIMER = None
    t_dev = torch.empty(B, dtype=torch.float32, device=dev)        # persistent: the step's launch plan holds their addresses
    sig_dev = torch.empty(B, dtype=torch.bfloat16, device=dev)

    def stage_text(b):
        """pad + mask (train_sana.py:168-180): packed text rows (no padding rows through the text-side GEMMs, see
        SanaRecipe.packs_text / SanaTransformer2DModelHIP.forward_impl) unless YAT_TEXT_PACK=0"""
        if recipe.packs_text(b["lens"]):
            enc_b = recipe.packed_enc(B, T, Cc, b["rows"])
            ops.pack_mask(b["src"], b["offsets"], B, T, Cc, enc_b, mask, bias, kvl)
            return enc_b, b["offsets"][:B]
        ops.pad_mask(b["src"], b["offsets"], B, T, Cc, enc, mask, bias, kvl)
        return enc, None

    def text_rows(b):
        """text rows per image that the step computes (FLOP accounting)"""
        return b["rows"] / B if recipe.packs_text(b["lens"]) else T

    def step(i):
        b = batches[i % len(batches)]
        enc_b, kv_off = stage_text(b)                                                               # train_sana.py:168-180
        noise = torch.randn(b["lat"].shape, generator=noise_gen, device=dev, dtype=torch.bfloat16)  # :183
        _, t_host, sig_host = recipe.scheduler.sample(B, ts_gen)                                    # :185-204
        t_dev.copy_(t_host, non_blocking=True)
        sig_dev.copy_(sig_host, non_blocking=True)
        recipe.train_step_device(b["lat"], enc_b, (bias, kvl), noise, t_dev, sig_dev, loss_dev, kv_work=b["work"],
                                 kv_off=kv_off)                                                     # :206-218 + bwd
        if ddp:
            ddp.wait()
        if OPT_TIMER is not None:           # roofline pass only: HIP events around clip + AdamW on its stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            opt.step()
            e1.record()
            OPT_TIMER.append((e0, e1))
        else:
            opt.step()                                                                              # trainer.py:347-356
        return b["h"] * b["w"], text_rows(b)

    def barrier():
        if world > 1 or force_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # The step runs on a stream of the compute-stream set (high priority level, one hardware queue each) rather than on the
    # default stream, which shares its hardware queue with whatever else the process creates -- with a process group around
    # that was the weight-gradient and optimizer streams, and the step lost 16 ms (profiles/LOG_r01_r03.md section 6, "hardware queues").
    from yat_amd.flat import compute_stream, isolate_streams
    if isolate_streams() or "chain=" in os.environ.get("YAT_PRIO", ""):      # (only with a process group around; YAT_PRIO: diagnostic)
        hp = compute_stream(dev)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    log("inputs resident; warm-up")
    for i in range(args.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss={loss_dev.item():.4f}")
    barrier()
    log(f"timing {args.steps} steps")
    t0 = time.perf_counter()
    flops = 0.0
    for i in range(args.steps):
        ntok, trows = step(args.warmup + i)
        flops += B * train_flops_per_image(cfg, ntok, trows)
    issue = time.perf_counter() - t0          # host time to enqueue all steps (the host runs ahead of the GPU)
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"host enqueue time {1e3 * issue / args.steps:.1f} ms/step")
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    loss_val = loss_dev.item()
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps, loss={loss_val:.4f}")
    # host cost of one step: enqueue onto an idle GPU (nothing pushes back), every bucket's launch plan already recorded
    host_ms = []
    for i in range(len(BUCKETS)):
        barrier()
        th = time.perf_counter()
        step(args.warmup + args.steps + i)
        host_ms.append(1e3 * (time.perf_counter() - th))
    barrier()
    log(f"host enqueue of one step onto an idle GPU: {min(host_ms):.1f} ms (per bucket: {', '.join(f'{v:.1f}' for v in host_ms)}; "
        f"launch plans {'on' if model.use_plans else 'off'})")

    # ---- data-parallel diagnostics (after the timed region; SURVEY 8(d) C4 / C5: "report overlap fraction", "report all-reduce
    # latency for the ~MB-scale gradient set"; replaces what Accelerate's DDP hides behind common/trainer.py:253,344).  Three
    # passes of --comm-steps steps on the same buckets: the step as timed; the step with the collective itself switched off
    # (hooks, events, streams, the optimizer's wait all still there): the difference is what the collective costs the step,
    # overlap and CU interference included; and a pass with HIP events around every bucket's collective on the communication
    # stream and around the compute stream's wait for it.  Every rank runs them (collectives), rank 0 reports.
    comm = None
    if ddp is not None:
        # A failure on ONE rank must not leave the others inside a collective: every pass ends with an agreement (all-reduce MAX
        # of a local error flag) and all ranks abandon the diagnostics together; the arithmetic on the harvested events is
        # local and has its own guard.  (A rank that dies INSIDE a pass still strands its peers until the process-group
        # timeout -- as in the reference, common/trainer.py:34 -- but an error raised at a pass boundary, e.g. by a HIP event
        # or an allocation, no longer does.)
        K = max(2, args.comm_steps)
        base = args.warmup + args.steps + len(BUCKETS)
        base += (-base) % len(BUCKETS)                     # every pass starts on the same bucket
        state = {"err": None, "dry_started": False}

        def agreed_ok():
            bad = 0 if state["err"] is None else 1
            if world > 1:
                f = torch.tensor([bad], dtype=torch.int32, device=ctl_dev)
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                bad = int(f.item())
            return bad == 0

        def timed_pass():
            barrier()
            tp = time.perf_counter()
            try:
                for i in range(K):
                    step(base + i)
            except Exception as e:                         # noqa: BLE001 -- reported in the comm object
                state["err"] = e
            barrier()
            dt = time.perf_counter() - tp
            if world > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = tt.item()
            return 1e3 * dt / K

        ms_live = ms_dry = ms_timing = float("nan")
        b0, n0 = ddp.bytes_reduced, ddp.buckets_reduced
        try:
            step(base)                                     # (plans of this bucket order are recorded by now; one settle step)
        except Exception as e:                             # noqa: BLE001
            state["err"] = e
        ok = agreed_ok()
        if ok:
            ms_live = timed_pass()
            ok = agreed_ok()
        if ok:
            ddp.dryrun, state["dry_started"] = True, True
            ms_dry = timed_pass()
            ddp.dryrun = False
            ok = agreed_ok()
        if state["dry_started"] and world > 1:
            ddp.broadcast_parameters()                     # the dry steps did not average: bring the replicas back together
        if ok:
            ddp.timing, ddp.timed_buckets, ddp.timed_waits = True, [], []
            b0, n0 = ddp.bytes_reduced, ddp.buckets_reduced
            ms_timing = timed_pass()
            ddp.timing = False
            ok = agreed_ok()
        ddp.dryrun = ddp.timing = False
        torch.cuda.synchronize()
        def comm_report():
            durs = sorted(e0.elapsed_time(e1) for _, _, e0, e1 in ddp.timed_buckets)
            sizes = [nb for _, nb, _, _ in ddp.timed_buckets]
            waits = [w0.elapsed_time(w1) for w0, w1 in ddp.timed_waits]
            comm_ms = sum(durs) / K
            exposed = max(0.0, ms_live - ms_dry)
            big = [(nb, e0.elapsed_time(e1)) for _, nb, e0, e1 in ddp.timed_buckets if nb >= (32 << 20)]
            comm = {
                "world": world, "forced_one_rank": bool(force_ddp and world == 1),
                # what the communicator itself says (not the launcher's environment): the driver can see that RCCL saw N ranks
                "communicator": communicator_view(),
                "transport": "native (yat_comm_* behind the C ABI)" if ddp.native is not None else f"torch.distributed ({backend})",
                "buckets_per_step": (ddp.buckets_reduced - n0) / K, "bytes_per_step": (ddp.bytes_reduced - b0) / K,
                "bucket_mb_min_max": [min(sizes) / 2 ** 20, max(sizes) / 2 ** 20] if sizes else None,
                "bucket_us_min_median_max": [1e3 * durs[0], 1e3 * durs[len(durs) // 2], 1e3 * durs[-1]] if durs else None,
                "comm_stream_ms_per_step": comm_ms,
                # per-rank algorithm bandwidth of the large buckets (bytes / time) and the bus bandwidth a ring moves for it
                "algbw_gbps_large_buckets": (sum(nb for nb, _ in big) / (1e-3 * sum(t for _, t in big)) / 1e9) if big else None,
                "busbw_factor": 2.0 * (world - 1) / world,
                "step_ms": ms_live, "step_ms_collective_off": ms_dry, "step_ms_timing_pass": ms_timing,
                "exposed_ms_per_step": exposed,
                "optimizer_wait_ms_per_step": sum(waits) / max(1, len(waits)),
                "overlap_frac": (1.0 - min(1.0, exposed / comm_ms)) if comm_ms > 0 else None,
                "rccl_channels": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS")},
                "coalesce": ddp.coalesce, "steps_per_pass": K,
                "optimizer": ("sharded: reduce-scatter -> AdamW on 1 / %d of every bucket -> all-gather" % ddp.world) if ddp.shard is not None
                else "replicated: all-reduce -> AdamW over every parameter on every rank",
                "note": ("exposed = step - step with the collective switched off (same inputs, same hooks / events / streams); "
                         "overlap_frac = 1 - exposed / summed comm-stream time; optimizer_wait = how long the compute stream sat "
                         "in HipDDP.wait() before clip + AdamW" + ("; ONE rank: RCCL runs its one-rank copy kernel, the link "
                         "figures mean nothing -- the line is a rehearsal of the reporting" if world == 1 else "")),
            }
            if args.lokr or args.lora:
                # BASELINE config 5: ONE small bucket (the adapter set's flat gradient), latency-bound
                comm["adapter_allreduce_latency_us"] = 1e3 * durs[len(durs) // 2] if durs else None
                comm["adapter_bucket_bytes"] = sizes[0] if sizes else None
            log(f"data-parallel diagnostics: step {ms_live:.2f} ms, collective off {ms_dry:.2f} ms, comm stream {comm_ms:.2f} ms/step "
                f"-> overlap {comm['overlap_frac']}")
            return comm

        if not ok:
            comm = {"error": repr(state["err"]) if state["err"] is not None else "another rank failed"}
            log(f"data-parallel diagnostics abandoned on every rank: {comm['error']}")
        else:
            try:
                comm = comm_report()
            except Exception as e:     # local arithmetic on harvested events only (no collective inside): never sinks the bench line
                comm = {"error": repr(e)}
                log(f"data-parallel diagnostics failed: {e!r}")

    # ---- optional phase probe (after the timed region; nothing of it runs otherwise): GPU timestamps of the step's phases
    # from a handful of HIP events per step -- unlike a profiler's kernel trace it does not slow the host's enqueue, so the
    # streams overlap as they do in the timed region.
    if args.phases and rank == 0:
        side, opt_stream = model._side_stream(), None
        rows = []
        barrier()
        for i in range(6):
            ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "fwd", "bwd_main", "side", "norm", "opt")}
            b = batches[(args.warmup + args.steps + i) % len(batches)]
            main = torch.cuda.current_stream()
            ev["start"].record(main)
            enc_b, kv_off = stage_text(b)
            noise = torch.randn(b["lat"].shape, generator=noise_gen, device=dev, dtype=torch.bfloat16)
            _, t_host, sig_host = recipe.scheduler.sample(B, ts_gen)
            t_dev.copy_(t_host, non_blocking=True)
            sig_dev.copy_(sig_host, non_blocking=True)
            noisy, target = ops.flow_mix(b["lat"], noise, sig_dev, recipe._noisy(b["lat"]), recipe._target(b["lat"]))
            pred = model.forward_device(noisy, enc_b, t_dev, bias, kvl, kv_work=b["work"], kv_off=kv_off)
            ev["fwd"].record(main)
            dpred = recipe._dpred(pred)
            ops.mse_fwd_bwd(pred, target, loss_dev, dpred, recipe._mse_ws)
            model.backward_device(dpred)
            ev["bwd_main"].record(main)        # main stream after the backward (includes its join with the side stream)
            ev["side"].record(side)            # side stream: the last weight gradient
            opt.step()
            ev["norm"].record(main)            # gradient norm + clip coefficient (main stream)
            ev["opt"].record(opt._stream if opt._stream is not None else main)
            rows.append(ev)
        barrier()
        with open(args.phases, "w") as f:
            f.write("# ms after the step's first launch (HIP events; B=%d, buckets round-robin)\n" % B)
            f.write("# step  forward_done  backward_done(main,joined)  side_stream_done  gradnorm_done  adamw_done  next_step_start\n")
            for i, ev in enumerate(rows):
                nxt = rows[i + 1]["start"] if i + 1 < len(rows) else None
                f.write("%5d %13.2f %27.2f %17.2f %14.2f %11.2f %16s\n" % (
                    i, ev["start"].elapsed_time(ev["fwd"]), ev["start"].elapsed_time(ev["bwd_main"]),
                    ev["start"].elapsed_time(ev["side"]), ev["start"].elapsed_time(ev["norm"]),
                    ev["start"].elapsed_time(ev["opt"]),
                    "%.2f" % ev["start"].elapsed_time(nxt) if nxt is not None else "-"))
        log(f"phase probe written to {args.phases}")

    # ---- roofline pass (after the timed region): per-launch GEMM durations by HIP events on the launch stream.
    # The step overlaps independent GEMMs on two streams, so in the timed region two kernels share the CUs and
    # their individual durations overlap; the per-kernel figure is therefore taken with the streams serialized
    # (YAT_SERIAL=1 behaviour), same kernels, same shapes, same inputs.
    timer = None
    if not args.no_gemm_timer:          # every rank runs it (the DDP collectives need all of them); rank 0 reports
        saved = (model.side_wgrad, opt.overlap_update, model.fwd_chains)
        model.side_wgrad, opt.overlap_update, model.fwd_chains = False, False, 1
        step(0)
        torch.cuda.synchronize()
        timer = []
        ops.GEMM_TIMER = timer
        OPT_TIMER = []
        for i in range(args.roofline_steps):
            step(1 + i)
        torch.cuda.synchronize()
        ops.GEMM_TIMER = None
        opt_timer, OPT_TIMER = OPT_TIMER, None
        model.side_wgrad, opt.overlap_update, model.fwd_chains = saved
        barrier()

    if rank == 0:
        img_s = world * B * args.steps / elapsed
        res = {
            "metric": "images/sec (whole node) SANA-1.6B 1024px bf16 training step",
            "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("train_sana.py: SANA-1.6B (D=2240, 20 blocks) 1024px, bf16, "
                                    + (f"LoKr rank {args.lokr} adapters on a frozen base (BASELINE config 5), " if args.lokr
                                       else f"LoRA rank {args.lora} adapters on a frozen base, " if args.lora
                                       else "full fine-tune, ")
                                    + (f"adapter arithmetic: {adapter_arithmetic(trained)}, " if (args.lokr or args.lora) else "") +
                                    f"cached latents/text embeds, aspect buckets {BUCKETS} round-robin, prompts of 20..300 tokens "
                                    + ("(text rows packed: the masked padding rows to T=512 are not computed)"
                                       if recipe.packs_text(batches[0]["lens"]) else "padded to T=512") + ", AdamW+clip"),
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": 1024, "parallelism": f"dp{world}",
                       "num_layers": cfg.num_layers, "params": model.numel_flat},
            "loss": loss_val,
            "hbm_peak_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
            "host_enqueue_ms_per_step": min(host_ms),
            "mfma_util_step": (flops / args.steps / (elapsed / args.steps)) / (PEAK_BF16_TFLOPS * 1e12),
            "algorithmic_tflop_per_step": flops / args.steps / 1e12,
        }
        if comm is not None:
            res["comm"] = comm
        if getattr(ops, "DIAGNOSTIC_INVALID", None):   # set only by scripts/step_ablation.py (kernel families skipped): NOT a
            res["invalid"] = ops.DIAGNOSTIC_INVALID     # measurement of the workload
        if timer:
            gf = sum(t[0] for t in timer)
            gms = sum(t[1].elapsed_time(t[2]) for t in timer)
            ach = gf / (gms * 1e-3) / 1e12
            # HBM traffic of the same kernel family per launch, from the committed PMC passes (collected separately, as
            # the profiling guide prescribes: --pmc runs cannot be combined with the timed run)
            traffic, traffic_src = None, None
            tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "gemm_traffic.json")
            if os.path.exists(tj):
                with open(tj) as f:
                    tdat = json.load(f)
                traffic, traffic_src = tdat["hbm_bytes_per_launch"], tdat["source"]
                col = tdat.get("collected")
                traffic_src = (f"{traffic_src} -- NOT measured in this run: collected {col['when']} on {col['where']}"
                               + (f", tree {col['tree']}" if col.get("tree") else "")) if col else \
                    f"{traffic_src} -- NOT measured in this run (separate rocprofv3 --pmc passes on a builder's box)"
            alg_bytes = sum(t[4] for t in timer if len(t) > 4) / max(1, sum(1 for t in timer if len(t) > 4))
            res["roofline"] = {"bound": "mfma", "kernel": "gemm256_kernel / gemm_bf16_kernel (NT/NN/TN, all epilogues)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                               "traffic_unit": "bytes/launch (PMC FETCH_SIZE x 2 + WRITE_SIZE: requests of the L2s to the fabric -- "
                                               "re-reads served by the 256 MB Infinity Cache are counted, so this bounds the HBM "
                                               "bytes from above; profiles/LOG_r01_r03.md section 14, item 6b)",
                               "traffic_source": traffic_src, "algorithmic_operand_bytes_per_launch": alg_bytes,
                               "mode": f"serialized-stream pass of {args.roofline_steps} steps after the timed region",
                               "launches": len(timer), "avg_launch_us": 1e3 * gms / len(timer),
                               "gemm_ms_per_step_serialized": gms / args.roofline_steps}
        if timer:
            # second roofline, for the dominant HBM-bound pass: gradient norm + clip + AdamW over the flat buffers
            # (algorithmic 2 B/param read for the norm + 14 B/param for the update; PMC: profiles/r01_g_pmc_per_kernel.txt
            # shows exactly 12.8 GB fetched + 9.6 GB written by adamw_kernel per step)
            oms = sum(a.elapsed_time(b_) for a, b_ in opt_timer) / max(1, len(opt_timer))
            nopt = trained.numel_flat                 # what the optimizer walks: the model, or only the adapters' flat buffer
            obytes = 16.0 * nopt
            res["roofline_hbm"] = {"bound": "hbm", "kernel": "gradnorm_partial_kernel + adamw_kernel (clip + AdamW step)",
                                   "achieved": obytes / (oms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": obytes / (oms * 1e-3) / 1e9 / 8000.0, "traffic": 16.0 * nopt,
                                   "ms": oms, "mode": "serialized-stream pass"}
        if timer and args.gemm_detail:
            agg = {}
            for fl, e0, e1, key, *_ in timer:
                a = agg.setdefault(key, [0, 0.0, fl])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            with open(args.gemm_detail, "w") as f:
                f.write("layout      M      N      K  act gate res aux  calls/step   avg_us    TFLOP/s   ms/step\n")
                for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    lay, M_, N_, K_, act, gt, rs, ax = key
                    f.write(f"{lay:4s} {M_:7d} {N_:6d} {K_:6d} {act:>5s} {int(gt):3d} {int(rs):3d} {int(ax):3d} "
                            f"{n / args.roofline_steps:9.1f} {1e3 * ms / n:9.1f} {fl * n / (ms * 1e-3) / 1e12:9.1f} "
                            f"{ms / args.roofline_steps:9.3f}\n")
        if world == 1 and not args.no_cpu_baseline:
            try:
                model.join_pending_update()
                sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
                res["cpu_baseline"] = cpu_baseline(sd, cfg.num_layers)
            except Exception as e:  # the baseline is a reported side number; never let it sink the bench line
                res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": usable_cores(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        emit_json(res)
    if ddp is not None and ddp.native is not None:
        torch.cuda.synchronize()
        ddp.native.destroy()               # the library's communicator goes before the process does (no RCCL teardown at exit)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
