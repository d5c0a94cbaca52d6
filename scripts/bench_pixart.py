#!/usr/bin/env python3
"""PixArt-Sigma-XL 1024 px training step (BASELINE config 3, SURVEY.md 8(d) "C3") on one MI355X -- a side measurement,
NOT the headline bench (that is bench.py, SANA-1.6B).  Same shape of workload as the reference's train_pixart_sigma.py:
cached latents [B, 4, h/8, w/8] over the 1024 px aspect table, T5 embeddings [L_i <= 300, 4096] padded to 300 on the device,
DDPM add_noise, epsilon prediction, bf16 MSE, clip 1.0 + AdamW; synthetic data, random-init weights of the real architecture.

    python scripts/bench_pixart.py [--batch 8] [--steps 10] [--warmup 3] [--layers 28] [--gemm-detail FILE]

Prints one JSON line in bench.py's format (metric images/s; roofline = the GEMM family, serialized pass).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PEAK_BF16_TFLOPS = 2500.0
# latent grids (H, W) = ASPECT_RATIO_1024_BIN / 8: 4096 tokens of 2x2 patches each
BUCKETS = [(128, 128), (64, 256), (96, 168), (176, 88)]


def train_flops_per_image(cfg, N, T):
    """6 x forward MACs (fwd + dgrad + wgrad), no recompute; SURVEY.md 8(d): 19.90 TFLOP at N=4096, T=300."""
    D = cfg.inner_dim
    blk = N * 14 * D * D + T * 2 * D * D + 2 * N * N * D + 2 * N * T * D
    emb = N * cfg.in_channels * cfg.patch_size ** 2 * D + T * (cfg.caption_channels * D + D * D) + \
        N * D * cfg.out_channels * cfg.patch_size ** 2
    return 6.0 * (cfg.num_layers * blk + emb)


def log(msg):
    print(f"[bench_pixart] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--gemm-detail", default=None)
    ap.add_argument("--chains", type=int, default=0, help="independent forward chains over image ranges (0 = the model's default)")
    ap.add_argument("--host-profile", default=None, metavar="FILE", help="cProfile of four steps enqueued onto an idle GPU")
    ap.add_argument("--roofline-steps", type=int, default=2)
    ap.add_argument("--lora", type=int, default=0, metavar="RANK", help="plain LoRA adapters of this rank on a frozen base")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (the HIP path has no CPU fallback)")
    from yat_amd.common.host import cap_host_threads
    cap_host_threads()                    # as the trainer does (HipAccelerator): the OpenMP pool vs the CPUs this job really has
    dev = torch.device("cuda", 0)
    from yat_amd import ops
    from yat_amd.pixart import PixArtConfig, PixArtTransformer2DModelHIP
    from yat_amd.recipe import PixArtRecipe
    from yat_amd.optim import FlatAdamW

    cfg = PixArtConfig(num_layers=args.layers)
    model = PixArtTransformer2DModelHIP(cfg, device=dev).init_synthetic(seed=0)
    log(f"PixArt-Sigma: {args.layers} blocks, {model.numel_flat / 1e6:.1f} M parameters")
    trained = model
    if args.lora:
        from yat_amd.lora import LoRAAdapters
        trained = LoRAAdapters(model, ["to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"], r=args.lora,
                               alpha=float(args.lora))
        log(f"LoRA rank {args.lora}: {len(trained.entries)} adapted modules, {trained.num_parameters():,} trainable parameters")
    if args.chains > 0:
        model.fwd_chains = args.chains
    opt = FlatAdamW(trained, lr=1e-5, weight_decay=0.0, max_grad_norm=1.0, overlap_update=True)
    recipe = PixArtRecipe(model, device=dev)
    B, T, Cc = args.batch, 300, cfg.caption_channels
    g = torch.Generator(device=dev).manual_seed(1234)
    hg = torch.Generator().manual_seed(1234)
    batches = []
    for (Hl, Wl) in BUCKETS:
        lat = (torch.randn(B, cfg.in_channels, Hl, Wl, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        lens = torch.randint(20, 301, (B,), generator=hg).tolist()
        offs = [0]
        for L in lens:
            offs.append(offs[-1] + L)
        src = torch.randn(offs[-1], Cc, generator=g, device=dev).to(torch.bfloat16)
        batches.append(dict(Hl=Hl, Wl=Wl, lat=lat, src=src, offsets=torch.tensor(offs, dtype=torch.int32, device=dev),
                            work=ops.kv_work_list(lens, T, dev)))
    enc = torch.empty(B, T, Cc, dtype=torch.bfloat16, device=dev)
    mask = torch.empty(B, T, dtype=torch.int64, device=dev)
    bias = torch.empty(B, T, dtype=torch.float32, device=dev)
    kvl = torch.empty(B, dtype=torch.int32, device=dev)
    loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
    noise_gen = torch.Generator(device=dev).manual_seed(99)
    ts_gen = torch.Generator().manual_seed(77)

    t_dev = torch.empty(B, dtype=torch.float32, device=dev)        # persistent: the step's launch plan holds their addresses
    a_dev, c_dev = (torch.empty(B, dtype=torch.bfloat16, device=dev) for _ in range(2))

    def step(i):
        b = batches[i % len(batches)]
        ops.pad_mask(b["src"], b["offsets"], B, T, Cc, enc, mask, bias, kvl)                        # :158-168
        noise = torch.randn(b["lat"].shape, generator=noise_gen, device=dev, dtype=torch.bfloat16)   # :170
        t, a, c = recipe.scheduler.sample(B, ts_gen)                                                 # :172-174
        t_dev.copy_(t, non_blocking=True)                 # int64 timestep -> the float the embedder takes (exact)
        a_dev.copy_(a, non_blocking=True)
        c_dev.copy_(c, non_blocking=True)
        recipe.train_step_device(b["lat"], enc, (bias, kvl), noise, t_dev, a_dev, c_dev, loss_dev, kv_work=b["work"])   # :176-184 + bwd
        opt.step()
        return (b["Hl"] // cfg.patch_size) * (b["Wl"] // cfg.patch_size)

    for i in range(args.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss={loss_dev.item():.4f}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flops = 0.0
    for i in range(args.steps):
        flops += B * train_flops_per_image(cfg, step(args.warmup + i), T)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss_val = loss_dev.item()
    log(f"{elapsed:.3f}s for {args.steps} steps (host enqueue {1e3 * issue / args.steps:.1f} ms/step), loss={loss_val:.4f}")
    # host cost of one step: enqueue onto an idle GPU, every bucket's launch plan already recorded
    host_ms = []
    for i in range(len(BUCKETS)):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step(args.warmup + args.steps + i)
        host_ms.append(1e3 * (time.perf_counter() - th))
    torch.cuda.synchronize()
    log(f"host enqueue of one step onto an idle GPU: {min(host_ms):.1f} ms (per bucket: {', '.join(f'{v:.1f}' for v in host_ms)}; "
        f"launch plans {'on' if model.use_plans else 'off'}, {getattr(model, 'plan_replays', 0)} replays)")

    if args.host_profile:
        import cProfile, pstats, io
        pr = cProfile.Profile()
        for i in range(len(BUCKETS)):
            torch.cuda.synchronize()
            pr.enable()
            step(args.warmup + args.steps + i)
            pr.disable()
        torch.cuda.synchronize()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(35)
        open(args.host_profile, "w").write(buf.getvalue())

    # serialized pass for the per-launch GEMM figure
    saved = (model.side_wgrad, opt.overlap_update, model.fwd_chains)
    model.side_wgrad, opt.overlap_update, model.fwd_chains = False, False, 1
    step(0)
    torch.cuda.synchronize()
    timer = []
    ops.GEMM_TIMER = timer
    for i in range(args.roofline_steps):
        step(1 + i)
    torch.cuda.synchronize()
    ops.GEMM_TIMER = None
    model.side_wgrad, opt.overlap_update, model.fwd_chains = saved

    gf = sum(t[0] for t in timer)
    gms = sum(t[1].elapsed_time(t[2]) for t in timer)
    ach = gf / (gms * 1e-3) / 1e12
    res = {
        "metric": "images/sec PixArt-Sigma-XL 1024px bf16 training step (BASELINE config 3, side measurement)",
        "value": B * args.steps / elapsed, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"train_pixart_sigma.py: PixArt-Sigma-XL-2 (D=1152, 16x72 heads, {cfg.num_layers} blocks) 1024px, "
                               f"bf16, {'LoRA rank %d on a frozen base' % args.lora if args.lora else 'full fine-tune'}, cached latents/T5 embeds, aspect buckets {BUCKETS} round-robin, T=300, "
                               "DDPM eps-prediction, AdamW+clip", "per_gpu_batch": B, "seq_len": 4096, "params": model.numel_flat},
        "loss": loss_val, "hbm_peak_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
        "host_enqueue_ms_per_step": min(host_ms),
        "mfma_util_step": (flops / args.steps / (elapsed / args.steps)) / (PEAK_BF16_TFLOPS * 1e12),
        "algorithmic_tflop_per_step": flops / args.steps / 1e12,
        "roofline": {"bound": "mfma", "kernel": "gemm256_kernel / gemm_bf16_kernel", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                     "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                     "mode": f"serialized-stream pass of {args.roofline_steps} steps", "launches": len(timer),
                     "gemm_ms_per_step_serialized": gms / args.roofline_steps},
    }
    if args.gemm_detail:
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
    print(json.dumps(res))


if __name__ == "__main__":
    main()
