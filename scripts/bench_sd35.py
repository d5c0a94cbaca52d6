#!/usr/bin/env python3
"""SD3.5-Medium (MMDiT) 1024 px training step (BASELINE config 4, SURVEY.md 8(d) "C4") on one MI355X -- a side measurement,
NOT the headline bench (that is bench.py, SANA-1.6B).  Same shape of workload as the reference's train_sd35.py:165-194:
cached latents [B, 16, h/8, w/8] over the 1024 px aspect table, prompt embeddings [B, 333, 4096] (77 CLIP + 256 T5 tokens),
pooled projections [B, 2048], flow-matching scale_noise, target noise - latents, bf16 MSE, clip 1.0 + AdamW; synthetic data,
random-init weights of the real architecture (24 blocks, 13 of them with the second attention).

    python scripts/bench_sd35.py [--batch 8] [--steps 6] [--warmup 2] [--layers 24] [--gemm-detail FILE]

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
    """6 x forward MACs (fwd + dgrad + wgrad), no recompute.  Per block: image stream 12 D^2 N (qkv 3, out 1, ff 8), text
    stream 12 D^2 T (4 D^2 T in the last block: its text side ends inside the attention), joint attention 2 (N+T)^2 D;
    dual-attention blocks add 4 D^2 N + 2 N^2 D.  33.8 TFLOP per image at N = 4096, T = 333."""
    D, L = cfg.inner_dim, cfg.num_layers
    macs = 0.0
    for i in range(L):
        last, dual = i == L - 1, i in cfg.dual_attention_layers
        macs += 12 * D * D * N + (3 if last else 12) * D * D * T + 2 * (N + T) ** 2 * D
        if dual:
            macs += 4 * D * D * N + 2 * N * N * D
    macs += N * cfg.in_channels * cfg.patch_size ** 2 * D + T * cfg.joint_attention_dim * D + \
        N * D * cfg.out_channels * cfg.patch_size ** 2
    return 6.0 * macs


def log(msg):
    print(f"[bench_sd35] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--gemm-detail", default=None)
    ap.add_argument("--chains", type=int, default=0, help="independent forward chains over image ranges (0 = the model's default)")
    ap.add_argument("--host-profile", default=None, metavar="FILE", help="cProfile of four steps enqueued onto an idle GPU")
    ap.add_argument("--roofline-steps", type=int, default=2)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (the HIP path has no CPU fallback)")
    from yat_amd.common.host import cap_host_threads
    cap_host_threads()                    # as the trainer does (HipAccelerator): the OpenMP pool vs the CPUs this job really has
    dev = torch.device("cuda", 0)
    from yat_amd import ops
    from yat_amd.sd3 import SD3Config, SD3Transformer2DModelHIP
    from yat_amd.recipe import SD3Recipe
    from yat_amd.optim import FlatAdamW

    cfg = SD3Config(num_layers=args.layers, dual_attention_layers=tuple(i for i in range(13) if i < args.layers))
    model = SD3Transformer2DModelHIP(cfg, device=dev).init_synthetic(seed=0)
    log(f"SD3.5-Medium MMDiT: {args.layers} blocks ({len(cfg.dual_attention_layers)} dual), {model.numel_flat / 1e6:.1f} M parameters")
    if args.chains > 0:
        model.fwd_chains = args.chains
    opt = FlatAdamW(model, lr=1e-5, weight_decay=0.0, max_grad_norm=1.0, overlap_update=True)
    recipe = SD3Recipe(model, device=dev)
    B, T = args.batch, 333
    g = torch.Generator(device=dev).manual_seed(1234)
    batches = []
    for (Hl, Wl) in BUCKETS:
        batches.append(dict(Hl=Hl, Wl=Wl,
                            lat=(torch.randn(B, cfg.in_channels, Hl, Wl, generator=g, device=dev) * 0.5).to(torch.bfloat16),
                            prompt=torch.randn(B, T, cfg.joint_attention_dim, generator=g, device=dev).to(torch.bfloat16),
                            pooled=torch.randn(B, cfg.pooled_projection_dim, generator=g, device=dev).to(torch.bfloat16)))
    loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
    noise_gen = torch.Generator(device=dev).manual_seed(99)
    ts_gen = torch.Generator().manual_seed(77)

    t_dev = torch.empty(B, dtype=torch.float32, device=dev)        # persistent: the step's launch plan holds their addresses
    sig_dev = torch.empty(B, dtype=torch.bfloat16, device=dev)

    def step(i):
        b = batches[i % len(batches)]
        noise = torch.randn(b["lat"].shape, generator=noise_gen, device=dev, dtype=torch.bfloat16)   # :180
        _, t, sig = recipe.scheduler.sample(B, ts_gen)                                               # :182-184
        t_dev.copy_(t, non_blocking=True)
        sig_dev.copy_(sig, non_blocking=True)
        recipe.train_step_device(b["lat"], b["prompt"], b["pooled"], noise, t_dev, sig_dev, loss_dev)   # :185-193 + backward
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
        "metric": "images/sec SD3.5-Medium 1024px bf16 training step (BASELINE config 4, side measurement)",
        "value": B * args.steps / elapsed, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"train_sd35.py: SD3.5-Medium MMDiT (D=1536, 24x64 heads, {cfg.num_layers} blocks, "
                               f"{len(cfg.dual_attention_layers)} dual-attention) 1024px, bf16, full fine-tune, cached latents / prompt "
                               f"embeds / pooled projections, aspect buckets {BUCKETS} round-robin, T=333, flow matching, AdamW+clip",
                   "per_gpu_batch": B, "seq_len": 4096 + T, "params": model.numel_flat},
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
