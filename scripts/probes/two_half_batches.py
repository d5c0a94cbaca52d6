#!/usr/bin/env python3
"""Round 6 (review item 1b, 'the backward as two chains over the image halves'): an upper bound without rebuilding the backward.
Two SANA-1.6B instances with B = 4 each run their forward + backward CONCURRENTLY on two stream sets (each with its own side
streams), against one instance with B = 8 -- the same images, the same kernels, twice the independent dependent chains.  The
two-instance form even does a little more work (weight gradients over two K = 4096 halves instead of one K = 8192).  No
optimizer step on either side (weights are static: forward + backward only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
from yat_amd.recipe import SanaRecipe
BF, DEV = torch.bfloat16, "cuda"
cfg = SanaConfig()
g = torch.Generator().manual_seed(1)
B = 8
lat = (torch.randn(B, cfg.in_channels, 32, 32, generator=g) * 0.5).to(BF)
lens = torch.randint(20, 301, (B,), generator=g).tolist()
embs = [torch.randn(n, cfg.caption_channels, generator=g).to(BF) for n in lens]
STEPS, WARM = 12, 4


def run(models, parts):
    """models: [(model, recipe, stream)], parts: [(lo, hi)] image ranges; every step enqueues each model's step on its stream."""
    def one():
        for (m, r, st), (lo, hi) in zip(models, parts):
            with torch.cuda.stream(st):
                r.optimize_device(lat[lo:hi], embs[lo:hi], torch.Generator())
    for _ in range(WARM):
        one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(STEPS):
        one()
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0) / STEPS


def make(n):
    out = []
    for i in range(n):
        m = SanaTransformer2DModelHIP(cfg, device=DEV).init_synthetic(0)
        m.train()
        out.append((m, SanaRecipe(m, pad_to=512, device=DEV), torch.cuda.Stream()))
    return out


one = make(1)
t8 = run(one, [(0, 8)])
t4 = run(one, [(0, 4)])
del one
torch.cuda.empty_cache()
two = make(2)
t44 = run(two, [(0, 4), (4, 8)])
print(f"forward + backward, no optimizer, 32 x 32 latents: one instance B = 8: {t8:.2f} ms; one instance B = 4 alone: {t4:.2f} ms "
      f"(x 2 = {2 * t4:.2f}); two instances B = 4 + 4 concurrently: {t44:.2f} ms")
