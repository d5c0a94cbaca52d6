#!/usr/bin/env python3
"""clip + AdamW over the SANA-1.6B flat buffers (1.6 B bf16 parameters, 14 B/param + 2 B/param for the norm), alone on the chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
n = 1604462752 // 8 * 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
p = (torch.randn(n, device=dev, generator=g) * 0.02).to(torch.bfloat16)
gr = (torch.randn(n, device=dev, generator=g) * 0.01).to(torch.bfloat16)
m = torch.zeros_like(p); v = torch.zeros_like(p)
coef = torch.ones(1, device=dev)
def run(step):
    ops.adamw_step(p, gr, m, v, coef, 1e-5, 0.9, 0.999, 1e-8, 0.0, step, zero_grad=False)
for s in range(1, 3): run(s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for s in range(3, 9): run(s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 6
print(f"adamw variant {os.environ.get('YAT_ADAMW_VARIANT', '0')} blocks {os.environ.get('YAT_ADAMW_BLOCKS', '8192')}: {ms:.3f} ms  {14.0 * n / ms / 1e6:.0f} GB/s  checksum {p.float().sum().item():.4f}")
