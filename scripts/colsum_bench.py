#!/usr/bin/env python3
"""Time yat_colsum_bf16 (bias gradients) on the SANA shapes; GB/s of the one read pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
out_line = []
for rows, cols in ((8192, 2240), (8192, 11200), (4096, 4480)):
    x = torch.randn(rows, cols, device=dev).to(BF)
    out = torch.empty(cols, dtype=BF, device=dev)
    ws = torch.empty(int(ops._lib().yat_colsum_workspace_bytes(rows, cols)), dtype=torch.uint8, device=dev)
    for _ in range(3):
        ops.colsum(x, out, ws) if hasattr(ops, "colsum") else None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.colsum(x, out, ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    ref = x.float().sum(0)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    out_line.append(f"{rows}x{cols}: {us:6.1f} us {rows * cols * 2 / us / 1e3:6.0f} GB/s (rel err {err:.1e}, sum {out.float().sum().item():.6f})")
print(" | ".join(out_line))
