#!/usr/bin/env python3
"""Time a few SANA GEMM shapes with the policy's variant (used with the -DYAT_ABL_* diagnostic builds via YAT_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
M, D, Hc = 8192, 2240, 5600
SHAPES = [("qkv_fwd", "nt", M, 3 * D, D), ("out_fwd", "nt", M, D, D), ("inv_fwd", "nt", M, 2 * Hc, D),
          ("inv_dgrad", "nn", M, D, 2 * Hc), ("qkv_wgrad", "tn", 3 * D, D, M), ("inv_wgrad", "tn", 2 * Hc, D, M)]
out_line = []
for name, lay, m, n, k in SHAPES:
    a_t, b_t = lay == "tn", lay in ("nn", "tn")
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    for _ in range(3):
        ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out_line.append(f"{name} {us:6.1f}us {2.0 * m * n * k / us / 1e6:6.0f}TF")
print(" | ".join(out_line))
