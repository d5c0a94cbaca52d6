#!/usr/bin/env python3
"""Weight-gradient GEMMs (tt layout: dW[out,in] = dY^T X, K = tokens) of SD3.5 / PixArt / SANA, alone on the chip:
policy pick vs forced tile / split-K / tile-order group, with and without the fused bias gradient (row sums of dY)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
SHAPES = [(4608, 1536, 32768), (1536, 4608, 32768), (6144, 1536, 32768), (1536, 6144, 32768), (1536, 1536, 32768),
          (1152, 1152, 32768), (4608, 1152, 32768), (1152, 4608, 32768), (3456, 1152, 32768),
          (11200, 2240, 8192), (2240, 2240, 8192), (2240, 5600, 8192)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
def run(m, n, k, a, b, out, rs, variant, reps=4):
    f = lambda: ops.gemm(a, b, out, a_t=True, b_t=True, M=m, N=n, K=k, variant=variant, a_rowsum=rs)
    try:
        f()
    except Exception:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for m, n, k in SHAPES:
    a = (torch.randn(k, m, device=dev) * 0.5).to(BF)
    b = (torch.randn(k, n, device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    rs = torch.empty(m, dtype=BF, device=dev)
    line = [f"tt {m:5d}x{n:5d}x{k:5d}:"]
    for label, variant, r in (("policy", 0, None), ("policy+rowsum", 0, rs), ("256", 4, None), ("320", 5, None),
                              ("256 k2", 204, None), ("320 k2", 205, None), ("256 k3", 304, None), ("320 k3", 305, None),
                              ("256 k4", 404, None), ("256 g8", 8000004, None), ("256 g2", 2000004, None), ("256 g1", 1000004, None),
                              ("256 k2 g8", 8000204, None), ("256 k2 g1", 1000204, None)):
        us = run(m, n, k, a, b, out, r, variant)
        line.append(f"{label} " + (f"{us:6.0f}us {2.0 * m * n * k / us / 1e6:5.0f}TF" if us else "   n/a"))
    print("  |  ".join(line), flush=True)
