#!/usr/bin/env python3
"""Three SANA GEMMs (conv_inverted forward / dgrad / wgrad: the largest of a block, B = 8) a few times each under the library
named by YAT_HIP_LIB, for `rocprofv3 --pmc FETCH_SIZE` passes over tile-order variants (-DYAT_GEMM_GROUP=n builds of gemm256.hip)
and, without the profiler, their times.  Prints per-shape microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
M, D, Hc = 8192, 2240, 5600
SH = [("inv_fwd  nn 8192x11200x2240", False, False, M, 2 * Hc, D), ("inv_dgrad nt 8192x2240x11200", False, True, M, D, 2 * Hc),
      ("inv_wgrad tt 11200x2240x8192", True, True, 2 * Hc, D, M)]
for name, a_t, b_t, m, n, k in SH:
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    for _ in range(2):
        ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 6 * 1e3
    print(f"{name}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF/s   algorithmic operand+output bytes {2 * (m * k + n * k + m * n) / 1e6:.1f} MB", flush=True)
