#!/usr/bin/env python3
"""Tile order (row-group size of the XCD-contiguous order, policy word of yat_gemm_bf16_ex) on COLD operands: round 3's group
sweep (profiles/r03_d_*) ran back-to-back launches, i.e. on operands resident in the Infinity Cache, where the order moved the
L2-miss traffic 2.7 x and the time < 1 %.  The step runs cold (scripts/gemm_cold_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
SHAPES = [("nn", 8192, 2240, 5600), ("nn", 8192, 11200, 2240), ("nn", 8192, 6720, 2240), ("nn", 8192, 2240, 2240),
          ("nt", 8192, 5600, 2240), ("nt", 8192, 2240, 6720), ("nt", 8192, 2240, 11200), ("nt", 8192, 2240, 2240),
          ("tt", 2240, 5600, 8192), ("tt", 6720, 2240, 8192), ("tt", 11200, 2240, 8192), ("tt", 2240, 2240, 8192)]
GROUPS = [0, 1, 2, 4, 8, 16, 32]
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def cold(f):
    ts = []
    for _ in range(5):
        junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[2]
for lay, m, n, k in SHAPES:
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    line = [f"{lay} {m:6d}x{n:6d}x{k:6d} cold us:"]
    for tile in (5, 4):
        for g in GROUPS:
            f = lambda: ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=tile + 1000000 * g)
            try:
                f(); torch.cuda.synchronize()
            except Exception:
                line.append(f"t{tile}g{g} n/a"); continue
            line.append(f"t{tile}g{g} {cold(f):6.1f}")
        line.append("|")
    print("  ".join(line), flush=True)
    del a, b, out
