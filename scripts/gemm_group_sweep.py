#!/usr/bin/env python3
"""Tile-order sweep: for every GEMM shape of a SANA block (B = 8) the default policy's kernel with the row-group size of the
XCD-contiguous tile order forced to 1 / 2 / 4 / 8 / 16 / 32 (policy word + 1000000 * group), interleaved rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
M, Mt, D, Hc = 8192, 1536, 2240, 5600
SHAPES = [("qkv_fwd", "nn", M, 3 * D, D), ("out_fwd", "nn", M, D, D), ("kv_fwd", "nn", Mt, 2 * D, D), ("inv_fwd", "nn", M, 2 * Hc, D),
          ("point_fwd", "nn", M, D, Hc), ("qkv_dgrad", "nt", M, D, 3 * D), ("out_dgrad", "nt", M, D, D), ("kv_dgrad", "nt", Mt, D, 2 * D),
          ("inv_dgrad", "nt", M, D, 2 * Hc), ("point_dgrad", "nt", M, Hc, D), ("qkv_wgrad", "tt", 3 * D, D, M), ("out_wgrad", "tt", D, D, M),
          ("kv_wgrad", "tt", 2 * D, D, Mt), ("inv_wgrad", "tt", 2 * Hc, D, M), ("point_wgrad", "tt", D, Hc, M)]
GROUPS = (4, 1, 2, 8, 16, 32)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ops.gemm_concurrency(streams)
print(f"# us per launch (median of {rounds} rounds x 3 launches), policy planned for {streams} stream(s); group 4 = the product's order")
print("shape          layout      M      N      K " + " ".join(f"   g{g:<3d}" for g in GROUPS) + "   best")
tot = {g: 0.0 for g in GROUPS}
for name, lay, m, n, k in SHAPES:
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    t = {g: [] for g in GROUPS}
    for r in range(rounds + 1):
        for g in GROUPS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=1000000 * g)
            e1.record()
            torch.cuda.synchronize()
            if r:
                t[g].append(e0.elapsed_time(e1) / 3 * 1e3)
    med = {g: sorted(v)[len(v) // 2] for g, v in t.items()}
    for g in GROUPS:
        tot[g] += med[g]
    best = min(med, key=med.get)
    print(f"{name:14s} {lay:4s} {m:7d} {n:6d} {k:6d} " + " ".join(f"{med[g]:7.1f}" for g in GROUPS) + f"   g{best} ({100 * (med[4] / med[best] - 1):+.1f} %)", flush=True)
print("sum of the block's shapes:                  " + " ".join(f"{tot[g]:7.1f}" for g in GROUPS))
