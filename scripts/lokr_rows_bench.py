#!/usr/bin/env python3
"""Micro-benchmark of the LoKr row-streaming products at config-5 sizes (B = 32: M = 32768 tokens): T1 = x' w2_b^T (forward, plain and
flat layouts) and dx' += H' w2_b, against their HBM traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 32768
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for K, im, n_ in ((2240, 40, 56), (5600, 70, 80), (11200, 100, 112)):
    R = 8
    x = torch.randn(M * im, n_, device=dev).to(BF)
    wb = (torch.randn(R, n_, device=dev) * 0.3).to(BF)
    t1 = torch.empty(M * im, R, dtype=BF, device=dev)
    slab = torch.zeros(M, K, dtype=BF, device=dev)
    h = torch.randn(M * im, R, device=dev).to(BF)
    dx = torch.randn(M * im, n_, device=dev).to(BF)
    mb_f = (x.numel() + t1.numel()) * 2 / 1e6
    mb_b = (2 * dx.numel() + h.numel()) * 2 / 1e6
    def cold(fn):
        def run():
            junk.fill_(1); fn()
        return run
    tj = timeit(lambda: junk.fill_(1))
    f = timeit(lambda: ops.lokr_rows_fwd(x, wb, t1))
    fc = timeit(cold(lambda: ops.lokr_rows_fwd(x, wb, t1))) - tj
    ff = timeit(lambda: ops.lokr_rows_fwd_flat(x, wb, slab[:, :im * R], im))
    b = timeit(lambda: ops.lokr_rows_bwd(h, wb, dx))
    print(f"K={K} in_m={im} in_n={n_}: fwd {f:6.1f} us ({mb_f / f:5.2f} TB/s; cold {fc:6.1f}) flat {ff:6.1f} us | bwd {b:6.1f} us ({mb_b / b:5.2f} TB/s)", flush=True)
