#!/usr/bin/env python3
"""Do two GEMM streams slow each other down?  Times a dgrad shape and a wgrad shape of the SANA block alone (back to back on
one stream) and together (one stream each), same launches, and prints aggregate TFLOP/s both ways -- the backward of the step
runs exactly this pair all the time.  Also a GEMM beside a memory-bound kernel (the AdamW pass over a 1 GB buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
M, D, Hc = 8192, 2240, 5600


def mk(lay, m, n, k):
    a_t, b_t = lay == "tn", lay in ("nn", "tn")
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    return lambda: ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k), 2.0 * m * n * k


pairs = [("inv_dgrad", mk("nn", M, D, 2 * Hc), "inv_wgrad", mk("tn", 2 * Hc, D, M)),
         ("qkv_dgrad", mk("nn", M, D, 3 * D), "qkv_wgrad", mk("tn", 3 * D, D, M)),
         ("inv_fwd half batch", mk("nt", M // 2, 2 * Hc, D), "qkv_fwd half batch", mk("nt", M // 2, 3 * D, D))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REP = 12
for conc in (1, 2):
    ops.gemm_concurrency(conc)
    for n1, (f1, fl1), n2, (f2, fl2) in pairs:
        for _ in range(3):
            f1(); f2()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            f1(); f2()
        e1.record(); torch.cuda.synchronize()
        alone = e0.elapsed_time(e1)
        e0.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(REP):
                f1()
        with torch.cuda.stream(s2):
            for _ in range(REP):
                f2()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        both = e0.elapsed_time(e1)
        tf = lambda ms: REP * (fl1 + fl2) / (ms * 1e-3) / 1e12
        print(f"policy streams={conc}: {n1} + {n2}: one stream {alone / REP * 1e3:7.1f} us/pair = {tf(alone):6.0f} TF/s | "
              f"two streams {both / REP * 1e3:7.1f} us/pair = {tf(both):6.0f} TF/s")
