#!/usr/bin/env python3
"""Backward pass 2 of the GLU depthwise convolution with s = SiLU(z) recomputed from z (round 5) against the build that reads
s (-DYAT_DW_S_FROM_Z=0, loaded through YAT_HIP_LIB): SHA-1 of dz / dW / db / the dz column sums and the time per call, per
aspect bucket.  (s, z) come out of ONE GEMM launch (SiLU + pre-activation epilogue), as in the step, so that s is the
library's own bf16(z sigmoid(z)).  Run once per build and diff the listings."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
B, Hc = 8, 5600


def sha(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.contiguous().view(torch.int16).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
for (h, w) in ((32, 32), (16, 64), (24, 42), (44, 22), (7, 9)):
    M = B * h * w
    x = torch.randn(M, 64, generator=g, device=dev).to(BF)
    wt = (torch.randn(2 * Hc, 64, generator=g, device=dev) * 0.25).to(BF)
    s, z = torch.empty(M, 2 * Hc, dtype=BF, device=dev), torch.empty(M, 2 * Hc, dtype=BF, device=dev)
    ops.gemm(x, wt, s, M=M, N=2 * Hc, K=64, activation="silu", aux_out=z)
    wdw = (torch.randn(2 * Hc, 9, generator=g, device=dev) * 0.3).to(BF)
    bdw = (torch.randn(2 * Hc, generator=g, device=dev) * 0.1).to(BF)
    du = torch.randn(M, 2 * Hc, generator=g, device=dev).to(BF)
    dz = torch.empty(M, 2 * Hc, dtype=BF, device=dev)
    dw, db, dzs = torch.empty_like(wdw), torch.empty_like(bdw), torch.empty(2 * Hc, dtype=BF, device=dev)
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc), dtype=torch.uint8, device=dev)
    run = lambda: ops.dwconv_glu_bwd(s, z, B, h, w, Hc, wdw, bdw, None, dz, dw, db, ws, dz_colsum=dzs, du=du)
    us = timeit(run)
    torch.cuda.synchronize()
    print(f"dw bwd2 {h:2d}x{w:2d}: {sha(dz, dw, db, dzs)}  {us:7.1f} us", flush=True)
