#!/usr/bin/env python3
"""Round 6 (review item 3, tile tails): would splitting the rows of the three 2.25 / 4.375 / 2.625-round forward-layout GEMMs into
"whole rounds of 256 x 320 tiles" + "a remainder on whatever the policy picks for it" shorten the launch?  Times the whole shape,
the whole-round part and the remainder (plain epilogues; the SiLU + aux form for the 11200-wide one) alone, one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, act in (("conv_inverted fwd", 8192, 11200, 2240, "silu"), ("qkv fwd", 8192, 6720, 2240, "none")):
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    bias = torch.randn(N, device=dev).to(BF)
    out, aux = torch.empty(M, N, dtype=BF, device=dev), torch.empty(M, N, dtype=BF, device=dev)
    nbn = (N + 319) // 320
    rounds = (M // 256) * nbn / 256.0
    full_rows = int(int(rounds) * 256 / nbn) * 256              # row tiles that make whole rounds
    kw = dict(bias=bias, activation=act, aux_out=aux if act != "none" else None)

    def run(r0, r1):
        ops.gemm(a[r0:r1], b, out[r0:r1], M=r1 - r0, N=N, K=K, **{k: (v[r0:r1] if k == "aux_out" and v is not None else v)
                                                                  for k, v in kw.items()})
    whole = t(lambda: run(0, M))
    part = t(lambda: run(0, full_rows))
    rest = t(lambda: run(full_rows, M))
    both = t(lambda: (run(0, full_rows), run(full_rows, M)))
    print(f"{name} {M}x{N}x{K}: {rounds:.3f} rounds of 256x320 tiles; whole launch {whole:.1f} us; rows [0,{full_rows}) {part:.1f} us + "
          f"rows [{full_rows},{M}) {rest:.1f} us = {part + rest:.1f}; the two launches back to back {both:.1f} us", flush=True)
