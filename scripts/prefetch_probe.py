#!/usr/bin/env python3
"""Would pulling the NEXT GEMM's weights into the Infinity Cache while the current GEMM runs pay?  Pairs of forward GEMMs of
a SANA block (QKV 8192 x 6720 x 2240, then conv_inverted 8192 x 11200 x 2240) over R rotating weight sets (cold weights, as in
the step), activations shared (hot, as in the step).  Timed: the pair alone; the pair with a strided read of the second
GEMM's weight (one word per 128-byte line) on a second stream beside the FIRST GEMM.  Diagnostic only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
M, D = 8192, 2240
SHAPES = [(6720, 11200), (2240, 11200), (11200, 2240)]
R = 24
STRIDE = int(os.environ.get("PREFETCH_STRIDE", "32"))
x = (torch.randn(M, D, device=dev) * 0.5).to(BF)
side = torch.cuda.Stream()


def run(n1, n2, prefetch, reps=60):
    w1 = [(torch.randn(n1, D, device=dev) * 0.05).to(BF) for _ in range(R)]
    k2 = D if n2 != 2240 or True else D
    w2 = [(torch.randn(n2, D, device=dev) * 0.05).to(BF) for _ in range(R)]
    o1 = torch.empty(M, n1, dtype=BF, device=dev)
    o2 = torch.empty(M, n2, dtype=BF, device=dev)
    sink = torch.zeros(1, dtype=torch.int32, device=dev)

    def pair(i):
        a, b = w1[i % R], w2[i % R]
        if prefetch:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                sink.add_(b.view(torch.int32).view(-1)[::STRIDE].sum(dtype=torch.int32))       # STRIDE 32: one word per 128-B line; 1: every byte
        ops.gemm(x, a, o1, M=M, N=n1, K=D)
        ops.gemm(x, b, o2, M=M, N=n2, K=D)
    for i in range(12):
        pair(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        pair(12 + i)
    torch.cuda.current_stream().wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n1, n2 in SHAPES:
    for rnd in range(2):
        t0 = run(n1, n2, False)
        t1 = run(n1, n2, True)
        print(f"pair nn {M}x{n1}x{D} -> nn {M}x{n2}x{D}: alone {t0:7.1f} us | with the second weight prefetched beside the first GEMM "
              f"{t1:7.1f} us ({100 * (t1 / t0 - 1):+.1f} %)", flush=True)
