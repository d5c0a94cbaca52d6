#!/usr/bin/env python3
"""GEMM tile-variant sweep at PixArt-Sigma shapes (M = B*N = 32768 token rows, D = 1152): forward (NT... 'nn' = x W^T),
dgrad ('nt' = dy W) and wgrad ('tt' = dy^T x) of every Linear of a block, for the 256x256 / 256x320 tiles and the
policy's own choice (variant 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF = torch.bfloat16
dev = "cuda"
M, D = 32768, 1152


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(BF)
print(f"{'op':28s} " + " ".join(f"{'v' + str(v):>16s}" for v in (0, 4, 5)))
for name, nout, nin in (("qkv", 3 * D, D), ("to_out / q2 / out2", D, D), ("ff.net.0 (4D)", 4 * D, D), ("ff.net.2", D, 4 * D)):
    x, w, dy = rnd(M, nin), rnd(nout, nin), rnd(M, nout)
    y, dx, dw = torch.empty(M, nout, dtype=BF, device=dev), torch.empty(M, nin, dtype=BF, device=dev), torch.empty(nout, nin, dtype=BF, device=dev)
    fl = 2.0 * M * nout * nin
    for kind, fn in (("fwd", lambda v: ops.gemm(x, w, y, M=M, N=nout, K=nin, variant=v)),
                     ("dgrad", lambda v: ops.gemm(dy, w, dx, b_t=True, M=M, N=nin, K=nout, variant=v)),
                     ("wgrad", lambda v: ops.gemm(dy, x, dw, a_t=True, b_t=True, M=nout, N=nin, K=M, lda=nout, ldb=nin, ldc=nin, variant=v))):
        cells = []
        for v in (0, 4, 5):
            try:
                t = timeit(lambda: fn(v))
                cells.append(f"{t:7.1f}us {fl / t / 1e6:5.0f}TF")
            except Exception as e:
                cells.append(f"{'n/a':>16s}")
        print(f"{name + ' ' + kind:28s} " + " ".join(f"{c:>16s}" for c in cells), flush=True)
