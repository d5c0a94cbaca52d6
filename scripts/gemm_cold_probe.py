#!/usr/bin/env python3
"""Hot vs cold operands: each GEMM timed alone (HIP events around ONE launch) right after a launch of itself (operands in the
256 MB Infinity Cache where they fit) and right after a 1 GiB fill that evicts them (operands from HBM, as in the step, where
a weight gradient's operands were written or last read milliseconds earlier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
# (layout, M, N, K): tt = weight gradient, nt = input gradient, nn = forward  (bench.py's naming)
SHAPES = [("tt", 11200, 2240, 8192), ("tt", 2240, 5600, 8192), ("tt", 6720, 2240, 8192), ("tt", 2240, 2240, 8192),
          ("nn", 8192, 11200, 2240), ("nn", 8192, 6720, 2240), ("nn", 8192, 2240, 5600), ("nt", 8192, 2240, 11200),
          ("nt", 8192, 5600, 2240), ("nt", 8192, 2240, 6720),
          ("tt", 4608, 1536, 32768), ("tt", 1536, 6144, 32768), ("nn", 32768, 4608, 1536), ("nt", 32768, 1536, 4608)]
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def one(f, cold, warm=()):
    if cold:
        junk.fill_(1)
        for w in warm:                      # bring ONE operand back (a read pass over it) before the timed launch
            w.view(torch.int16).sum()
    else:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for lay, m, n, k in SHAPES:
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    VARIANT = int(os.environ.get("PROBE_VARIANT", "0"))       # 0 policy, 1 = 128 x 128 kernel (two workgroups per CU), 4 / 5 = 256-row tiles
    f = lambda: ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=VARIANT)
    for _ in range(2): f()
    torch.cuda.synchronize()
    hot = sorted(one(f, False) for _ in range(5))[2]
    cold = sorted(one(f, True) for _ in range(5))[2]
    cold_a = sorted(one(f, True, (a,)) for _ in range(5))[2]       # only A re-read after the eviction
    cold_b = sorted(one(f, True, (b,)) for _ in range(5))[2]       # only B
    fl = 2.0 * m * n * k
    print(f"{lay} {m:6d}x{n:6d}x{k:6d}: hot {hot:7.1f} us {fl / hot / 1e6:6.0f} TF   cold {cold:7.1f} us {fl / cold / 1e6:6.0f} TF   "
          f"+{100 * (cold / hot - 1):4.1f} %   A warm {cold_a:7.1f} us   B warm {cold_b:7.1f} us   A {2e-6 * m * k:6.1f} MB  B {2e-6 * k * n:6.1f} MB", flush=True)
    del a, b, out
