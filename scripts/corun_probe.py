#!/usr/bin/env python3
"""Can a memory-bound kernel share CUs with a gemm256 workgroup?  A 256 x 320 tile kernel holds 2 x ~248 VGPRs per SIMD and
144 KiB of LDS: nothing else fits on its CU, so a second stream only ever gets the CUs a GEMM workgroup has left.  The
256 x 256 tile kernel allocates 2 x 208: 96 registers per SIMD (and 32 KiB of LDS) stay free -- enough for one wave of a
streaming kernel per SIMD.  This probe times REP launches of one GEMM shape beside a torch elementwise pass (few VGPRs, no
LDS) sized to last about as long, on two streams, per tile variant:  alone / alone / together.  Diagnostic only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
SHAPES = [("nn", 8192, 6720, 2240), ("nn", 8192, 11200, 2240), ("nt", 8192, 2240, 11200)]
REP = 16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for lay, m, n, k in SHAPES:
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    for variant in (5, 4):
        def gemms():
            for _ in range(REP):
                ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=variant)
        for _ in range(2):
            gemms()
        t_g = timed(gemms)
        # memory pass sized to ~ the GEMM batch's time at ~4 TB/s (read + write): x.mul_ over nbytes
        nbytes = int(t_g * 1e-6 * 4e12 / 2)
        x = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        chunks = x.chunk(REP)

        def mem():
            for c in chunks:
                c.mul_(1.0000001)
        mem()
        t_m = timed(mem)

        def both():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                gemms()
            with torch.cuda.stream(s2):
                mem()
            cur.wait_stream(s1); cur.wait_stream(s2)
        both()
        t_b = timed(both)
        fl = 2.0 * m * n * k * REP
        print(f"{lay} {m}x{n}x{k} tile 256x{64 * variant}: gemm alone {t_g / REP:7.1f} us ({fl / t_g / 1e6:5.0f} TF/s) | "
              f"stream pass alone {t_m / REP:7.1f} us ({2 * nbytes / t_m / 1e6:5.2f} TB/s) | together {t_b / REP:7.1f} us "
              f"= {100 * t_b / (t_g + t_m):5.1f} % of the sum, {100 * t_b / max(t_g, t_m):5.1f} % of the longer", flush=True)
        del x, chunks
        torch.cuda.empty_cache()
