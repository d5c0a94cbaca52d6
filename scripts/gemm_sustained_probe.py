#!/usr/bin/env python3
"""Hot vs cold operands at the SUSTAINED clock.  scripts/gemm_cold_probe.py times one launch after an idle gap (the chip at its
boost clock); this probe runs each mode back to back for ~1 s first (the chip settles at the clock it holds under the power
cap) and then times ~0.5 s of launches with one pair of HIP events:

  hot        the same (A, B, C) every launch -- operands and output live in the 256 MB Infinity Cache
  cold       R rotating (A, B, C) sets, > 1.3 GB in all -- every launch reads its operands from HBM and writes a cold output
  cold_in    rotating A, B; the same C                  -- only the reads are cold
  cold_out   the same A, B; rotating C                  -- only the writes are cold
  cold_a / cold_b  (PROBE_AB=1) rotating A only / B only, the same C

Diagnostic only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
SHAPES = [("tt", 2240, 5600, 8192), ("tt", 6720, 2240, 8192), ("tt", 11200, 2240, 8192), ("nn", 8192, 11200, 2240),
          ("nn", 8192, 2240, 5600), ("nn", 8192, 6720, 2240), ("nt", 8192, 5600, 2240), ("nt", 8192, 2240, 11200),
          ("nt", 8192, 2240, 6720), ("nt", 8192, 2240, 2240)]
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["PROBE_SHAPES"].split(",")]
if os.environ.get("PROBE_CUSTOM"):               # "nn:32768:2240:320,nt:32768:320:2240"
    SHAPES = [(c.split(":")[0], *map(int, c.split(":")[1:])) for c in os.environ["PROBE_CUSTOM"].split(",")]
VARIANT = int(os.environ.get("PROBE_VARIANT", "0"))
WARM_S, TIME_S = float(os.environ.get("PROBE_WARM_S", "1.0")), float(os.environ.get("PROBE_TIME_S", "0.5"))


def run(sets_in, sets_out, m, n, k, a_t, b_t, est_us):
    def launch(i):
        a, b = sets_in[i % len(sets_in)]
        ops.gemm(a, b, sets_out[i % len(sets_out)], a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=VARIANT)
    nw, nt = max(8, int(WARM_S * 1e6 / est_us)), max(8, int(TIME_S * 1e6 / est_us))
    for i in range(nw):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(nt):
        launch(nw + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / nt


print(f"# sustained probe: {WARM_S} s settle + {TIME_S} s timed per mode, variant {VARIANT}; us per launch (TFLOP/s)")
for lay, m, n, k in SHAPES:
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    per_set = 2 * (m * k + k * n + m * n)
    R = max(3, -(-1400_000_000 // per_set))
    zero = float(os.environ.get("PROBE_ZERO", "0"))          # 1: all-zero operands (same instructions, no toggling: the power test)
    ins = [((torch.randn((k, m) if a_t else (m, k), device=dev) * (0.0 if zero else 0.5)).to(BF),
            (torch.randn((k, n) if b_t else (n, k), device=dev) * (0.0 if zero else 0.05)).to(BF)) for _ in range(R)]
    outs = [torch.empty(m, n, dtype=BF, device=dev) for _ in range(R)]
    fl = 2.0 * m * n * k
    est = fl / 1.0e9          # us at 1000 TF/s
    res = {}
    modes = [("hot", ins[:1], outs[:1]), ("cold", ins, outs), ("cold_in", ins, outs[:1]), ("cold_out", ins[:1], outs)]
    if os.environ.get("PROBE_AB"):            # which operand's coldness costs: rotate only A / only B
        modes = [("hot", ins[:1], outs[:1]), ("cold", ins, outs), ("cold_a", [(a, ins[0][1]) for a, _ in ins], outs[:1]),
                 ("cold_b", [(ins[0][0], b) for _, b in ins], outs[:1])]
    if os.environ.get("PROBE_MODES"):         # e.g. "hot" or "cold": one mode per process (PMC passes: one counter set per mode)
        modes = [md for md in modes if md[0] in os.environ["PROBE_MODES"].split(",")]
    for mode, si, so in modes:
        res[mode] = run(si, so, m, n, k, a_t, b_t, est)
    res.setdefault("hot", float("nan")); res.setdefault("cold", float("nan"))
    print(f"{lay} {m:6d}x{n:6d}x{k:6d} R={R}: " + "   ".join(f"{md} {us:7.1f} us ({fl / us / 1e6:5.0f})" for md, us in res.items())
          + f"   cold/hot +{100 * (res['cold'] / res['hot'] - 1):.1f} %", flush=True)
    del ins, outs
    torch.cuda.empty_cache()
