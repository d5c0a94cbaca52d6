#!/usr/bin/env python3
"""Energy view of the gemm256 main loop: the product kernel against the ablation builds (scripts/build_variant.py ...
-DYAT_ABL_NO_DMA / -DYAT_ABL_NO_LDSREAD: results are WRONG, only rate, clock and power are read) on one forward and one
weight-gradient shape, each in a 3 s loop beside rocm-smi.  What a piece of the loop costs = what the chip gains in FLOP/s
under the same power cap when that piece is left out.  Run once per library: YAT_HIP_LIB=<variant> python this.py NAME."""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
samples, stop = [], False


def sampler():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True,
                                          text=True, timeout=5).stdout)["card0"]
            samples.append((time.time(), int(re.search(r"(\d+)", d["sclk clock speed:"]).group(1)),
                            float(next(v for k, v in d.items() if "Power" in k))))
        except Exception:
            pass
        time.sleep(0.25)


threading.Thread(target=sampler, daemon=True).start()
name = sys.argv[1] if len(sys.argv) > 1 else "product"
M, D, Hc = 8192, 2240, 5600
for label, lay, m, n, k in (("inv_fwd nt 8192x11200x2240", "nt", M, 2 * Hc, D), ("inv_wgrad tn 11200x2240x8192", "tn", 2 * Hc, D, M)):
    a_t, b_t = lay == "tn", lay in ("nn", "tn")
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=5); torch.cuda.synchronize()
    t0 = time.time(); it = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(200):
            ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=5)
        it += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    t1 = time.time()
    tf = 2.0 * m * n * k * it / (e0.elapsed_time(e1) * 1e-3) / 1e12
    w = [(s, p) for t, s, p in samples if t0 + 0.8 <= t <= t1 - 0.2]
    s_, p_ = sorted(x[0] for x in w), sorted(x[1] for x in w)
    sc, pw = (s_[len(s_) // 2], p_[len(p_) // 2]) if w else (0, 0)
    print(f"{name:10s} {label}: {tf:7.1f} TFLOP/s  sclk {sc} MHz  power {pw:.0f} W  -> {pw / tf:.3f} pJ/FLOP all in", flush=True)
stop = True
