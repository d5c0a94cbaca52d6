#!/usr/bin/env python3
"""One GEMM shape at a time, in a loop for ~3 s each, with rocm-smi sampled beside it: TFLOP/s, shader clock and socket power
of the kernel when it has the chip to itself -- is a lone GEMM already at the power cap?  Also the
clip + AdamW stream and an idle baseline."""
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
            sclk = int(re.search(r"(\d+)", d["sclk clock speed:"]).group(1))
            pw = float(next(v for k, v in d.items() if "Power" in k))
            samples.append((time.time(), sclk, pw))
        except Exception:
            pass
        time.sleep(0.25)


threading.Thread(target=sampler, daemon=True).start()


def window(t0, t1):
    w = [(s, p) for t, s, p in samples if t0 + 0.8 <= t <= t1 - 0.2]
    if not w:
        return "no samples"
    s = sorted(x[0] for x in w); p = sorted(x[1] for x in w)
    return f"sclk {s[len(s) // 2]:4d} MHz  power {p[len(p) // 2]:5.0f} W  ({len(w)} samples)"


M, D, Hc = 8192, 2240, 5600
SHAPES = [("inv_fwd   nt 8192x11200x2240", "nt", M, 2 * Hc, D, 0), ("inv_dgrad nn 8192x2240x11200", "nn", M, D, 2 * Hc, 0),
          ("inv_wgrad tn 11200x2240x8192", "tn", 2 * Hc, D, M, 0), ("out_fwd   nt 8192x2240x2240 ", "nt", M, D, D, 0),
          ("inv_fwd   nt  .. 128x128 tile", "nt", M, 2 * Hc, D, 1), ("inv_fwd   nt  .. 256x256 tile", "nt", M, 2 * Hc, D, 4),
          ("inv_wgrad tn  .. 256x256 tile", "tn", 2 * Hc, D, M, 4)]
time.sleep(1.5)
t0 = time.time(); time.sleep(2.0); print("idle                          :", window(t0 - 0.8, time.time() + 0.2), flush=True)
for name, lay, m, n, k, var in SHAPES:
    a_t, b_t = lay == "tn", lay in ("nn", "tn")
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=var); torch.cuda.synchronize()
    t0 = time.time(); it = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(200):
            ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=var)
        it += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    tf = 2.0 * m * n * k * it / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"{name}: {tf:7.1f} TFLOP/s  {window(t0, time.time())}", flush=True)
    del a, b, out
# HBM stream: a 3.2 GB bf16 copy-like pass (clip + AdamW shaped traffic is in bench.py; here a plain elementwise add)
x = torch.empty(1 << 30, dtype=BF, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
t0 = time.time(); it = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 3.0:
    for _ in range(20):
        ops.add_bf16(x, y, z)
    it += 20; torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print(f"add_bf16 stream (6 B/elem)    : {6.0 * x.numel() * it / (e0.elapsed_time(e1) * 1e-3) / 1e12:7.2f} TB/s     {window(t0, time.time())}", flush=True)
stop = True
