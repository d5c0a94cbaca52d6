#!/usr/bin/env python3
"""Per-workgroup timeline of the attention forward (needs the -DYAT_SDPA_STAMPS build as YAT_HIP_LIB): when each workgroup
entered the kernel, its key loop, left the loop and the kernel (100 MHz clock), and on which CU.  Diagnostic only."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yat_amd import ops, lib as L
BF, dev = torch.bfloat16, "cuda"
lib = L.load()
fn = lib.yat_debug_sdpa_wg_times
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint32 * (4096 * 6))()
for name, B, N, H, dh in (("pixart", 8, 4096, 16, 72), ("sd3.5", 8, 4429, 24, 64)):
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, device=dev, generator=g).to(BF)
    out = torch.empty(B * N, D, dtype=BF, device=dev)
    lse = torch.empty(B, H, N, device=dev)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    sc = 1 / math.sqrt(dh)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        ops.sdpa_fwd(q, k, v, B, N, N, H, dh, sc, None, None, out, lse)
        e1.record()
    torch.cuda.synchronize()
    assert fn(ctypes.addressof(buf)) == 0
    a = np.frombuffer(buf, dtype=np.uint32).reshape(4096, 6).astype(np.int64)
    nwg = (N + 255) // 256 * H * B
    a = a[:min(nwg, 4096)]
    t0 = a[:, 0].min()
    ent, l0, l1, ex = [(a[:, i] - t0) / 100.0 for i in range(4)]          # us
    print(f"{name}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us by events; {len(a)} workgroups; first entry 0, last exit {ex.max():.0f} us")
    print(f"  per workgroup (us): entry->loop {np.mean(l0 - ent):.1f} (max {np.max(l0 - ent):.1f})   loop {np.mean(l1 - l0):.1f} "
          f"(min {np.min(l1 - l0):.1f} max {np.max(l1 - l0):.1f})   loop->exit {np.mean(ex - l1):.1f}")
    cu = (a[:, 4] & 0xffffff00) * 16 + a[:, 5]          # everything of HW_ID above the wave / SIMD bits, plus the XCC
    gaps, per_cu = [], {}
    for i in np.argsort(ent):
        per_cu.setdefault(int(cu[i]), []).append((ent[i], ex[i]))
    print(f"  distinct CU ids {len(per_cu)}; workgroups per CU: min {min(map(len, per_cu.values()))} max {max(map(len, per_cu.values()))}")
    busy = [sum(e - s for s, e in v) for v in per_cu.values()]
    print(f"  sum of workgroup lifetimes per CU: mean {np.mean(busy):.0f} us (min {np.min(busy):.0f} max {np.max(busy):.0f}) against last exit {ex.max():.0f}")
    order = np.sort(ent)
    print("  entries by time: " + " ".join(f"{order[int(q * (len(order) - 1))]:.0f}" for q in (0, .1, .25, .26, .5, .51, .75, .76, .99, 1)))
