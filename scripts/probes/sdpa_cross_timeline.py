#!/usr/bin/env python3
"""Round 6: per-workgroup timeline of the SANA cross-attention forward (B = 8, N = 1024, 20 x 112, ragged 20..300 keys of
T = 512) from the -DYAT_SDPA_STAMPS build (YAT_HIP_LIB): entry -> key loop -> loop exit -> kernel exit per workgroup, CU id."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from yat_amd import ops, lib as L
BF, dev = torch.bfloat16, "cuda"
lib = L.load()
fn = lib.yat_debug_sdpa_wg_times
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint32 * (4096 * 6))()
B, N, T, H, dh = 8, 1024, 512, 20, 112
D = H * dh
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B * N, D, device=dev, generator=g).to(BF)
kv = torch.randn(B * T, 2 * D, device=dev, generator=g).to(BF)
out = torch.empty(B * N, D, dtype=BF, device=dev)
lse = torch.empty(B, H, N, device=dev)
for name, lens in (("all64", [64] * B), ("mixed", [20, 64, 100, 160, 200, 256, 300, 130])):
    mask = torch.zeros(B, T)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    bias = ((1 - mask) * -9984.0).to(dev)
    kvl = torch.tensor(lens, dtype=torch.int32, device=dev)
    sc = 1 / math.sqrt(dh)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        ops.sdpa_fwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, lse)
        e1.record()
    torch.cuda.synchronize()
    assert fn(ctypes.addressof(buf)) == 0
    a = np.frombuffer(buf, dtype=np.uint32).reshape(4096, 6).astype(np.int64)
    nwg = (N + 127) // 128 * H * B
    a = a[:nwg]
    valid = a[:, 0] != 0
    print(f"  stamped workgroups: {int(valid.sum())} of {nwg} (first invalid index {int(np.argmin(valid)) if not valid.all() else -1})")
    a = a[valid]
    t0 = a[:, 0].min()
    ent, l0, l1, ex = [(a[:, i] - t0) / 100.0 for i in range(4)]
    print(f"{name}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events; {nwg} workgroups; last exit {ex.max():.1f} us")
    print(f"  per workgroup (us): entry->loop {np.mean(l0 - ent):.2f} (max {np.max(l0 - ent):.2f})   loop {np.mean(l1 - l0):.2f} "
          f"(min {np.min(l1 - l0):.2f} max {np.max(l1 - l0):.2f})   loop->exit {np.mean(ex - l1):.2f} (max {np.max(ex - l1):.2f})")
    cu = (a[:, 4] & 0xffffff00) * 16 + a[:, 5]
    per_cu = {}
    for i in np.argsort(ent):
        per_cu.setdefault(int(cu[i]), []).append((ent[i], ex[i]))
    busy = [sum(e - s for s, e in v) for v in per_cu.values()]
    print(f"  distinct CU ids {len(per_cu)}; workgroups per CU min {min(map(len, per_cu.values()))} max {max(map(len, per_cu.values()))}; "
          f"sum of lifetimes per CU mean {np.mean(busy):.1f} us (max {np.max(busy):.1f})")
    order = np.sort(ent)
    print("  entries by time (us): " + " ".join(f"{order[int(x * (len(order) - 1))]:.1f}" for x in (0, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1)))
    # how many workgroups are alive at a time, chip-wide (every 2 us)
    alive = [int(((ent <= t) & (ex > t)).sum()) for t in np.arange(0, ex.max(), 2.0)]
    print("  alive workgroups every 2 us: " + " ".join(map(str, alive)))
    exo = np.sort(ex)
    print("  exits by time (us):   " + " ".join(f"{exo[int(x * (len(exo) - 1))]:.1f}" for x in (0, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1)))
