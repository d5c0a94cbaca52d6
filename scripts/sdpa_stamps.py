#!/usr/bin/env python3
"""Where a key / query tile of the attention backward kernels spends its time: needs a -DYAT_SDPA_STAMPS build
(scripts/build_variant.py stamps sdpa.hip -DYAT_SDPA_STAMPS) passed as YAT_HIP_LIB.  Prints, for the four waves of one workgroup,
the s_memtime ticks per tile in each loop segment.  Diagnostic only."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops, lib as L
BF, dev = torch.bfloat16, "cuda"
lib = L.load()
fn = lib.yat_debug_sdpa_stamps
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint32 * 136)()
FWD = ["wait+bar", "stage", "S mfma", "softmax", "PV mfma"]
DQ = ["wait+bar", "stage", "S,dP mfma", "softmax", "dQ mfma"]
DKV = ["wait+bar", "stage", "S,dP h0", "softmax h0", "dV,dK h0", "S,dP h1", "softmax h1", "dV,dK h1"]
for name, B, N, H, dh in (("pixart", 8, 4096, 16, 72), ("sd3.5", 8, 4429, 24, 64)):
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, device=dev, generator=g).to(BF)
    out = torch.empty(B * N, D, dtype=BF, device=dev); dout = torch.randn(B * N, D, device=dev, generator=g).to(BF)
    lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    sc = 1 / math.sqrt(dh)
    ops.sdpa_fwd(q, k, v, B, N, N, H, dh, sc, None, None, out, lse)
    a = (q, k, v, B, N, N, H, dh, sc, None, None, out, dout, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
    for part, names in ((0, FWD), (1, DQ), (2, DKV)):
        for _ in range(3):
            if part == 0:
                ops.sdpa_fwd(q, k, v, B, N, N, H, dh, sc, None, None, out, lse)
            else:
                ops.sdpa_bwd(*a, parts=part)
        torch.cuda.synchronize()
        assert fn(ctypes.addressof(buf)) == 0
        n = buf[128]
        print(f"{name} part {part}: loop {buf[129]} s_memtime ticks in {buf[130]} ticks of the 100 MHz clock -> {buf[129] / max(buf[130], 1) / 10:.3f} GHz", flush=True)
        for w in range(4):
            per = [buf[w * 16 + i] / n for i in range(len(names))]
            print(f"{name} {('fwd', 'dq ', 'dkv')[part]} wave {w} tiles {n}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per))
                  + f"  | total {sum(per):6.0f}", flush=True)
