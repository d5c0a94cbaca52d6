#!/usr/bin/env python3
"""What does a CU mask on a HIP stream do on this chip?  Times a many-tile GEMM on streams created with
hipExtStreamCreateWithCUMask under a few masks (fraction of the 256 CUs enabled, two bit layouts)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
hip = ctypes.CDLL("libamdhip64.so")
BF, dev = torch.bfloat16, "cuda"
torch.cuda.init(); torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


m, n, k = 8192, 11200, 2240
a = (torch.randn(m, k, device=dev) * 0.5).to(BF); b = (torch.randn(n, k, device=dev) * 0.05).to(BF)
out = torch.empty(m, n, dtype=BF, device=dev)
full = (1 << 256) - 1
masks = {"all 256": full, "first 128 bits": (1 << 128) - 1, "first 192 bits": (1 << 192) - 1,
         "3 of every 4 bits": int("7" * 64, 16), "every other bit": int("5" * 64, 16),
         "first 24 of every 32": int("00ffffff" * 8, 16)}
for name, bits in masks.items():
    st = masked_stream(bits)
    with torch.cuda.stream(st):
        for _ in range(3):
            ops.gemm(a, b, out, M=m, N=n, K=k)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            ops.gemm(a, b, out, M=m, N=n, K=k)
        e1.record(st)
        st.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:24s} ({bin(bits).count('1'):3d} CUs): {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.0f} TF/s")
