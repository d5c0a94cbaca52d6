#!/usr/bin/env python3
"""Where a gemm256 K-loop iteration spends its time: needs a -DYAT_GEMM_STAMPS build (scripts/build_variant.py stamps
gemm256.hip -DYAT_GEMM_STAMPS) passed as YAT_HIP_LIB.  Prints, per wave group, the s_memtime ticks per iteration in each
of the 8 slots (LOAD / rendezvous / COMPUTE / rendezvous for the two 32-deep sub-steps).  Diagnostic only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops, lib as L
BF, dev = torch.bfloat16, "cuda"
M, D, Hc = 8192, 2240, 5600
SHAPES = [("qkv_fwd", "nt", M, 3 * D, D), ("out_fwd", "nt", M, D, D), ("inv_fwd", "nt", M, 2 * Hc, D),
          ("inv_dgrad", "nn", M, D, 2 * Hc), ("qkv_wgrad", "tn", 3 * D, D, M), ("inv_wgrad", "tn", 2 * Hc, D, M)]
lib = L.load()
fn = lib.yat_debug_gemm_stamps
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint32 * 72)()
names = ["L0", "b", "C0", "b", "L1", "b", "C1", "b"]
for name, lay, m, n, k in SHAPES:
    a_t, b_t = lay == "tn", lay in ("nn", "tn")
    a = (torch.randn((k, m) if a_t else (m, k), device=dev) * 0.5).to(BF)
    b = (torch.randn((k, n) if b_t else (n, k), device=dev) * 0.05).to(BF)
    out = torch.empty(m, n, dtype=BF, device=dev)
    for _ in range(5):
        if os.environ.get("STAMPS_COLD"):                 # operands from HBM: evict the Infinity Cache before every launch
            if "junk" not in globals():
                junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            junk.fill_(1)
        ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k)
    torch.cuda.synchronize()
    assert fn(ctypes.addressof(buf)) == 0
    nt = buf[64]
    print(f"{name:10s} {lay} workgroup 0, wave 0: prologue {buf[65]} ticks, K loop {buf[66]} ({buf[66] / nt:.0f}/iteration), epilogue {buf[67]}")
    for g in (0, 1):
        per = [sum(buf[w * 8 + s] for w in range(4 * g, 4 * g + 4)) / 4 / nt for s in range(8)]
        print(f"{name:10s} {lay} group{g} nt={nt:3d} ticks/iter: " + " ".join(f"{n_}{v:6.0f}" for n_, v in zip(names, per))
              + f" | total {sum(per):6.0f}  mfma-floor {8 * (n >= 2240 and 5 or 4) * 2 * 16}")
