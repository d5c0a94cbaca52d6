#!/usr/bin/env python3
"""Micro-benchmark of the GLU depthwise-conv kernels at SANA-1.6B shapes (B=8, Hc=5600) for each aspect bucket.
Prints per-call times and output checksums (to compare kernel variants; parity itself lives in tests/)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF = torch.bfloat16
dev = "cuda"
B, Hc = 8, 5600

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator(device=dev).manual_seed(0)
for (h, w) in ((32, 32), (16, 64), (24, 42), (44, 22)):
    M = B * h * w
    z = torch.randn(M, 2 * Hc, generator=g, device=dev).to(BF)
    s = torch.nn.functional.silu(z.float()).to(BF)
    wdw = (torch.randn(2 * Hc, 9, generator=g, device=dev) * 0.3).to(BF)
    bdw = (torch.randn(2 * Hc, generator=g, device=dev) * 0.1).to(BF)
    dy = torch.randn(M, Hc, generator=g, device=dev).to(BF)
    y = torch.empty(M, Hc, dtype=BF, device=dev)
    dz = torch.empty(M, 2 * Hc, dtype=BF, device=dev)
    dw, db = torch.empty_like(wdw), torch.empty_like(bdw)
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc), dtype=torch.uint8, device=dev)
    u = torch.empty(M, 2 * Hc, dtype=BF, device=dev)
    du = torch.randn(M, 2 * Hc, generator=g, device=dev).to(BF)
    dzs = torch.empty(2 * Hc, dtype=BF, device=dev)
    fu = timeit(lambda: ops.dwconv_glu_fwd(s, B, h, w, Hc, wdw, bdw, y, u_out=u))       # the step's configuration: keeps u
    b2 = timeit(lambda: ops.dwconv_glu_bwd(s, z, B, h, w, Hc, wdw, bdw, None, dz, dw, db, ws, dz_colsum=dzs, du=du))   # pass 2 only
    import hashlib
    hsh = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes() + u.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"dwconv {h:2d}x{w:2d}: sha1(y,u)={hsh}")
    print(f"dwconv {h:2d}x{w:2d}: in-step config  fwd+u={fu:7.1f}us ({(M * 2 * Hc * 2 + M * Hc) * 2 / fu / 1e6:5.2f} TB/s)  "
          f"bwd2(du given)={b2:7.1f}us ({(M * 2 * Hc * 4) * 2 / b2 / 1e6:5.2f} TB/s)", flush=True)
    f = timeit(lambda: ops.dwconv_glu_fwd(s, B, h, w, Hc, wdw, bdw, y))
    b_ = timeit(lambda: ops.dwconv_glu_bwd(s, z, B, h, w, Hc, wdw, bdw, dy, dz, dw, db, ws))
    alg_f = (M * 2 * Hc + M * Hc) * 2
    alg_b = (M * 2 * Hc * 2 + M * Hc + M * 2 * Hc * 2 + M * 2 * Hc) * 2      # bwd1: s,dy -> du ; bwd2: du,s,z -> dz
    print(f"dwconv {h:2d}x{w:2d}: fwd={f:7.1f}us ({alg_f / f / 1e6:5.2f} TB/s)  bwd={b_:7.1f}us ({alg_b / b_ / 1e6:5.2f} TB/s)  "
          f"chk y={y.float().sum().item():.3f} dz={dz.float().sum().item():.3f} dw={dw.float().sum().item():.3f} "
          f"db={db.float().sum().item():.3f}", flush=True)
