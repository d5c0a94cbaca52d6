#!/usr/bin/env python3
"""gate backward, LayerNorm-modulate forward / backward (rows, cols) on the SANA activation shape (8 x 1024 x 2240):
microseconds and GB/s of the algorithmic bytes, alone on the chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
B, N, D = 8, 1024, 2240
M = B * N
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(BF)
x, dy, lin, dres = rn(M, D), rn(M, D), rn(M, D), rn(M, D)
mod = rn(B, 6 * D)
dlin, dx, y = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
acc = torch.zeros(B, 6 * D, device=dev)
ws_g = torch.empty(int(ops._lib().yat_gate_bwd_workspace_bytes(M, D, N)), dtype=torch.uint8, device=dev)
ws_l = torch.empty(ops.ln_bwd_workspace_bytes(M, D, N), dtype=torch.uint8, device=dev)
dbias = torch.zeros(D, dtype=BF, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)

def timed(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

act = M * D * 2
cases = [
    ("gate_bwd (+bias)", lambda: ops.gate_bwd(dy, lin, mod[:, :D], 6 * D, N, dlin, acc[:, :D], 6 * D, ws_g, dbias), 3 * act),
    ("gate_bwd", lambda: ops.gate_bwd(dy, lin, mod[:, :D], 6 * D, N, dlin, acc[:, :D], 6 * D, ws_g), 3 * act),
    ("ln_fwd", lambda: ops.ln_modulate_fwd(x, mod[:, D:2 * D], mod[:, 2 * D:3 * D], 6 * D, N, 1e-6, y, mean, rstd), 2 * act),
    ("ln_bwd rows", lambda: ops.ln_modulate_bwd(x, mean, rstd, mod[:, 2 * D:3 * D], 6 * D, N, dy, dres, dx, acc[:, D:2 * D], acc[:, 2 * D:3 * D], 6 * D, ws_l, parts=1), 4 * act),
    ("ln_bwd cols", lambda: ops.ln_modulate_bwd(x, mean, rstd, mod[:, 2 * D:3 * D], 6 * D, N, dy, dres, dx, acc[:, D:2 * D], acc[:, 2 * D:3 * D], 6 * D, ws_l, parts=2), 2 * act),
]
for name, f, nbytes in cases:
    us = timed(f)
    print(f"{name:18s} {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s")
print("checks", dlin.float().sum().item(), acc.sum().item(), dbias.float().sum().item(), dx.float().sum().item())
