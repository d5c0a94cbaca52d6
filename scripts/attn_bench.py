#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at SANA-1.6B shapes (B=8, N=1024, T=512): masked cross-attention
fwd / bwd for several kv_len patterns, linear attention fwd / bwd."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF = torch.bfloat16
dev = "cuda"
B, N, T, H, dh = 8, 1024, 512, 20, 112
D = H * dh

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

q = torch.randn(B * N, D, device=dev).to(BF)
kv = torch.randn(B * T, 2 * D, device=dev).to(BF)
out = torch.empty(B * N, D, dtype=BF, device=dev); dout = torch.randn(B * N, D, device=dev).to(BF)
lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
for name, lens in (("all64", [64] * B), ("all160", [160] * B), ("all512", [512] * B), ("mixed", [20, 64, 100, 160, 200, 256, 300, 130])):
    mask = torch.zeros(B, T)
    for b, L in enumerate(lens): mask[b, :L] = 1
    bias = ((1 - mask) * -9984.0).to(dev)
    kvl = torch.tensor(lens, dtype=torch.int32, device=dev)
    sc = 1 / math.sqrt(dh)
    f = timeit(lambda: ops.sdpa_fwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, lse))
    work = ops.kv_work_list(lens, T, dev)
    b_ = timeit(lambda: ops.sdpa_bwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, dout, lse, delta, dq, dkv[:, :D], dkv[:, D:], work=work))
    tiles = sum((L + 63) // 64 for L in lens)
    bq = timeit(lambda: ops.sdpa_bwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, dout, lse, delta, dq, dkv[:, :D], dkv[:, D:], work=work, parts=1))
    print(f"sdpa {name:8s} key-tiles={tiles:3d}  fwd={f:7.1f}us  bwd(dq+dkv)={b_:7.1f}us  dq alone={bq:7.1f}us  dkv={b_ - bq:7.1f}us  "
          f"checksum {dkv.float().abs().sum().item():.4e}", flush=True)
H1, D1 = 70, 2240
qkv = torch.randn(B * N, 3 * D1, device=dev).to(BF)
o1 = torch.empty(B * N, D1, dtype=BF, device=dev); do1 = torch.randn(B * N, D1, device=dev).to(BF); dqkv = torch.empty_like(qkv)
ws = torch.empty(ops.linear_attn_workspace_bytes(B, N, H1), dtype=torch.uint8, device=dev)
st = torch.empty(B * H1 * 33 * 32, dtype=torch.float32, device=dev)
f = timeit(lambda: ops.linear_attn_fwd(qkv, B, N, H1, D1, 2 * D1, o1, st))
b_ = timeit(lambda: ops.linear_attn_bwd(qkv, B, N, H1, D1, 2 * D1, do1, dqkv, ws, state=st))
print(f"linear attention fwd={f:7.1f}us bwd={b_:7.1f}us")
