#!/usr/bin/env python3
"""Micro-benchmark of the softmax-attention kernels at PixArt-Sigma shapes (B=8, 16 heads x 72): self-attention over
N = T = 4096 tokens (fused [3D] projection) and cross-attention over T = 300 padded T5 keys; fwd / dQ / dK+dV separately,
with a correctness check of the forward and of the gradients against torch SDPA on one (batch, head)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF = torch.bfloat16
dev = "cuda"
B, N, H, dh = 8, 4096, 16, 72
D = H * dh
sc = 1 / math.sqrt(dh)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def check(q, k, v, T, bias, out, dout, dq, dk, dv, b=1, h=3):
    sl = slice(h * dh, (h + 1) * dh)
    qq = q[b * N:(b + 1) * N, sl].float().requires_grad_(True)
    kk = k[b * T:(b + 1) * T, sl].float().requires_grad_(True)
    vv = v[b * T:(b + 1) * T, sl].float().requires_grad_(True)
    o = torch.softmax(qq @ kk.T * sc + bias[b][None].float(), -1) @ vv
    o.backward(dout[b * N:(b + 1) * N, sl].float())
    rel = lambda a, r: ((a.float() - r).norm() / r.norm()).item()
    return (rel(out[b * N:(b + 1) * N, sl], o.detach()), rel(dq[b * N:(b + 1) * N, sl], qq.grad),
            rel(dk[b * T:(b + 1) * T, sl], kk.grad), rel(dv[b * T:(b + 1) * T, sl], vv.grad))


g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device=dev, generator=g).to(BF)
out = torch.empty(B * N, D, dtype=BF, device=dev); dout = torch.randn(B * N, D, device=dev, generator=g).to(BF)
lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
dqkv = torch.empty_like(qkv)
zero = torch.zeros(B, N, device=dev); full = torch.full((B,), N, dtype=torch.int32, device=dev)
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
flops = 4.0 * B * H * N * N * dh
f = timeit(lambda: ops.sdpa_fwd(q, k, v, B, N, N, H, dh, sc, zero, full, out, lse))
a = (q, k, v, B, N, N, H, dh, sc, zero, full, out, dout, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
t_dq = timeit(lambda: ops.sdpa_bwd(*a, parts=1))
t_dkv = timeit(lambda: ops.sdpa_bwd(*a, parts=2))
print(f"self  N=T=4096: fwd={f:8.1f}us ({flops / f / 1e6:6.1f} TF)  dq={t_dq:8.1f}us ({1.5 * flops / t_dq / 1e6:6.1f} TF)  "
      f"dkv={t_dkv:8.1f}us ({2.0 * flops / t_dkv / 1e6:6.1f} TF)", flush=True)
print("self  rel err (out, dq, dk, dv) vs fp32 softmax:", ["%.2e" % e for e in
      check(q, k, v, N, zero, out, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])], flush=True)
# the same without a key bias (what the models launch for self-attention): the no-bias instantiations
f = timeit(lambda: ops.sdpa_fwd(q, k, v, B, N, N, H, dh, sc, None, None, out, lse))
a = (q, k, v, B, N, N, H, dh, sc, None, None, out, dout, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
t_dq = timeit(lambda: ops.sdpa_bwd(*a, parts=1))
t_dkv = timeit(lambda: ops.sdpa_bwd(*a, parts=2))
print(f"self  N=T=4096 no bias: fwd={f:8.1f}us ({flops / f / 1e6:6.1f} TF)  dq={t_dq:8.1f}us ({1.5 * flops / t_dq / 1e6:6.1f} TF)  "
      f"dkv={t_dkv:8.1f}us ({2.0 * flops / t_dkv / 1e6:6.1f} TF)", flush=True)
print("self  no bias rel err (out, dq, dk, dv) vs fp32 softmax:", ["%.2e" % e for e in
      check(q, k, v, N, zero, out, dout, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])], flush=True)
# SD3.5-Medium's joint attention: 24 heads x 64 over 4096 + 333 tokens
Hs, dhs, Ls = 24, 64, 4429
Ds = Hs * dhs
qkv_s = torch.randn(B * Ls, 3 * Ds, device=dev, generator=g).to(BF)
out_s = torch.empty(B * Ls, Ds, dtype=BF, device=dev); dout_s = torch.randn(B * Ls, Ds, device=dev, generator=g).to(BF)
lse_s = torch.empty(B, Hs, Ls, device=dev); delta_s = torch.empty(B, Hs, Ls, device=dev)
dqkv_s = torch.empty_like(qkv_s)
scs = 1 / math.sqrt(dhs)
fl_s = 4.0 * B * Hs * Ls * Ls * dhs
for tag, bias_s, len_s in (("zero bias", torch.zeros(B, Ls, device=dev), torch.full((B,), Ls, dtype=torch.int32, device=dev)),
                           ("no bias  ", None, None)):
    qs_, ks_, vs_ = qkv_s[:, :Ds], qkv_s[:, Ds:2 * Ds], qkv_s[:, 2 * Ds:]
    f = timeit(lambda: ops.sdpa_fwd(qs_, ks_, vs_, B, Ls, Ls, Hs, dhs, scs, bias_s, len_s, out_s, lse_s))
    a = (qs_, ks_, vs_, B, Ls, Ls, Hs, dhs, scs, bias_s, len_s, out_s, dout_s, lse_s, delta_s, dqkv_s[:, :Ds], dqkv_s[:, Ds:2 * Ds],
         dqkv_s[:, 2 * Ds:])
    t_dq = timeit(lambda: ops.sdpa_bwd(*a, parts=1))
    t_dkv = timeit(lambda: ops.sdpa_bwd(*a, parts=2))
    print(f"sd3.5 joint L=4429 dh=64 {tag}: fwd={f:8.1f}us ({fl_s / f / 1e6:6.1f} TF)  dq={t_dq:8.1f}us ({1.5 * fl_s / t_dq / 1e6:6.1f} TF)  "
          f"dkv={t_dkv:8.1f}us ({2.0 * fl_s / t_dkv / 1e6:6.1f} TF)", flush=True)
del qkv_s, out_s, dout_s, dqkv_s

T = 300
lens = [20, 64, 100, 160, 200, 256, 300, 130]
q2 = torch.randn(B * N, D, device=dev, generator=g).to(BF)
kv = torch.randn(B * T, 2 * D, device=dev, generator=g).to(BF)
mask = torch.zeros(B, T)
for b, L in enumerate(lens): mask[b, :L] = 1
bias = ((1 - mask) * -9984.0).to(dev)
kvl = torch.tensor(lens, dtype=torch.int32, device=dev)
work = ops.kv_work_list(lens, T, dev)
dq2 = torch.empty_like(q2); dkv = torch.empty_like(kv)
f = timeit(lambda: ops.sdpa_fwd(q2, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, lse))
a = (q2, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, dout, lse, delta, dq2, dkv[:, :D], dkv[:, D:])
t_dq = timeit(lambda: ops.sdpa_bwd(*a, work=work, parts=1))
t_dkv = timeit(lambda: ops.sdpa_bwd(*a, work=work, parts=2))
print(f"cross T=300 (mixed lens): fwd={f:8.1f}us  dq={t_dq:8.1f}us  dkv={t_dkv:8.1f}us", flush=True)
print("cross rel err (out, dq, dk, dv):", ["%.2e" % e for e in check(q2, kv[:, :D], kv[:, D:], T, bias, out, dout, dq2, dkv[:, :D], dkv[:, D:], b=6)])
