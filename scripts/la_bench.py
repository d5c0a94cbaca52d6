#!/usr/bin/env python3
"""Linear attention (SANA self-attention, 70 heads x 32) forward / backward alone at the bench's size: back to back on the same
buffers and after a 1 GiB fill (operands from HBM, as in the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF, dev = torch.bfloat16, "cuda"
B, N, H, D = 8, 1024, 70, 2240
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(B * N, 3 * D, device=dev, generator=g).to(BF)
out = torch.empty(B * N, D, dtype=BF, device=dev)
dout = torch.randn(B * N, D, device=dev, generator=g).to(BF)
dqkv = torch.empty_like(qkv)
ws = torch.empty(ops.linear_attn_workspace_bytes(B, N, H), dtype=torch.uint8, device=dev)
st = torch.empty(B * H * 33 * 32, dtype=torch.float32, device=dev)
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def t(fn, cold):
    xs = []
    for _ in range(7):
        if cold:
            junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        xs.append(e0.elapsed_time(e1) * 1e3)
    return sorted(xs)[3]


fwd = lambda: ops.linear_attn_fwd(qkv, B, N, H, D, 2 * D, out, st)
bwd = lambda: ops.linear_attn_bwd(qkv, B, N, H, D, 2 * D, dout, dqkv, ws, state=st)
fwd(); bwd(); torch.cuda.synchronize()
fb, bb = 2 * (B * N * 3 * D + B * N * D), 2 * (B * N * 3 * D * 2 + B * N * D)        # bytes: fwd reads qkv (k, v twice) + writes out
print(f"linear attention  fwd hot {t(fwd, False):6.1f} us  cold {t(fwd, True):6.1f} us   bwd hot {t(bwd, False):6.1f} us  cold {t(bwd, True):6.1f} us"
      f"   (fwd {fb / 1e6:.0f} MB, bwd {bb / 1e6:.0f} MB)")
import hashlib
print("hash", hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12], hashlib.sha1(dqkv.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12])
