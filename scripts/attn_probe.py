import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yat_amd import ops
BF = torch.bfloat16; dev = "cuda"
B, N, H, dh = 8, 1024, 20, 112
D = H * dh
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for T, L in ((64, 64), (128, 128), (512, 64), (512, 512), (256, 256)):
    q = torch.randn(B * N, D, device=dev).to(BF); kv = torch.randn(B * T, 2 * D, device=dev).to(BF)
    out = torch.empty(B * N, D, dtype=BF, device=dev); dout = torch.randn(B * N, D, device=dev).to(BF)
    lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    mask = torch.zeros(B, T); mask[:, :L] = 1
    bias = ((1 - mask) * -9984.0).to(dev); kvl = torch.full((B,), L, dtype=torch.int32, device=dev)
    sc = 1 / math.sqrt(dh)
    ops.sdpa_fwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, lse)
    t = timeit(lambda: ops.sdpa_bwd(q, kv[:, :D], kv[:, D:], B, N, T, H, dh, sc, bias, kvl, out, dout, lse, delta, dq, dkv[:, :D], dkv[:, D:], work=ops.kv_work_list([L] * B, T, dev)))
    print(f"T={T} L={L}: bwd total {t:.1f}us", flush=True)
