#!/usr/bin/env python3
"""Bit-identity of two builds of the GEMM kernels: prints one SHA-1 per (layout, tile variant, K split, shape) of the output of
seeded inputs.  Run once per build (YAT_HIP_LIB selects it) and diff the two listings:

    python scripts/gemm_hash.py > a.txt;  YAT_HIP_LIB=yat_amd/build/variants/libyat_X.so python scripts/gemm_hash.py > b.txt

Shapes cover 1 / 2 / 3 / many K-tiles, a ragged last K-tile, ragged M and N edges, forced K splits and the fused epilogues of
the k-strided-B layouts (GLU backward, activation backward, row sums).  Also checks every result against torch in fp32."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

BF, dev = torch.bfloat16, "cuda"
SHAPES = [(256, 320, 64), (256, 320, 128), (256, 320, 192), (512, 640, 200), (1000, 704, 520), (2240, 5600, 8192),
          (8192, 2240, 2240), (8192, 5600, 2240), (4096, 2240, 11200), (304, 264, 72), (8192, 2240, 6720)]


def sha(t):
    return hashlib.sha1(t.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]


def main():
    g = torch.Generator(device=dev).manual_seed(7)
    bad = 0
    for lay in ("nt", "tt", "nn", "tn"):
        a_t, b_t = lay[0] == "t", lay[1] == "t"
        for (m, n, k) in SHAPES:
            a = (torch.randn((k, m) if a_t else (m, k), device=dev, generator=g) * 0.5).to(BF)
            b = (torch.randn((k, n) if b_t else (n, k), device=dev, generator=g) * 0.05).to(BF)
            ref = (a.float().t() if a_t else a.float()) @ (b.float() if b_t else b.float().t())
            for variant in (4, 5, 204, 305, 405):
                if variant >= 100 and (k < 512 or n % 8 or m % 8 or (variant // 100) * m * n * 4 > ops.GEMM_WS_BYTES):
                    continue
                out = torch.full((m, n), float("nan"), dtype=BF, device=dev)
                ops.gemm(a, b, out, a_t=a_t, b_t=b_t, M=m, N=n, K=k, variant=variant)
                torch.cuda.synchronize()
                err = ((out.float() - ref).norm() / ref.norm()).item()
                ok = err < 4e-3
                bad += not ok
                print(f"{lay} v{variant:3d} {m:5d}x{n:5d}x{k:5d} {sha(out)} rel {err:.2e}{'' if ok else '  <-- WRONG'}", flush=True)
    # fused epilogues on the k-strided-B layouts
    m, hc, d = 2048, 5600, 2240
    dy = (torch.randn(m, d, device=dev, generator=g) * 0.5).to(BF)
    w = (torch.randn(d, hc, device=dev, generator=g) * 0.05).to(BF)
    u = (torch.randn(m, 2 * hc, device=dev, generator=g)).to(BF)
    du = torch.empty(m, 2 * hc, dtype=BF, device=dev)
    ops.linear_dgrad_glu(dy, w, u, du)
    print(f"glu_bwd epilogue {sha(du)}")
    z = (torch.randn(m, hc, device=dev, generator=g)).to(BF)
    print(f"act_bwd epilogue {sha(ops.linear_dgrad_act(dy, w, z, 'silu'))}")
    x = (torch.randn(m, hc, device=dev, generator=g) * 0.5).to(BF)
    dw, db = torch.empty(d, hc, dtype=BF, device=dev), torch.empty(d, dtype=BF, device=dev)
    ops.linear_wgrad(dy, x, dw, bias_grad=db)
    print(f"wgrad + rowsum   {sha(dw)} {sha(db)}")
    ops.linear_wgrad(dy, x, dw, accumulate=True, bias_grad=db)
    print(f"wgrad accumulate {sha(dw)} {sha(db)}")
    # the forward layout's fused epilogue: bias, pre-gate copy, per-image gate, residual (and the SiLU + pre-activation form)
    B_, rows, D_, K_ = 4, 1024, 2240, 2240
    M_ = B_ * rows
    xx = (torch.randn(M_, K_, device=dev, generator=g) * 0.5).to(BF)
    ww = (torch.randn(D_, K_, device=dev, generator=g) * 0.02).to(BF)
    bias = torch.randn(D_, device=dev, generator=g).to(BF)
    mod = torch.randn(B_, 6, D_, device=dev, generator=g).to(BF)
    res = torch.randn(M_, D_, device=dev, generator=g).to(BF)
    lin, o = torch.empty(M_, D_, dtype=BF, device=dev), torch.empty(M_, D_, dtype=BF, device=dev)
    for variant in (4, 5):
        ops.gemm(xx, ww, o, M=M_, N=D_, K=K_, bias=bias, aux_out=lin, gate=mod[:, 2], ld_gate=6 * D_, residual=res,
                 rows_per_batch=rows, variant=variant)
        print(f"gate+res+aux v{variant} {sha(o)} {sha(lin)}")
        ops.gemm(xx, ww, o, M=M_, N=D_, K=K_, bias=bias, aux_out=lin, activation="silu", variant=variant)
        print(f"silu+aux     v{variant} {sha(o)} {sha(lin)}")
        dx = torch.empty(M_, K_, dtype=BF, device=dev)
        ops.gemm(o, ww, dx, b_t=True, M=M_, N=K_, K=D_, residual=res, variant=variant)          # dgrad + accumulate into a residual
        print(f"dgrad+res    v{variant} {sha(dx)}")
    torch.cuda.synchronize()
    print(f"wrong results: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
