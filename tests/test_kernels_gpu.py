"""Per-kernel parity tests (GPU): every C-ABI entry point against a plain PyTorch fp32 restatement of
the same op on the same bf16 inputs, rounded to bf16 where the reference's dtype flow rounds.

Tolerance (stated once): bf16 has 8 significant bits (ulp = 2^-8 relative; one rounding is an RMS
relative error of ~1.1e-3).  Two kinds of check:

* close(hip, ref): `ref` is computed with the SAME rounding points as the kernel (the reference's
  bf16 op flow).  Both sides may still differ by one bf16 ulp where fp32 values straddle a rounding
  boundary, so we require relative L2 <= 2e-3 and max |a-b| <= 2 bf16 ulps of the local magnitude.
* as_good_as(hip, flow, truth): for backward passes the reference IS torch's bf16 autograd
  (`flow`, run here with stock torch bf16 ops, every op rounding to bf16) and `truth` is the same
  math in fp32.  The kernel must sit as close to the truth as the reference's own arithmetic does:
  rel(hip, truth) <= 1.25 * rel(flow, truth) + 5e-4, and rel(hip, flow) <= 6e-3 (two independent
  chains of a few bf16 roundings).
Integer outputs (mask, kv_len) and the optimizer are checked bit-exactly.
All checks of a test are evaluated and printed before the test fails.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from yat_amd import ops as o
    o._lib()
    return o


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


_FAILS = []


@pytest.fixture(autouse=True)
def _collect_failures():
    _FAILS.clear()
    yield
    assert not _FAILS, "; ".join(_FAILS)


def close(a, b, name, tol=2e-3, ulps=2.0, atol=1e-6):
    a, b = a.float(), b.float()
    if not torch.isfinite(a).all():
        _FAILS.append(f"{name}: non-finite output")
        print(f"[parity] {name}: NON-FINITE")
        return
    r = rel(a, b)
    bound = ulps * 2.0 ** -8 * b.abs() + atol + 1e-3 * b.abs().mean()
    worst = ((a - b).abs() - bound).max().item()
    print(f"[parity] {name}: rel_l2={r:.3e} max_abs={(a - b).abs().max().item():.3e}")
    if r > tol:
        _FAILS.append(f"{name}: rel l2 {r:.3e} > {tol}")
    if worst > 0:
        _FAILS.append(f"{name}: element error exceeds {ulps} bf16 ulps by {worst:.3e}")


def as_good_as(hip, flow, truth, name, slack=1.25, floor=5e-4, tol_flow=6e-3):
    hip, flow, truth = hip.float(), flow.float(), truth.float()
    if not torch.isfinite(hip).all():
        _FAILS.append(f"{name}: non-finite output")
        print(f"[parity] {name}: NON-FINITE")
        return
    eh, ef, hf = rel(hip, truth), rel(flow, truth), rel(hip, flow)
    print(f"[parity] {name}: hip_vs_fp32={eh:.3e} torchbf16_vs_fp32={ef:.3e} hip_vs_torchbf16={hf:.3e}")
    if eh > slack * ef + floor:
        _FAILS.append(f"{name}: error vs fp32 truth {eh:.3e} > {slack} * reference's own {ef:.3e} + {floor}")
    if ef <= tol_flow and hf > tol_flow:      # (skipped when torch's own bf16 kernel is far from the truth)
        _FAILS.append(f"{name}: rel l2 vs torch bf16 flow {hf:.3e} > {tol_flow}")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def rb(x):
    return x.to(BF).float()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 96), (333, 264, 160), (1024, 2240, 5600), (64, 32, 256)])
def test_gemm_nt_plain(ops, M, N, K):
    x, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    y = ops.linear_fwd(x, w)
    close(y, (x.float() @ w.float().T).to(BF), f"gemm_nt {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 96), (1000, 2240, 160)])
def test_gemm_nn_dgrad(ops, M, N, K):
    dy, w = rnd(M, N, seed=3), rnd(N, K, scale=N ** -0.5, seed=4)
    dx = ops.linear_dgrad(dy, w)
    close(dx, (dy.float() @ w.float()).to(BF), f"gemm_nn {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 96), (1056, 264, 2240), (1000, 96, 32)])
def test_gemm_tn_wgrad(ops, M, N, K):
    dy, x = rnd(M, N, scale=M ** -0.5, seed=5), rnd(M, K, seed=6)
    dw = torch.empty(N, K, dtype=BF, device=DEV)
    ops.linear_wgrad(dy, x, dw)
    ref = (dy.float().T @ x.float())
    close(dw, ref.to(BF), f"gemm_tn {M}x{N}x{K}")
    ops.linear_wgrad(dy, x, dw, accumulate=True)          # grad accumulation: dw += dy^T x
    close(dw, (rb(ref) + rb(ref)).to(BF), f"gemm_tn_acc {M}x{N}x{K}")


def test_gemm_skinny_long_k(ops):
    """Skinny outputs with a long reduction take the K-split path of the 256-row kernel (csrc/gemm.hip: `skinny`): the
    embedders' M = B rows (time_embed.linear forward / dgrad at SANA width) and the 32-channel patch-embedding / output-head
    weight gradients over 8192 tokens -- all three layouts, with the epilogues those calls carry."""
    B, D = 8, 2240
    x, w, bias = rnd(B, D, seed=1), rnd(6 * D, D, scale=D ** -0.5, seed=2), rnd(6 * D, seed=3)
    y = ops.linear_fwd(x, w, bias)                                                   # nn 8 x 13440 x 2240
    close(y, (x.float() @ w.float().T + bias.float()).to(BF), "skinny fwd 8x13440x2240")     # one rounding, like torch's addmm
    z = torch.empty(B, D, dtype=BF, device=DEV)
    w2, b2 = rnd(D, D, scale=D ** -0.5, seed=4), rnd(D, seed=5)
    e = ops.linear_fwd(x, w2, b2, activation="silu", aux_out=z)                      # nn 8 x 2240 x 2240 + bias + SiLU + aux
    zr = (x.float() @ w2.float().T + b2.float()).to(BF)
    close(z, zr, "skinny fwd pre-activation")
    close(e, torch.nn.functional.silu(zr.float()).to(BF), "skinny fwd SiLU")
    dy = rnd(B, 6 * D, seed=6)
    dx = ops.linear_dgrad(dy, w)                                                     # nt 8 x 2240 x 13440
    close(dx, (dy.float() @ w.float()).to(BF), "skinny dgrad 8x2240x13440")
    M, C = 8192, 32
    g, t = rnd(M, D, scale=M ** -0.5, seed=7), rnd(M, C, seed=8)
    dw = torch.empty(D, C, dtype=BF, device=DEV)
    ops.linear_wgrad(g, t, dw)                                                       # tt 2240 x 32 x 8192
    ref = g.float().T @ t.float()
    close(dw, ref.to(BF), "skinny wgrad 2240x32x8192")
    ops.linear_wgrad(g, t, dw, accumulate=True)
    close(dw, (rb(ref) + rb(ref)).to(BF), "skinny wgrad accumulate")
    g2, t2 = rnd(M, C, scale=M ** -0.5, seed=9), rnd(M, D, seed=10)
    dw2, db2 = torch.empty(C, D, dtype=BF, device=DEV), torch.empty(C, dtype=BF, device=DEV)
    ops.linear_wgrad(g2, t2, dw2, bias_grad=db2)                                     # tt 32 x 2240 x 8192 + bias gradient
    close(dw2, (g2.float().T @ t2.float()).to(BF), "skinny wgrad 32x2240x8192")
    close(db2, g2.float().sum(0).to(BF), "skinny wgrad bias gradient", tol=4e-3, ulps=3.0)


def test_gemm_grouped_wgrad(ops):
    """yat_gemm_grouped_bf16: several dW = dy^T x of different shapes (ragged tiles, different K) in one launch are
    bit-identical to the same problems launched one by one on the 256x256 tile, plain and accumulating."""
    shapes = [(1056, 264, 2240), (1056, 520, 96), (512, 2240, 264), (200, 72, 96), (1056, 264, 264)]    # (tokens, N, K)
    items, singles = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(M, N, scale=M ** -0.5, seed=300 + i), rnd(M, K, seed=320 + i)
        items.append((dy, x, torch.zeros(N, K, dtype=BF, device=DEV)))
        one = torch.empty(N, K, dtype=BF, device=DEV)
        ops.gemm(dy, x, one, a_t=True, b_t=True, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, variant=4)
        singles.append(one)
    ops.wgrad_grouped(items)
    for (dy, x, out), one, sh in zip(items, singles, shapes):
        assert torch.equal(out, one), f"grouped wgrad differs from the single launch for {sh}"
        close(out, (dy.float().T @ x.float()).to(BF), f"gemm_grouped {sh}")
    ops.wgrad_grouped(items, accumulate=True)
    for (dy, x, out), one, sh in zip(items, singles, shapes):
        ref = rb(dy.float().T @ x.float())
        close(out, (ref + ref).to(BF), f"gemm_grouped_acc {sh}")
    with pytest.raises(Exception):
        ops.wgrad_grouped(items * 2)                      # more than 8 problems


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("M,N,K", [(256, 320, 64), (512, 640, 192), (300, 328, 96), (1024, 2240, 5600), (777, 1000, 264)])
def test_gemm256_all_layouts(ops, variant, M, N, K):
    """The 256-row staggered-wave-group kernel (gemm256.hip), every layout, ragged M/N/K tails."""
    x, w = rnd(M, K, seed=80), rnd(N, K, scale=K ** -0.5, seed=81)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(x, w, out, M=M, N=N, K=K, variant=variant)                                   # NT
    close(out, (x.float() @ w.float().T).to(BF), f"gemm256v{variant}_nt {M}x{N}x{K}")
    wt = w.T.contiguous()                                                                 # [K, N]
    ops.gemm(x, wt, out, b_t=True, M=M, N=N, K=K, variant=variant)                        # NN
    close(out, (x.float() @ wt.float()).to(BF), f"gemm256v{variant}_nn {M}x{N}x{K}")
    if M % 8 == 0:
        xt = x.T.contiguous()                                                             # [K, M]
        ops.gemm(xt, wt, out, a_t=True, b_t=True, M=M, N=N, K=K, variant=variant)         # TN
        close(out, (xt.float().T @ wt.float()).to(BF), f"gemm256v{variant}_tn {M}x{N}x{K}")
        ops.gemm(xt, w, out, a_t=True, b_t=False, M=M, N=N, K=K, variant=variant)         # TT
        close(out, (xt.float().T @ w.float().T).to(BF), f"gemm256v{variant}_tt {M}x{N}x{K}")


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 384, 72, 136, 200, 520])
def test_gemm256_deep_schedule_tile_counts(ops, variant, K):
    """The k-strided-B layouts run the DEEP LDS-DMA schedule (csrc/gemm256.hip): a four-slot ring of 32-deep halves, counted
    vmcnt waits, start-up batches that stand in for the iterations before the first.  Every K-tile count from 1 to 9 --
    start-up only, the first counted iteration, the switch to the tail form -- and a ragged last K-tile, on ragged M / N."""
    M, N = 520, 648
    x, w = rnd(M, K, seed=180), rnd(K, N, scale=K ** -0.5, seed=181)                     # dgrad layout: A k-contiguous, B k-strided
    out = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    ops.gemm(x, w, out, b_t=True, M=M, N=N, K=K, variant=variant)
    close(out, (x.float() @ w.float()).to(BF), f"deep_nn v{variant} K={K}")
    xt = x.T.contiguous()                                                                 # wgrad layout: both k-strided
    out.fill_(float("nan"))
    ops.gemm(xt, w, out, a_t=True, b_t=True, M=M, N=N, K=K, variant=variant)
    close(out, (xt.float().T @ w.float()).to(BF), f"deep_tn v{variant} K={K}")


@pytest.mark.parametrize("lay,M,N,K", [("nt", 8192, 5600, 2240), ("tt", 2240, 5600, 8192), ("nt", 4096, 2240, 11200)])
def test_gemm256_deep_schedule_is_timing_independent(ops, lay, M, N, K):
    """An LDS-DMA protocol error (a fragment read before its piece has landed, a refill before the last read) shows as
    results that depend on memory latency.  The same launch right after itself (operands in the Infinity Cache) and right
    after a 1 GiB fill (operands from HBM, several times the latency) must agree to the bit, over several rounds."""
    a_t, b_t = lay[0] == "t", lay[1] == "t"
    a = rnd(*((K, M) if a_t else (M, K)), seed=190)
    b = rnd(*((K, N) if b_t else (N, K)), scale=K ** -0.5, seed=191)
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    outs = [torch.empty(M, N, dtype=BF, device=DEV) for _ in range(2)]
    ops.gemm(a, b, outs[0], a_t=a_t, b_t=b_t, M=M, N=N, K=K)
    ops.gemm(a, b, outs[0], a_t=a_t, b_t=b_t, M=M, N=N, K=K)                             # hot
    ref = (a.float().T if a_t else a.float()) @ (b.float() if b_t else b.float().T)
    close(outs[0], ref.to(BF), f"deep_hot {lay}")
    for rnd_i in range(6):
        junk.fill_(rnd_i)
        outs[1].fill_(float("nan"))
        ops.gemm(a, b, outs[1], a_t=a_t, b_t=b_t, M=M, N=N, K=K)                         # cold
        assert torch.equal(outs[0], outs[1]), f"{lay}: cold launch {rnd_i} differs from the hot one"


@pytest.mark.parametrize("variant", [4, 5])
def test_gemm256_epilogue_and_identity(ops, variant):
    n = 512
    eye = torch.eye(n, dtype=BF, device=DEV)
    b = (torch.arange(n * n, device=DEV).reshape(n, n) % 251).to(BF)
    out = torch.empty(n, n, dtype=BF, device=DEV)
    ops.gemm(eye, b, out, M=n, N=n, K=n, variant=variant)
    assert torch.equal(out, b.T.contiguous())
    ops.gemm(eye, b, out, b_t=True, M=n, N=n, K=n, variant=variant)
    assert torch.equal(out, b)
    ops.gemm(b, eye, out, a_t=True, b_t=True, M=n, N=n, K=n, variant=variant)
    assert torch.equal(out, b.T.contiguous())
    B, rows, D, K = 3, 200, 328, 128
    M = B * rows
    x, w, bias = rnd(M, K, seed=82), rnd(D, K, scale=K ** -0.5, seed=83), rnd(D, seed=84)
    mod, res = rnd(B, 6, D, seed=85), rnd(M, D, seed=86)
    lin = torch.empty(M, D, dtype=BF, device=DEV)
    o = torch.empty(M, D, dtype=BF, device=DEV)
    ops.gemm(x, w, o, M=M, N=D, K=K, bias=bias, aux_out=lin, gate=mod[:, 2], ld_gate=6 * D, residual=res,
             rows_per_batch=rows, variant=variant)
    linr = rb(x.float() @ w.float().T + bias.float())
    close(lin, linr, f"gemm256v{variant}_epi_lin")
    close(o, (res.float() + rb(mod[:, 2].float().repeat_interleave(rows, 0) * linr)).to(BF), f"gemm256v{variant}_epi_out")
    s_out = torch.empty(M, D, dtype=BF, device=DEV)
    ops.gemm(x, w, s_out, M=M, N=D, K=K, bias=bias, aux_out=lin, activation="silu", variant=variant)
    close(s_out, F.silu(linr).to(BF), f"gemm256v{variant}_epi_silu")


@pytest.mark.parametrize("variant", [0, 4, 5])
@pytest.mark.parametrize("K,K2", [(64, 64), (128, 192), (320, 64), (2240, 320), (5600, 576), (72, 128), (200, 64)])
def test_gemm256_second_operand_pair(ops, variant, K, K2):
    """yat_gemm_epilogue.a2 / b2 / k2: C = epilogue(A B^T + A2 B2^T) with both products in one accumulator -- the K2 / 64 k-tiles
    of the second pair ride behind the last (possibly ragged) tile of K in the same LDS-DMA loop.  A2 / B2 are views with the
    row strides of A / B; the rest of their buffers holds NaNs that must never be read.  Every start-up / counted / tail form of
    the loop (1 .. 97 tiles), ragged M and N, then the fused epilogue on top."""
    M, N = 520, 648
    x, w = rnd(M, K, seed=400), rnd(N, K, scale=(K + K2) ** -0.5, seed=401)
    t, pm = rnd(M, K2, seed=402), rnd(N, K2, scale=(K + K2) ** -0.5, seed=403)
    lda, ldb = max(K, K2), max(K, K2)                     # (a 64-wide K with a 192-wide second pair: strides cover both)
    xs = torch.full((M, lda), float("nan"), dtype=BF, device=DEV); xs[:, :K] = x
    ws = torch.full((N, ldb), float("nan"), dtype=BF, device=DEV); ws[:, :K] = w
    a2 = torch.full((M, lda), float("nan"), dtype=BF, device=DEV); a2[:, :K2] = t
    b2 = torch.full((N, ldb), float("nan"), dtype=BF, device=DEV); b2[:, :K2] = pm
    ref = x.float() @ w.float().T + t.float() @ pm.float().T
    out = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    ops.gemm(xs[:, :K], ws[:, :K], out, M=M, N=N, K=K, lda=lda, ldb=ldb, a2=a2[:, :K2], b2=b2[:, :K2], k2=K2, variant=variant)
    close(out, ref.to(BF), f"gemm256v{variant}_pair K={K}+{K2}")
    # the same sum as ONE product over the concatenated operands: bit-identical (same tiles in the same order) when K is whole tiles
    if K % 64 == 0:
        xc, wc = torch.cat([x, t], 1).contiguous(), torch.cat([w, pm], 1).contiguous()
        one = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm(xc, wc, one, M=M, N=N, K=K + K2, variant=variant if variant else 0)
        if variant:
            assert torch.equal(out, one), "second operand pair differs from the concatenated product"
    bias, res = rnd(N, seed=404), rnd(M, N, seed=405)
    lin = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(xs[:, :K], ws[:, :K], out, M=M, N=N, K=K, lda=lda, ldb=ldb, a2=a2[:, :K2], b2=b2[:, :K2], k2=K2, variant=variant,
             bias=bias, aux_out=lin, activation="silu", residual=res)
    linr = rb(ref + bias.float())
    close(lin, linr, f"gemm256v{variant}_pair_lin K={K}+{K2}")
    # (only a gate rounds in between; from the kernel's own pre-activation copy, so that a rounding tie there does not count twice)
    close(out, (res.float() + F.silu(lin.float())).to(BF), f"gemm256v{variant}_pair_epi K={K}+{K2}")


def test_gemm256_second_operand_pair_blocks_and_rejections(ops):
    """a2_group_n: the column blocks of a fused q|k|v Linear each take their own K2 columns of A2 (one adapter per block); and what
    the entry point refuses: other layouts, a K2 that is not whole tiles, strides that differ, blocks the column tile straddles."""
    M, D, K, K2 = 520, 640, 256, 128
    x, w = rnd(M, K, seed=410), rnd(3 * D, K, scale=K ** -0.5, seed=411)
    t, pm = rnd(M, 3 * K2, seed=412), rnd(3 * D, K2, scale=K ** -0.5, seed=413)
    lda = 3 * K2
    xs = torch.full((M, lda), float("nan"), dtype=BF, device=DEV); xs[:, :K] = x
    a2 = t.contiguous()                                   # [M, 3 K2]: row stride 3 K2 = lda
    b2 = torch.full((3 * D, K), float("nan"), dtype=BF, device=DEV); b2[:, :K2] = pm
    ref = x.float() @ w.float().T
    for j in range(3):
        ref[:, j * D:(j + 1) * D] += t[:, j * K2:(j + 1) * K2].float() @ pm[j * D:(j + 1) * D].float().T
    out = torch.full((M, 3 * D), float("nan"), dtype=BF, device=DEV)
    for variant in (0, 5):                                # 640 = 2 x 320: the 320-wide tile only
        out.fill_(float("nan"))
        ops.gemm(xs[:, :K], w, out, M=M, N=3 * D, K=K, lda=lda, a2=a2, b2=b2[:, :K2], k2=K2, a2_group_n=D, variant=variant)
        close(out, ref.to(BF), f"gemm256_pair_blocks v{variant}")
    from yat_amd import lib as L
    bad = [dict(variant=4, a2_group_n=D),                 # 640 is not a multiple of 256
           dict(k2=96), dict(variant=1), dict(variant=205), dict(a2_group_n=100)]
    for kw in bad:
        args = dict(M=M, N=3 * D, K=K, lda=lda, a2=a2, b2=b2[:, :K2], k2=K2, a2_group_n=D, variant=0)
        args.update(kw)
        with pytest.raises(L.YatLibraryError):
            ops.gemm(xs[:, :K], w, out, **args)
    with pytest.raises(L.YatLibraryError):                # dgrad layout
        ops.gemm(xs[:, :K], w.T.contiguous(), out, b_t=True, M=M, N=3 * D, K=K, lda=lda, ldb=3 * D, a2=a2,
                 b2=w.T.contiguous()[:K2], k2=K2)
    with pytest.raises(ValueError):                       # A2 with another row stride
        ops.gemm(xs[:, :K], w, out, M=M, N=3 * D, K=K, lda=lda, a2=t[:, :K2].contiguous(), b2=b2[:, :K2], k2=K2)


@pytest.mark.parametrize("code", [204, 405, 205, 804, 304, 1005, 305])
def test_gemm256_split_k(ops, code):
    """Split-K (fp32 slabs + reduce kernel with the fused epilogue), forced via variant = 100*ksplit + tile."""
    M, N, K = 520, 648, 2048 + 96                     # ragged everything; K tail lands in the last slice
    dy, x = rnd(K, M, scale=K ** -0.5, seed=90), rnd(K, N, seed=91)            # wgrad layout: both k-strided
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(dy, x, out, a_t=True, b_t=True, M=M, N=N, K=K, variant=code)
    ref = dy.float().T @ x.float()
    close(out, ref.to(BF), f"splitk{code}_tn")
    ops.gemm(dy, x, out, a_t=True, b_t=True, M=M, N=N, K=K, variant=code, residual=out)      # accumulate
    close(out, (rb(ref) + rb(ref)).to(BF), f"splitk{code}_tn_acc")
    a, w, bias = rnd(M, K, seed=92), rnd(N, K, scale=K ** -0.5, seed=93), rnd(N, seed=94)
    z = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(a, w, out, M=M, N=N, K=K, variant=code, bias=bias, activation="silu", aux_out=z)
    zr = rb(a.float() @ w.float().T + bias.float())
    close(z, zr, f"splitk{code}_nt_preact")
    close(out, F.silu(z.float()).to(BF), f"splitk{code}_nt_silu")      # activation of the kernel's own rounded pre-activation


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C-write or a swapped fragment map."""
    n = 128
    eye = torch.eye(n, dtype=BF, device=DEV)
    b = (torch.arange(n * n, device=DEV).reshape(n, n) % 251).to(BF)      # exact small integers
    y = ops.linear_fwd(eye, b)                      # I @ b^T
    assert torch.equal(y, b.T.contiguous())
    y = ops.linear_dgrad(eye, b)                    # I @ b
    assert torch.equal(y, b)
    dw = torch.empty(n, n, dtype=BF, device=DEV)
    ops.linear_wgrad(eye, b, dw)                    # I^T @ b
    assert torch.equal(dw, b)
    ops.linear_wgrad(b, eye, dw)                    # b^T @ I
    assert torch.equal(dw, b.T.contiguous())


@pytest.mark.parametrize("act", ["silu", "gelu_tanh"])
def test_gemm_epilogue_bias_act(ops, act):
    M, N, K = 264, 200, 96
    x, w, b = rnd(M, K, seed=7), rnd(N, K, scale=K ** -0.5, seed=8), rnd(N, seed=9)
    z = torch.empty(M, N, dtype=BF, device=DEV)
    y = ops.linear_fwd(x, w, bias=b, activation=act, aux_out=z)
    zr = rb(x.float() @ w.float().T + b.float())
    close(z, zr, f"epi_{act}_preact")
    yr = F.silu(zr) if act == "silu" else F.gelu(zr, approximate="tanh")
    close(y, yr.to(BF), f"epi_{act}_out")


def test_gemm_epilogue_gate_residual(ops):
    B, n, D, K = 3, 88, 136, 64
    M = B * n
    x, w, b = rnd(M, K, seed=10), rnd(D, K, scale=K ** -0.5, seed=11), rnd(D, seed=12)
    mod = rnd(B, 6, D, seed=13)
    res = rnd(M, D, seed=14)
    lin = torch.empty(M, D, dtype=BF, device=DEV)
    gate = mod[:, 2]
    out = ops.linear_fwd(x, w, bias=b, aux_out=lin, gate=gate, ld_gate=6 * D, residual=res, rows_per_batch=n)
    linr = rb(x.float() @ w.float().T + b.float())
    close(lin, linr, "epi_gate_lin")
    g = gate.float().repeat_interleave(n, 0)
    close(out, (res.float() + rb(g * linr)).to(BF), "epi_gate_out")
    out2 = ops.linear_fwd(x, w, bias=b, residual=res)       # cross-attention: ungated residual
    close(out2, (res.float() + linr).to(BF), "epi_residual_out")


def test_gemm_tt_and_transpose(ops):
    Bn, R, Cc = 3, 50, 24
    x = rnd(Bn, R, Cc, seed=70)
    assert torch.equal(ops.transpose(x), x.transpose(1, 2).contiguous())
    M, N, K = 136, 72, 64                      # A stored [K, M], B stored [N, K]
    a, b = rnd(K, M, seed=71), rnd(N, K, scale=K ** -0.5, seed=72)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(a, b, out, a_t=True, b_t=False, M=M, N=N, K=K)
    close(out, (a.float().T @ b.float().T).to(BF), "gemm_tt")


def test_colsum(ops):
    rows, cols = 1000, 264
    x = rnd(rows, cols, seed=15)
    ws = torch.empty(int(ops._lib().yat_colsum_workspace_bytes(rows, cols)), dtype=torch.uint8, device=DEV)
    out = torch.zeros(cols, dtype=BF, device=DEV)
    ops.colsum(x, out, ws)
    close(out, x.float().sum(0).to(BF), "colsum")
    ops.colsum(x, out, ws, accumulate=True)
    close(out, (2 * rb(x.float().sum(0))).to(BF), "colsum_acc")


# ------------------------------------------------------------------------------------------------ norms
def _ln_ref(x, shift, scale, eps, n):
    xf = x.float()
    xh = rb(F.layer_norm(xf, (xf.shape[-1],), None, None, eps))
    sc = rb(1 + scale.float()).repeat_interleave(n, 0)
    sh = shift.float().repeat_interleave(n, 0)
    return rb(rb(xh * sc) + sh)


@pytest.mark.parametrize("B,n,D", [(2, 16, 64), (3, 41, 2240), (2, 33, 1152), (1, 7, 4096)])
def test_ln_modulate_fwd_bwd(ops, B, n, D):
    M = B * n
    x = rnd(M, D, scale=2.0, seed=20) + 0.5
    mod = rnd(B, 6, D, scale=0.3, seed=21)
    shift, scale = mod[:, 3], mod[:, 4]
    y, mean, rstd = ops.ln_modulate_fwd(x, shift, scale, 6 * D, n, 1e-6)
    close(y, _ln_ref(x, shift, scale, 1e-6, n), f"ln_fwd D={D}")
    # backward: torch autograd of the reference's op sequence, once in bf16 (the reference) and once in fp32
    dy = rnd(M, D, seed=22)
    dres = rnd(M, D, seed=23)

    def run(dt):
        xr = x.detach().to(dt).clone().requires_grad_(True)
        scr = scale.detach().to(dt).clone().requires_grad_(True)
        shr = shift.detach().to(dt).clone().requires_grad_(True)
        yr = F.layer_norm(xr, (D,), None, None, 1e-6) * (1 + scr).repeat_interleave(n, 0) + shr.repeat_interleave(n, 0)
        yr.backward(dy.to(dt))
        return xr.grad + dres.to(dt), shr.grad, scr.grad
    flow, truth = run(BF), run(torch.float32)
    dx = torch.empty_like(x)
    acc = torch.zeros(B, 6, D, dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.ln_bwd_workspace_bytes(M, D, n), dtype=torch.uint8, device=DEV)
    ops.ln_modulate_bwd(x, mean, rstd, scale, 6 * D, n, dy, dres, dx, acc[:, 3], acc[:, 4], 6 * D, ws)
    as_good_as(dx, flow[0], truth[0], f"ln_bwd_dx D={D}")
    as_good_as(acc[:, 3], flow[1], truth[1], f"ln_bwd_dshift D={D}", tol_flow=2e-2)
    as_good_as(acc[:, 4], flow[2], truth[2], f"ln_bwd_dscale D={D}", tol_flow=2e-2)
    assert acc[:, [0, 1, 2, 5]].abs().max().item() == 0.0
    # the two halves launched separately (dx on the dependent chain, shift/scale gradients elsewhere) give the same bits
    dx_p, acc_p = torch.empty_like(dx), torch.zeros_like(acc)
    ops.ln_modulate_bwd(x, mean, rstd, scale, 6 * D, n, dy, dres, dx_p, acc_p[:, 3], acc_p[:, 4], 6 * D, ws, parts=1)
    assert torch.equal(dx_p, dx) and acc_p.abs().max().item() == 0.0
    ops.ln_modulate_bwd(x, mean, rstd, scale, 6 * D, n, dy, None, None, acc_p[:, 3], acc_p[:, 4], 6 * D, ws, parts=2)
    assert torch.equal(acc_p, acc)


def test_rmsnorm_fwd_bwd(ops):
    M, D = 300, 2240
    x, w = rnd(M, D, scale=1.5, seed=24), (1 + 0.1 * torch.randn(D)).to(BF).to(DEV)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    xf = x.float()
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    close(y, (rb(xf * r) * w.float()).to(BF), "rmsnorm_fwd")
    dy = rnd(M, D, seed=25)

    def run(dt):       # diffusers RMSNorm op sequence (oracle/sana_ref.py RMSNorm)
        xr, wr = x.detach().to(dt).clone().requires_grad_(True), w.detach().to(dt).clone().requires_grad_(True)
        var = xr.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hcur = (xr * torch.rsqrt(var + 1e-5)).to(dt) * wr
        hcur.backward(dy.to(dt))
        return xr.grad, wr.grad
    flow, truth = run(BF), run(torch.float32)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    ws = torch.empty(int(ops._lib().yat_rmsnorm_bwd_workspace_bytes(M, D)), dtype=torch.uint8, device=DEV)
    ops.rmsnorm_bwd(x, w, rstd, dy, dx, dw, ws)
    as_good_as(dx, flow[0], truth[0], "rmsnorm_bwd_dx")
    as_good_as(dw, flow[1], truth[1], "rmsnorm_bwd_dw", tol_flow=1e-2)


def test_modulation_fwd_bwd(ops):
    B, S, D = 3, 6, 136
    table, tmod = rnd(S, D, seed=26), rnd(B, S * D, seed=27)
    mod = ops.modulation_fwd(table, tmod, D)
    close(mod, (table.float()[None] + tmod.float().view(B, S, D)).to(BF), "modulation_fwd")
    emb = rnd(B, D, seed=28)
    mod2 = ops.modulation_fwd(table[:2].contiguous(), emb, 0)
    close(mod2, (table[:2].float()[None] + emb.float()[:, None]).to(BF), "modulation_fwd_shared")
    dmod = torch.randn(B, S, D, device=DEV)
    dtab = torch.empty(S, D, dtype=BF, device=DEV)
    dt = torch.zeros(B, S * D, device=DEV)
    ops.modulation_bwd(dmod, dtab, dt, D)
    close(dtab, rb(dmod).sum(0).to(BF), "modulation_bwd_table")
    close(dt, rb(dmod).view(B, S * D), "modulation_bwd_tmod")
    dt2 = torch.zeros(B, D, device=DEV)
    ops.modulation_bwd(dmod[:, :2].contiguous(), dtab[:2], dt2, 0)
    close(dt2, rb(dmod[:, :2]).sum(1), "modulation_bwd_shared")


def test_gate_bwd(ops):
    B, n, D = 3, 150, 264
    M = B * n
    dout, lin, mod = rnd(M, D, seed=29), rnd(M, D, seed=30), rnd(B, 6, D, seed=31)
    gate = mod[:, 5]
    dlin = torch.empty(M, D, dtype=BF, device=DEV)
    acc = torch.zeros(B, 6, D, device=DEV)
    ws = torch.empty(int(ops._lib().yat_gate_bwd_workspace_bytes(M, D, n)), dtype=torch.uint8, device=DEV)
    ops.gate_bwd(dout, lin, gate, 6 * D, n, dlin, acc[:, 5], 6 * D, ws)
    close(dlin, (gate.float().repeat_interleave(n, 0) * dout.float()).to(BF), "gate_bwd_dlin")
    close(acc[:, 5], rb(dout.float() * lin.float()).view(B, n, D).sum(1), "gate_bwd_dgate", atol=1e-2)
    # fused bias gradient of the gated Linear = column sum of dlin (same pass), plain and accumulating
    dlin2, acc2 = torch.empty_like(dlin), torch.zeros_like(acc)
    dbias = torch.empty(D, dtype=BF, device=DEV)
    ops.gate_bwd(dout, lin, gate, 6 * D, n, dlin2, acc2[:, 5], 6 * D, ws, dbias=dbias)
    assert torch.equal(dlin2, dlin) and torch.equal(acc2, acc)
    want = dlin.float().sum(0)
    close(dbias, want.to(BF), "gate_bwd_dbias", atol=1e-2)
    ref_cs = torch.empty(D, dtype=BF, device=DEV)
    ops.colsum(dlin, ref_cs, torch.empty(int(ops._lib().yat_colsum_workspace_bytes(M, D)), dtype=torch.uint8, device=DEV))
    close(dbias, ref_cs, "gate_bwd_dbias_vs_colsum", atol=1e-2)
    ops.gate_bwd(dout, lin, gate, 6 * D, n, dlin2, acc2[:, 5], 6 * D, ws, dbias=dbias, accumulate_bias=True)
    close(dbias, (2 * want).to(BF), "gate_bwd_dbias_acc", tol=1e-2, atol=2e-2)


# ------------------------------------------------------------------------------------------------ attention
def _linattn_ref(qkv, B, N, H):
    D = H * 32
    q, k, v = qkv.float().view(B, N, 3, H, 32).permute(2, 0, 3, 1, 4)      # [B,H,N,32]
    q, k = F.relu(q), F.relu(k)
    v1 = F.pad(v, (0, 1), value=1.0)                                        # [B,H,N,33]
    S = v1.transpose(-1, -2) @ k                                            # [B,H,33,32]
    U = q @ S.transpose(-1, -2)                                             # [B,H,N,33]
    o = U[..., :32] / (U[..., 32:] + 1e-15)
    return o.permute(0, 2, 1, 3).reshape(B * N, D)


@pytest.mark.parametrize("B,N,H", [(2, 16, 2), (2, 300, 3), (1, 1024, 5)])
def test_linear_attn_fwd_bwd(ops, B, N, H):
    D = H * 32
    qkv = rnd(B * N, 3 * D, seed=32)
    ws = torch.empty(ops.linear_attn_workspace_bytes(B, N, H), dtype=torch.uint8, device=DEV)
    out = torch.empty(B * N, D, dtype=BF, device=DEV)
    ops.linear_attn_fwd(qkv, B, N, H, D, 2 * D, out, ws)
    qr = qkv.float().requires_grad_(True)
    ref = _linattn_ref(qr, B, N, H)
    close(out, ref.to(BF), f"linattn_fwd N={N}")
    dout = rnd(B * N, D, seed=33)
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    ops.linear_attn_bwd(qkv, B, N, H, D, 2 * D, dout, dqkv, ws)
    close(dqkv, qr.grad.to(BF), f"linattn_bwd N={N}", tol=4e-3, ulps=3, atol=1e-4)
    # with the forward's state handed back (what the model does) the result is bit-identical
    state = ws.view(torch.float32)[: B * H * 33 * 32].clone()
    dqkv2 = torch.empty_like(qkv)
    ops.linear_attn_bwd(qkv, B, N, H, D, 2 * D, dout, dqkv2, ws, state=state)
    assert torch.equal(dqkv, dqkv2)


def _sdpa_ref(q, k, v, bias, B, N, T, H, dh, scale):
    qf = q.float().view(B, N, H, dh).transpose(1, 2)
    kf = k.float().view(B, T, H, dh).transpose(1, 2)
    vf = v.float().view(B, T, H, dh).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * scale + bias[:, None, None, :]
    p = torch.softmax(s, -1)
    return (p @ vf).transpose(1, 2).reshape(B * N, H * dh), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,T,H,dh,lens", [
    (2, 64, 64, 1, 32, [64, 20]), (2, 100, 128, 2, 112, [77, 128]), (3, 130, 512, 2, 112, [300, 5, 0]), (1, 48, 40, 3, 64, [33]),
    (2, 70, 90, 2, 72, [90, 41]), (1, 40, 200, 2, 24, [130]), (2, 33, 64, 1, 128, [64, 9]),
    # ceil(N/128) * H * B >= 1024: the 128-query workgroup variants of forward and dQ, one per head-dim instantiation
    (8, 250, 200, 64, 32, [200, 7, 64, 65, 128, 199, 1, 100]), (16, 200, 100, 32, 64, [100, 3] * 8),
    (16, 256, 150, 32, 72, [150, 149, 64, 1] * 4), (8, 130, 130, 64, 112, [130, 64, 65, 2] * 2),
    (32, 130, 70, 16, 128, [70, 1, 64, 33] * 8),
    # ceil(N/192) * H * B >= 1024 and dh <= 80: the 192-query forward
    (16, 400, 150, 32, 72, [150, 149, 64, 1] * 4), (8, 200, 100, 128, 32, [100, 3] * 4), (16, 385, 90, 32, 64, [90, 17] * 8),
    # ceil(T/128) * H * B >= 1024, dh <= 80, dense grid: dK/dV with 32 keys per wave (ragged T: partial last workgroup)
    (8, 150, 300, 64, 72, [300, 299, 130, 0, 64, 1, 200, 257]), (16, 100, 200, 32, 32, [200, 5] * 8)])
def test_sdpa_fwd_bwd(ops, B, N, T, H, dh, lens):
    D = H * dh
    scale = 1.0 / math.sqrt(dh)
    q = rnd(B * N, D, seed=34)
    kv = rnd(B * T, 2 * D, seed=35)
    k, v = kv[:, :D], kv[:, D:]
    mask = torch.zeros(B, T)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
    bias = rb((1 - mask.to(BF).float()) * -10000.0).to(DEV)
    kvlen = torch.tensor(lens, dtype=torch.int32, device=DEV)
    out = torch.empty(B * N, D, dtype=BF, device=DEV)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=DEV)
    ops.sdpa_fwd(q, k, v, B, N, T, H, dh, scale, bias, kvlen, out, lse)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k.contiguous(), v.contiguous()))
    ref, lse_ref = _sdpa_ref(qr, kr, vr, bias, B, N, T, H, dh, scale)
    # the reference's op: F.scaled_dot_product_attention on bf16 tensors with the additive bf16 mask
    qb, kb, vb = (t.clone().view(B, -1, H, dh).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    mb = bias.to(BF)[:, None, None, :].expand(B, H, 1, T)
    flow = F.scaled_dot_product_attention(qb, kb, vb, attn_mask=mb)
    flow2d = flow.transpose(1, 2).reshape(B * N, D)
    as_good_as(out, flow2d, ref, f"sdpa_fwd T={T} dh={dh}")
    close(lse, lse_ref, f"sdpa_lse T={T}", tol=1e-4, ulps=0.01, atol=1e-3)
    dout = rnd(B * N, D, seed=36)
    ref.backward(dout.float())
    flow2d.backward(dout)
    fl = [t.grad.transpose(1, 2).reshape(-1, D) for t in (qb, kb, vb)]
    dq = torch.empty_like(q)
    dkv = torch.full_like(kv, float("nan"))
    delta = torch.empty(B, H, N, dtype=torch.float32, device=DEV)
    ops.sdpa_bwd(q, k, v, B, N, T, H, dh, scale, bias, kvlen, out, dout, lse, delta, dq, dkv[:, :D], dkv[:, D:])
    # the compact (batch, key tile) work list must give bit-identical gradients to the dense grid
    dq2, dkv2 = torch.empty_like(q), torch.full_like(kv, float("nan"))
    ops.sdpa_bwd(q, k, v, B, N, T, H, dh, scale, bias, kvlen, out, dout, lse, delta, dq2, dkv2[:, :D], dkv2[:, D:],
                 work=ops.kv_work_list(lens, T, DEV))
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2)
    # dQ (+delta) and dK/dV as two separate calls (the second may run on another stream): same bits
    dq3, dkv3, delta3 = torch.empty_like(dq), torch.full_like(kv, float("nan")), torch.empty_like(delta)
    for part in (1, 2):
        ops.sdpa_bwd(q, k, v, B, N, T, H, dh, scale, bias, kvlen, out, dout, lse, delta3, dq3, dkv3[:, :D], dkv3[:, D:],
                     work=ops.kv_work_list(lens, T, DEV), parts=part)
    assert torch.equal(dq, dq3) and torch.equal(dkv, dkv3) and torch.equal(delta, delta3)
    as_good_as(dq, fl[0], qr.grad, f"sdpa_dq T={T}", tol_flow=1e-2)
    as_good_as(dkv[:, :D], fl[1], kr.grad, f"sdpa_dk T={T}", tol_flow=1e-2)
    as_good_as(dkv[:, D:], fl[2], vr.grad, f"sdpa_dv T={T}", tol_flow=1e-2)


@pytest.mark.parametrize("B,N,H,dh", [
    (2, 64, 1, 32), (2, 100, 2, 112), (1, 77, 3, 64), (2, 130, 2, 72), (1, 50, 2, 24), (2, 33, 1, 128), (1, 200, 2, 80),
    # head dims between the models' own: 104 passes the argument checks and must NOT take the dh-112 row-sum form (its ones
    # column would sit in another output tile: round-3 advisor finding), 88 / 96 fall between the ONES classes
    (2, 90, 2, 104), (1, 70, 2, 88), (2, 65, 1, 96), (1, 130, 3, 40), (1, 64, 2, 56),
    # the 128- and 192-query workgroups (ceil(N/128 | 192) * H * B >= 1024) and the 32-key-per-wave dK/dV, ragged last tiles
    (8, 250, 64, 32), (16, 200, 32, 64), (16, 400, 32, 72), (8, 130, 64, 112), (16, 385, 32, 64), (8, 300, 64, 72)])
def test_sdpa_without_bias_matches_zero_bias(ops, B, N, H, dh):
    """key_bias = NULL (self-attention: PixArt-Sigma attn1, the MMDiT's joint and image-only attentions, SANA's softmax
    attn1): the no-bias instantiations -- scale folded into the exponential's multiply-add, row sums out of the P V product
    where the head dim leaves (or is given) a padding column, lazy rescale -- against the zero-bias entry and against torch:
    as good as torch's own bf16 kernel measured from the fp32 truth, forward (+ lse) and all three gradients, for every
    head-dim class and for sequence lengths that end inside a 64-key tile."""
    D, T = H * dh, N
    scale = 1.0 / math.sqrt(dh)
    qkv = rnd(B * N, 3 * D, seed=44)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    dout = rnd(B * N, D, seed=45)
    zero, full = torch.zeros(B, T, device=DEV), torch.full((B,), T, dtype=torch.int32, device=DEV)
    res = {}
    for tag, bias, kvl in (("zero", zero, full), ("none", None, None)):
        out, lse = torch.empty(B * N, D, dtype=BF, device=DEV), torch.empty(B, H, N, device=DEV)
        ops.sdpa_fwd(q, k, v, B, N, T, H, dh, scale, bias, kvl, out, lse)
        dqkv = torch.full_like(qkv, float("nan"))
        ops.sdpa_bwd(q, k, v, B, N, T, H, dh, scale, bias, kvl, out, dout, lse, torch.empty(B, H, N, device=DEV),
                     dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
        assert torch.isfinite(dqkv.float()).all()
        res[tag] = (out, lse, dqkv)

    def torch_path(dt):
        t = qkv.to(dt).clone().requires_grad_(True)
        heads = lambda x: x.reshape(B, N, H, dh).transpose(1, 2)
        o = F.scaled_dot_product_attention(heads(t[:, :D]), heads(t[:, D:2 * D]), heads(t[:, 2 * D:]))
        o = o.transpose(1, 2).reshape(B * N, D)
        o.backward(dout.to(dt))
        return o.detach(), t.grad
    o32, g32 = torch_path(torch.float32)
    obf, gbf = torch_path(BF)
    s32 = (q.float().view(B, N, H, dh).transpose(1, 2) @ k.float().view(B, N, H, dh).permute(0, 2, 3, 1)) * scale
    lse_ref = torch.logsumexp(s32, -1)
    out, lse, dqkv = res["none"]
    as_good_as(out, obf, o32, f"sdpa_nobias_fwd N={N} dh={dh}")
    close(lse, lse_ref, f"sdpa_nobias_lse N={N} dh={dh}", tol=1e-4, ulps=0.01, atol=1e-3)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        as_good_as(dqkv[:, sl], gbf[:, sl], g32[:, sl], f"sdpa_nobias_{name} N={N} dh={dh}", tol_flow=1e-2)
    # and next to the zero-bias entry: the same function, another rounding of P's scale -- a fraction of an ulp apart
    assert rel(out, res["zero"][0]) <= 3e-3 and rel(dqkv, res["zero"][2]) <= 6e-3
    # (lse: the no-bias forward sums the bf16-rounded probabilities -- its row sums are those of the P the P V product uses)
    assert (lse - res["zero"][1]).abs().max().item() <= 3e-3


@pytest.mark.parametrize("B,N,T,H,dh,masked", [(2, 130, 130, 2, 64, False), (2, 200, 200, 2, 72, False), (8, 300, 300, 64, 72, False),
                                               (2, 100, 140, 2, 112, True), (2, 96, 70, 2, 72, True)])
def test_sdpa_backward_with_sharp_logits(ops, B, N, T, H, dh, masked):
    """Logits of +-40 and row log-sum-exps of 20..45: the backward kernels start their S accumulators at -lse / scale (plus
    the key bias / scale) and their dP accumulators at -delta, so a large lse sits in the accumulator beside the dot product it is
    subtracted from -- the gradients must still be as good as torch's bf16 kernel measured from the fp32 truth, with and without
    a masking key bias, on the dense 32-key-per-wave dK/dV (H * B * ceil(T/128) >= 1024) and the work-list path."""
    D = H * dh
    scale = 1.0 / math.sqrt(dh)
    q = rnd(B * N, D, scale=3.5, seed=61)
    kv = rnd(B * T, 2 * D, scale=3.5, seed=62)
    kv[:, D:] = rnd(B * T, D, seed=63)
    k, v = kv[:, :D], kv[:, D:]
    dout = rnd(B * N, D, seed=64)
    lens = [T - 37 * (b % 3) for b in range(B)] if masked else [T] * B
    mask = torch.zeros(B, T)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
    bias = rb((1 - mask.to(BF).float()) * -10000.0).to(DEV) if masked else None
    kvl = torch.tensor(lens, dtype=torch.int32, device=DEV) if masked else None
    out, lse = torch.empty(B * N, D, dtype=BF, device=DEV), torch.empty(B, H, N, device=DEV)
    ops.sdpa_fwd(q, k, v, B, N, T, H, dh, scale, bias, kvl, out, lse)
    assert lse.max().item() > 20.0                                   # the case this test is about
    dq, dkv = torch.empty_like(q), torch.full_like(kv, float("nan"))
    ops.sdpa_bwd(q, k, v, B, N, T, H, dh, scale, bias, kvl, out, dout, lse, torch.empty(B, H, N, device=DEV), dq, dkv[:, :D],
                 dkv[:, D:], work=ops.kv_work_list(lens, T, DEV) if masked else None)

    def torch_path(dt):
        heads = lambda x, L: x.to(dt).reshape(B, L, H, dh).transpose(1, 2).clone().requires_grad_(True)
        qh, kh, vh = heads(q, N), heads(k, T), heads(v, T)
        mb = None if bias is None else bias.to(dt)[:, None, None, :].expand(B, H, 1, T)
        o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mb).transpose(1, 2).reshape(B * N, D)
        o.backward(dout.to(dt))
        return [o.detach()] + [t.grad.transpose(1, 2).reshape(-1, D) for t in (qh, kh, vh)]
    t32, tbf = torch_path(torch.float32), torch_path(BF)
    as_good_as(out, tbf[0], t32[0], f"sdpa_sharp_fwd dh={dh}")
    for name, mine, i in (("dq", dq, 1), ("dk", dkv[:, :D], 2), ("dv", dkv[:, D:], 3)):
        as_good_as(mine, tbf[i], t32[i], f"sdpa_sharp_{name} dh={dh} masked={masked}", tol_flow=2e-2)


# ------------------------------------------------------------------------------------------------ GLUMBConv middle
def _glu_ref(z, wdw, bdw, B, h, w, Hc):
    zi = z.float().view(B, h, w, 2 * Hc).permute(0, 3, 1, 2)
    s = rb(F.silu(zi))
    u = rb(F.conv2d(s, wdw.float().view(2 * Hc, 1, 3, 3), bdw.float(), padding=1, groups=2 * Hc))
    a, g = u.chunk(2, 1)
    y = a * rb(F.silu(g))
    return y.permute(0, 2, 3, 1).reshape(B * h * w, Hc)


# the last two cases take the direct (non-tiled) kernels: w > 64, and a channel count that is not a multiple of 8
# (the streaming forward takes the shapes whose rows fill its 16 run slots: 5x9, 32x32, 7x20, 9x16, 10x8, 31x30)
@pytest.mark.parametrize("B,h,w,Hc", [(2, 4, 4, 16), (2, 5, 9, 40), (1, 16, 64, 160), (2, 33, 3, 8), (2, 32, 32, 72),
                                      (1, 3, 70, 8), (1, 6, 5, 12), (1, 7, 20, 40), (2, 9, 16, 64), (1, 10, 8, 24),
                                      (3, 31, 30, 136)])
def test_dwconv_glu_fwd_bwd(ops, B, h, w, Hc):
    M = B * h * w
    z = rnd(M, 2 * Hc, seed=37)
    wdw, bdw = rnd(2 * Hc, 9, scale=1 / 3, seed=38), rnd(2 * Hc, scale=0.1, seed=39)
    y = torch.empty(M, Hc, dtype=BF, device=DEV)
    s_act = F.silu(z.float()).to(BF)          # what the conv_inverted GEMM epilogue stores next to z
    ops.dwconv_glu_fwd(s_act, B, h, w, Hc, wdw, bdw, y)
    close(y, _glu_ref(z, wdw, bdw, B, h, w, Hc).to(BF), f"dwconv_fwd {h}x{w}x{Hc}")
    # backward: torch autograd of diffusers GLUMBConv's op sequence in bf16 (the reference) and in fp32
    dy = rnd(M, Hc, seed=40)

    def run(dt):
        zr, wr, br = (t.detach().to(dt).clone().requires_grad_(True) for t in (z, wdw, bdw))
        zi = zr.view(B, h, w, 2 * Hc).permute(0, 3, 1, 2)
        u = F.conv2d(F.silu(zi), wr.view(2 * Hc, 1, 3, 3), br, padding=1, groups=2 * Hc)
        a, g = torch.chunk(u, 2, dim=1)
        yr = (a * F.silu(g)).permute(0, 2, 3, 1).reshape(M, Hc)
        yr.backward(dy.to(dt))
        return zr.grad, wr.grad, br.grad
    flow, truth = run(BF), run(torch.float32)
    dz, dw, db = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc), dtype=torch.uint8, device=DEV)
    ops.dwconv_glu_bwd(s_act, z, B, h, w, Hc, wdw, bdw, dy, dz, dw, db, ws)
    as_good_as(dz, flow[0], truth[0], f"dwconv_dz {h}x{w}x{Hc}", tol_flow=1e-2)
    as_good_as(dw, flow[1], truth[1], f"dwconv_dw {h}x{w}x{Hc}", tol_flow=1e-2)
    as_good_as(db, flow[2], truth[2], f"dwconv_db {h}x{w}x{Hc}", tol_flow=1e-2)
    # fused conv_inverted bias gradient: column sum of the dz the kernel itself wrote
    dz2, dw2, db2 = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
    dzs = torch.empty(2 * Hc, dtype=BF, device=DEV)
    ops.dwconv_glu_bwd(s_act, z, B, h, w, Hc, wdw, bdw, dy, dz2, dw2, db2, ws, dz_colsum=dzs)
    assert torch.equal(dz2, dz) and torch.equal(dw2, dw) and torch.equal(db2, db)
    close(dzs, dz.float().sum(0).to(BF), f"dwconv_dz_colsum {h}x{w}x{Hc}", atol=1e-2)


@pytest.mark.parametrize("B,h,w,Hc,Dm", [(2, 8, 8, 40, 64), (1, 16, 64, 160, 96), (2, 32, 32, 320, 264), (1, 3, 70, 8, 32)])
def test_glu_backward_in_gemm_epilogue(ops, B, h, w, Hc, Dm):
    """The forward keeps u = dwconv(s)+b; the GEMM that produces dy (conv_point dgrad) applies the GLU backward in its
    epilogue.  Must be bit-identical to the unfused path (dy GEMM -> depthwise backward pass 1 -> pass 2)."""
    M = B * h * w
    z = rnd(M, 2 * Hc, seed=50)
    s_act = F.silu(z.float()).to(BF)
    wdw, bdw = rnd(2 * Hc, 9, scale=1 / 3, seed=51), rnd(2 * Hc, scale=0.1, seed=52)
    y, y2, u = (torch.empty(M, c, dtype=BF, device=DEV) for c in (Hc, Hc, 2 * Hc))
    ops.dwconv_glu_fwd(s_act, B, h, w, Hc, wdw, bdw, y)
    ops.dwconv_glu_fwd(s_act, B, h, w, Hc, wdw, bdw, y2, u_out=u)
    assert torch.equal(y, y2)
    si = s_act.float().view(B, h, w, 2 * Hc).permute(0, 3, 1, 2)
    uref = F.conv2d(si, wdw.float().view(2 * Hc, 1, 3, 3), bdw.float(), padding=1, groups=2 * Hc)
    close(u, uref.permute(0, 2, 3, 1).reshape(M, 2 * Hc).to(BF), f"dwconv_u {h}x{w}x{Hc}")
    ua, ug = u[:, :Hc].float(), u[:, Hc:].float()
    assert torch.equal(y.float(), (ua * F.silu(ug).to(BF).float()).to(BF).float())      # y is exactly GLU(u)
    dlin, wp = rnd(M, Dm, scale=0.2, seed=53), rnd(Dm, Hc, scale=Dm ** -0.5, seed=54)   # conv_point: [D, Hc]
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc), dtype=torch.uint8, device=DEV)
    outs = []
    for fused in (False, True):
        dz, dw, db = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
        if fused:
            du = torch.empty(M, 2 * Hc, dtype=BF, device=DEV)
            ops.linear_dgrad_glu(dlin, wp, u, du)
            ops.dwconv_glu_bwd(s_act, z, B, h, w, Hc, wdw, bdw, None, dz, dw, db, ws, du=du)
        else:
            dy = ops.linear_dgrad(dlin, wp)
            ops.dwconv_glu_bwd(s_act, z, B, h, w, Hc, wdw, bdw, dy, dz, dw, db, ws)
        outs.append((dz, dw, db))
    for a, b_, nm in zip(outs[0], outs[1], ("dz", "dw", "db")):
        assert torch.equal(a, b_), f"fused GLU backward changes {nm}"


@pytest.mark.parametrize("B,h,w,Hc", [(2, 32, 32, 72), (1, 16, 64, 160), (2, 9, 16, 64), (1, 3, 70, 8), (1, 6, 5, 12)])
def test_dwconv_bwd_pass2_never_reads_s(ops, B, h, w, Hc):
    """include/yat_hip.h on `s` in yat_dwconv_glu_bwd (round-5 advisor): with du given, no pass-2 kernel -- band, global-z
    or the fallback for w > 64 / odd channel counts -- reads the `s` argument; each recomputes s = bf16(z sigmoid(z)) from z
    with the library's own SiLU.  So dwdw cannot depend on which kernel a shape selects: garbage in `s` changes nothing, and
    dwdw equals the fp32 sum over pixels of (library SiLU(z)) x du taps."""
    M = B * h * w
    z = rnd(M, 2 * Hc, seed=61)
    wdw, bdw = rnd(2 * Hc, 9, scale=1 / 3, seed=62), rnd(2 * Hc, scale=0.1, seed=63)
    du = rnd(M, 2 * Hc, scale=0.3, seed=64)
    s_lib = ops.act_fwd(z, "silu", torch.empty_like(z))
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, Hc), dtype=torch.uint8, device=DEV)
    outs = []
    for s_arg in (s_lib, torch.full_like(z, float("nan")), rnd(M, 2 * Hc, seed=65)):
        dz, dw, db = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
        ops.dwconv_glu_bwd(s_arg, z, B, h, w, Hc, wdw, bdw, None, dz, dw, db, ws, du=du)
        outs.append((dz, dw, db))
    for o in outs[1:]:
        for a, b_, nm in zip(outs[0], o, ("dz", "dw", "db")):
            assert torch.equal(a, b_), f"pass 2 read the s argument ({nm}, {h}x{w}x{Hc})"
    # dW against the library's s: dW[c, tap(di, dj)] = sum_{b,i,j} s[b, i+di, j+dj, c] du[b, i, j, c]
    si = s_lib.float().view(B, h, w, 2 * Hc).permute(0, 3, 1, 2)
    dui = du.float().view(B, h, w, 2 * Hc).permute(0, 3, 1, 2)
    sp = F.pad(si, (1, 1, 1, 1))
    ref = torch.stack([(sp[:, :, a:a + h, b_:b_ + w] * dui).sum((0, 2, 3)) for a in range(3) for b_ in range(3)], 1)
    close(outs[0][1], ref.to(BF), f"dwconv_dw_from_lib_s {h}x{w}x{Hc}", atol=2e-2)


# ------------------------------------------------------------------------------------------------ elementwise / recipe
def test_elementwise(ops):
    x, dy = rnd(1000, 37, seed=41), rnd(1000, 37, seed=42)      # numel not a multiple of 8 -> scalar tail
    for act, f in (("silu", F.silu), ("gelu_tanh", lambda t: F.gelu(t, approximate="tanh"))):
        close(ops.act_fwd(x, act), f(x.float()).to(BF), f"act_fwd_{act}")
        xr = x.float().requires_grad_(True)
        f(xr).backward(dy.float())
        close(ops.act_bwd(x, dy, act), xr.grad.to(BF), f"act_bwd_{act}", atol=1e-4)
    close(ops.add_bf16(x, dy), (x.float() + dy.float()).to(BF), "add")
    f32 = torch.randn(999, device=DEV)
    assert torch.equal(ops.f32_to_bf16(f32), f32.to(BF))


def test_timestep_embed(ops):
    t = torch.tensor([2.9940121, 957.3083, 510.554, 1000.0, 0.0], device=DEV)
    out = ops.timestep_embed(t, 256)
    half = 128
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=DEV) / half)
    arg = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1).to(BF)
    close(out, ref, "timestep_embed", tol=3e-3, ulps=2, atol=2e-3)


def test_pad_mask(ops):
    B, T, Cd = 4, 32, 96
    lens = [5, 32, 1, 17]
    embs = [rnd(L, Cd, seed=50 + i) for i, L in enumerate(lens)]
    src = torch.cat(embs)
    offs = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    dst = torch.full((B, T, Cd), 7.0, dtype=BF, device=DEV)
    mask = torch.full((B, T), -1, dtype=torch.int64, device=DEV)
    bias = torch.full((B, T), 3.0, device=DEV)
    kvl = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.pad_mask(src, offs, B, T, Cd, dst, mask, bias, kvl)
    for b, (L, e) in enumerate(zip(lens, embs)):
        assert torch.equal(dst[b, :L], e)
        if L < T:
            assert dst[b, L:].abs().max().item() == 0
            assert (bias[b, L:] == -9984.0).all()
        assert mask[b].tolist() == [1] * L + [0] * (T - L)
        assert bias[b, :L].abs().max().item() == 0
    assert kvl.tolist() == lens


def test_flow_mix_and_mse(ops):
    B, per = 4, 8 * 5 * 7
    x, n = rnd(B, per, scale=0.5, seed=60), rnd(B, per, seed=61)
    sig = torch.tensor([0.957, 0.5117, 1.0, 0.003], dtype=BF, device=DEV)
    noisy, target = ops.flow_mix(x, n, sig)
    s = sig.float()[:, None]
    ref = rb(rb(rb(1.0 - s) * x.float()) + rb(s * n.float()))
    assert torch.equal(noisy.float(), ref)
    assert torch.equal(target.float(), rb(n.float() - x.float()))
    pred = rnd(B, per, seed=62)
    loss = torch.zeros(1, device=DEV)
    dpred = torch.empty_like(pred)
    ws = torch.empty(256, device=DEV)
    ops.mse_fwd_bwd(pred, target, loss, dpred, ws)
    d = pred.float() - target.float()
    assert abs(loss.item() - d.pow(2).mean().item()) <= 1e-5 * d.pow(2).mean().item()
    close(dpred, (2 * d / d.numel()).to(BF), "mse_dpred", atol=1e-9)


# ------------------------------------------------------------------------------------------------ optimizer
def test_clip_and_adamw_match_torch_cpu(ops):
    """The optimizer oracle is the reference's own dependency: torch.nn.utils.clip_grad_norm_ +
    torch.optim.AdamW on bf16 CPU tensors (common/trainer.py:246-248,347-348)."""
    shapes = [(64, 40), (40,), (3, 3, 8), (1000,), (8,)]
    g = torch.Generator().manual_seed(7)
    params = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(BF)) for s in shapes]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01)
    # flat device copies, 16-B aligned segment starts
    starts, off = [], 0
    for p in params:
        starts.append(off)
        off += (p.numel() + 7) // 8 * 8
    n = off
    flat = {k: torch.zeros(n, dtype=BF, device=DEV) for k in ("p", "g", "m", "v")}
    seg = torch.tensor(starts + [n], dtype=torch.int64, device=DEV)
    for p, s in zip(params, starts):
        flat["p"][s:s + p.numel()] = p.data.flatten().to(DEV)
    ws = torch.empty(ops.gradnorm_workspace_bytes(n, len(params)), dtype=torch.uint8, device=DEV)
    norm, coef = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    for step in range(1, 4):
        for p, s in zip(params, starts):
            p.grad = (torch.randn(p.shape, generator=g) * (3.0 if step == 1 else 0.02)).to(BF)
            flat["g"][s:s + p.numel()] = p.grad.flatten().to(DEV)
        total = torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        ops.gradnorm_clip(flat["g"], seg, 1.0, norm, coef, ws)
        ops.adamw_step(flat["p"], flat["g"], flat["m"], flat["v"], coef, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
        assert abs(norm.item() - total.float().item()) <= 2 ** -7 * total.float().item(), (norm.item(), total)
        bad = 0
        for p, s in zip(params, starts):
            mine = flat["p"][s:s + p.numel()].cpu().view(p.shape)
            bad += (mine != p.data).sum().item()
            close(mine, p.data, f"adamw_step{step}", tol=1e-3, ulps=1)
        print(f"[parity] adamw step {step}: {bad} / {n} params differ bitwise from torch CPU")
        assert bad <= 0.002 * n
        assert flat["g"].abs().max().item() == 0.0     # zero_grad fused


def test_ema_matches_emamodel_restatement():
    """EMAModel.step (common/trainer.py:266-268,350-351) fused into the AdamW launch: 12 optimizer steps through
    FlatAdamW(use_ema=True) against torch CPU AdamW + the oracle's EMAModel restatement on bf16 tensors, including the decay
    warm-up (0 on the first step, (1+s)/(10+s) after, capped at 0.999).  The schedule is compared exactly; the shadows may
    differ only where the parameters already do (the AdamW kernel is bit-exact but for rare 1-ulp ties)."""
    from types import SimpleNamespace
    from oracle.recipe_ref import EMAModelRef
    from yat_amd.optim import FlatAdamW
    shapes = [(96, 40), (40,), (3, 3, 16), (2000,), (8,)]
    g = torch.Generator().manual_seed(11)
    params = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(BF)) for s in shapes]
    opt = torch.optim.AdamW(params, lr=2e-2, weight_decay=0.01)
    ema = EMAModelRef(params, decay=0.999)
    starts, off = [], 0
    for p in params:
        starts.append(off)
        off += (p.numel() + 7) // 8 * 8
    model = SimpleNamespace(flat_param=torch.zeros(off, dtype=BF, device=DEV), flat_grad=torch.zeros(off, dtype=BF, device=DEV),
                            seg_start=torch.tensor(starts + [off], dtype=torch.int64), numel_flat=off,
                            bucket_bounds=[(0, off)], param_events=None, join_pending_update=lambda: None)
    for p, s in zip(params, starts):
        model.flat_param[s:s + p.numel()] = p.data.flatten().to(DEV)
    hip = FlatAdamW(model, lr=2e-2, weight_decay=0.01, max_grad_norm=1.0, use_ema=True, ema_decay=0.999)
    for step in range(1, 13):
        for p, s in zip(params, starts):
            p.grad = (torch.randn(p.shape, generator=g) * (3.0 if step % 5 == 1 else 0.05)).to(BF)
            model.flat_grad[s:s + p.numel()] = p.grad.flatten().to(DEV)
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        ema.step(params)
        hip.step()
        assert hip._ema_decay_now() == ema.cur_decay_value, (step, hip._ema_decay_now(), ema.cur_decay_value)
        bad_p = bad_s = 0
        for p, sh, s in zip(params, ema.shadow_params, starts):
            mine_p = model.flat_param[s:s + p.numel()].cpu().view(p.shape)
            mine_s = hip.ema_shadow[s:s + p.numel()].cpu().view(p.shape)
            bad_p += (mine_p != p.data).sum().item()
            bad_s += (mine_s != sh).sum().item()
            close(mine_s, sh, f"ema_shadow step {step}", tol=1e-3, ulps=1)
        print(f"[parity] ema step {step:2d}: decay={ema.cur_decay_value:.6f}; shadow elements differing bitwise "
              f"{bad_s} / {off} (parameters {bad_p})")
        assert bad_s <= max(2 * bad_p, 0.002 * off)
    assert ema.cur_decay_value == (1 + 11) / (10 + 11)


@pytest.mark.parametrize("act", ["gelu_tanh", "silu"])
@pytest.mark.parametrize("M,N,K", [(300, 328, 96), (1024, 1152, 288), (520, 4608, 1152)])
def test_dgrad_with_activation_backward_epilogue(ops, act, M, N, K):
    """dz = (dy W) * act'(z) in the GEMM epilogue == linear_dgrad followed by act_bwd, bit for bit (the intermediate is
    rounded to bf16 in both), and close to the fp32 formula."""
    dy, w, z = rnd(M, K, seed=90), rnd(K, N, scale=K ** -0.5, seed=91), rnd(M, N, seed=92)
    fused = ops.linear_dgrad_act(dy, w, z, act)
    d = ops.linear_dgrad(dy, w)
    unfused = ops.act_bwd(z, d, act)
    assert torch.equal(fused, unfused)
    zf = z.float().requires_grad_(True)
    (F.gelu(zf, approximate="tanh") if act == "gelu_tanh" else F.silu(zf)).backward(rb(dy.float() @ w.float()))
    # against torch: the bf16 rounding of the intermediate can flip with the accumulation order, so one extra ulp
    close(fused, zf.grad.to(BF), f"dgrad_act_{act} {M}x{N}x{K}", ulps=3.0)


@pytest.mark.parametrize("M,N,K", [(2240, 512, 4096), (520, 648, 2048 + 96), (256, 256, 64), (4480, 2240, 1000)])
def test_wgrad_with_fused_bias_gradient(ops, M, N, K):
    """yat_gemm_epilogue.a_rowsum_out: the weight-gradient GEMM dW = dy^T x also returns the bias gradient (column sums of dy,
    i.e. row sums of its A operand) from one extra MFMA per A fragment against ones -- vs torch, plain and accumulating, and
    the weight gradient itself must be bit-identical to the launch without it."""
    dy, x = rnd(K, M, scale=0.5, seed=120), rnd(K, N, seed=121)          # wgrad layout: tokens are the reduction dimension
    ref_w = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm(dy, x, ref_w, a_t=True, b_t=True, M=M, N=N, K=K, variant=4)
    out, db = torch.empty(M, N, dtype=BF, device=DEV), torch.full((M,), 7.0, dtype=BF, device=DEV)
    ops.gemm(dy, x, out, a_t=True, b_t=True, M=M, N=N, K=K, a_rowsum=db)
    assert torch.equal(out, ref_w)
    ref_b = dy.float().sum(0)
    close(db, ref_b.to(BF), f"wgrad_rowsum {M}x{N}x{K}")
    # accumulate (gradient accumulation): dW += and db += with the bf16 rounding of the += convention
    ops.gemm(dy, x, out, a_t=True, b_t=True, M=M, N=N, K=K, residual=out, a_rowsum=db, a_rowsum_accumulate=True)
    close(db, (rb(ref_b) + rb(ref_b)).to(BF), f"wgrad_rowsum_acc {M}x{N}x{K}")
    # through the helpers the models call: single launch and grouped launch
    w2, b2 = torch.empty(M, N, dtype=BF, device=DEV), torch.empty(M, dtype=BF, device=DEV)
    ops.linear_wgrad(dy, x, w2, bias_grad=b2)       # fused from 96 tiles on, else the separate column-sum pass: same contract
    if ((M + 255) // 256) * ((N + 255) // 256) >= 96:
        assert torch.equal(w2, ref_w)
    else:
        close(w2, ref_w, "linear_wgrad (policy may split K)")
    close(b2, ref_b.to(BF), "linear_wgrad bias_grad")
    w3, b3 = torch.empty(M, N, dtype=BF, device=DEV), torch.empty(M, dtype=BF, device=DEV)
    w4 = torch.empty(N, N, dtype=BF, device=DEV)
    ops.wgrad_grouped([(dy, x, w3, b3), (x, x, w4)])
    assert torch.equal(w3, ref_w)
    close(b3, ref_b.to(BF), "wgrad_grouped bias_grad")
    # argument checks: only the weight-gradient layout, no split-K, not the 320-wide tile
    from yat_amd import lib as L
    with pytest.raises(L.YatLibraryError):
        ops.gemm(x, x, w4, M=N, N=N, K=K, a_rowsum=b2[:N].contiguous())          # forward layout
    with pytest.raises(L.YatLibraryError):
        ops.gemm(dy, x, out, a_t=True, b_t=True, M=M, N=N, K=K, a_rowsum=db, variant=204)
