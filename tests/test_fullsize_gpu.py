"""Parity at BASELINE config-2 sizes (SANA-1.6B: B=8, N=1024, T=512, D=2240, Hc=5600) -- GPU.

The CPU oracle needs seconds per *block* at these sizes, so the full-size checks use (a) a plain fp32 PyTorch
restatement evaluated on the GPU (test-side checker only) with the same bf16 tolerances as tests/test_kernels_gpu.py,
and (b) size-independent properties: linearity of the GEMM in its left operand, a kernel run on the whole batch equals
the same kernel run image by image (bit-exact), the optimizer on two halves of the flat buffer equals the whole.
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_kernels_gpu import BF, DEV, close, as_good_as, rel, _collect_failures, ops  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

B, N, T, D, HC, H1, H2, DH2 = 8, 1024, 512, 2240, 5600, 70, 20, 112
M, MT = B * N, B * T


def close_big(a, b, name, tol=2e-3, ulps=2.0, frac=1e-4, hard_ulps=8.0):
    """Full-size variant of close(): with 10^7..10^8 elements a few always straddle a bf16 rounding boundary of an
    intermediate differently from the checker, so the element-wise bound is statistical: at most `frac` of the elements
    beyond `ulps` bf16 ulps of the local magnitude, none beyond `hard_ulps`; the relative L2 bound is unchanged."""
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all(), f"{name}: non-finite output"
    r = rel(a, b)
    scale_ = 2.0 ** -8 * b.abs() + 1e-6 + 1e-3 * b.abs().mean()
    err = (a - b).abs() / scale_
    over = (err > ulps).float().mean().item()
    worst = err.max().item()
    print(f"[parity] {name}: rel_l2={r:.3e} frac>{ulps}ulp={over:.2e} worst={worst:.1f}ulp")
    assert r <= tol, f"{name}: rel l2 {r:.3e} > {tol}"
    assert over <= frac, f"{name}: {over:.2e} of the elements beyond {ulps} ulps"
    assert worst <= hard_ulps, f"{name}: worst element {worst:.1f} ulps"


def grnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(BF)


# every GEMM of one transformer block, forward / dgrad / wgrad, through the shape policy (tile variant, split-K)
SHAPES = [("qkv", 3 * D, D, M), ("out", D, D, M), ("kv", 2 * D, D, MT), ("inv", 2 * HC, D, M), ("point", D, HC, M)]


@pytest.mark.parametrize("name,nout,nin,rows", SHAPES)
def test_block_gemms_full_size(ops, name, nout, nin, rows):
    x, w, dy = grnd(rows, nin, seed=1), grnd(nout, nin, scale=nin ** -0.5, seed=2), grnd(rows, nout, scale=0.05, seed=3)
    y = ops.linear_fwd(x, w)
    close_big(y, (x.float() @ w.float().T).to(BF), f"{name}_fwd {rows}x{nout}x{nin}")
    dx = ops.linear_dgrad(dy, w)
    close_big(dx, (dy.float() @ w.float()).to(BF), f"{name}_dgrad")
    dw = torch.empty(nout, nin, dtype=BF, device=DEV)
    ops.linear_wgrad(dy, x, dw)
    close_big(dw, (dy.float().T @ x.float()).to(BF), f"{name}_wgrad", tol=3e-3)
    # linearity in the left operand: (2x) W^T == 2 (x W^T) exactly (a power-of-two scale commutes with every rounding)
    y2 = ops.linear_fwd((x.float() * 2).to(BF), w)
    assert torch.equal(y2.float(), y.float() * 2), f"{name}: GEMM not linear under a power-of-two scale"


def test_gated_residual_epilogue_full_size(ops):
    x, w, res = grnd(M, D, seed=4), grnd(D, D, scale=D ** -0.5, seed=5), grnd(M, D, seed=6)
    bias, mod = grnd(D, scale=0.1, seed=7), grnd(B, 6 * D, scale=0.5, seed=8)
    gate = mod[:, 2 * D:3 * D]
    out, lin = torch.empty(M, D, dtype=BF, device=DEV), torch.empty(M, D, dtype=BF, device=DEV)
    ops.linear_fwd(x, w, bias, out=out, aux_out=lin, gate=gate, ld_gate=6 * D, residual=res, rows_per_batch=N)
    lin_ref = (x.float() @ w.float().T + bias.float()).to(BF)
    close_big(lin, lin_ref, "gated_lin")
    gl = (gate.float().repeat_interleave(N, 0) * lin_ref.float()).to(BF)
    close_big(out, (res.float() + gl.float()).to(BF), "gated_out")


def test_ln_modulate_full_size(ops):
    x, mod = grnd(M, D, seed=9), grnd(B, 6 * D, scale=0.3, seed=10)
    y, mean, rstd = ops.ln_modulate_fwd(x, mod[:, 0:D], mod[:, D:2 * D], 6 * D, N, 1e-6)
    xf = x.float()
    ln = F.layer_norm(xf, (D,), None, None, 1e-6).to(BF).float()
    sc = (1 + mod[:, D:2 * D].float()).to(BF).float().repeat_interleave(N, 0)
    ref = ((ln * sc).to(BF).float() + mod[:, 0:D].float().repeat_interleave(N, 0)).to(BF)
    close_big(y, ref, "ln_modulate_fwd_full")
    # whole batch == image by image, bit for bit
    y1 = torch.empty_like(y)
    for b in range(B):
        r = slice(b * N, (b + 1) * N)
        ops.ln_modulate_fwd(x[r], mod[b:b + 1, 0:D], mod[b:b + 1, D:2 * D], 6 * D, N, 1e-6, y1[r], mean[r].clone(),
                            rstd[r].clone())
    assert torch.equal(y, y1)


def test_linear_attention_full_size(ops):
    qkv = grnd(M, 3 * D, seed=11)
    out = torch.empty(M, D, dtype=BF, device=DEV)
    st = torch.empty(B * H1 * 33 * 32, dtype=torch.float32, device=DEV)
    ops.linear_attn_fwd(qkv, B, N, H1, D, 2 * D, out, st)
    q, k, v = (qkv[:, i * D:(i + 1) * D].float().view(B, N, H1, 32).permute(0, 2, 1, 3) for i in range(3))
    q, k = F.relu(q), F.relu(k)
    vp = F.pad(v, (0, 1), value=1.0)                              # [B,H,N,33]
    s = torch.einsum("bhnc,bhnd->bhcd", vp, k)                    # [33, 32] state
    u = torch.einsum("bhnd,bhcd->bhnc", q, s)
    ref = (u[..., :32] / (u[..., 32:] + 1e-15)).permute(0, 2, 1, 3).reshape(M, D)
    close_big(out, ref.to(BF), "linear_attn_fwd_full", tol=3e-3)
    # images are independent: one image alone gives the same bits
    out1 = torch.empty(N, D, dtype=BF, device=DEV)
    ops.linear_attn_fwd(qkv[3 * N:4 * N], 1, N, H1, D, 2 * D, out1, st[:H1 * 33 * 32].clone())
    assert torch.equal(out1, out[3 * N:4 * N])


def test_cross_attention_full_size(ops):
    lens = [20, 64, 100, 160, 200, 256, 300, 130]
    q, kv = grnd(M, D, seed=12), grnd(MT, 2 * D, seed=13)
    mask = torch.zeros(B, T, device=DEV)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
    bias = ((1 - mask.to(BF)) * -10000.0).float().contiguous()
    kvl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    out, lse = torch.empty(M, D, dtype=BF, device=DEV), torch.empty(B, H2, N, device=DEV)
    scale = 1 / math.sqrt(DH2)
    ops.sdpa_fwd(q, kv[:, :D], kv[:, D:], B, N, T, H2, DH2, scale, bias, kvl, out, lse)
    qh = q.float().view(B, N, H2, DH2).transpose(1, 2)
    kh = kv[:, :D].float().view(B, T, H2, DH2).transpose(1, 2)
    vh = kv[:, D:].float().view(B, T, H2, DH2).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=bias[:, None, None, :]).transpose(1, 2).reshape(M, D)
    # outputs are averages with cancellation (small against their terms): judge against torch's own bf16 kernel
    flow = F.scaled_dot_product_attention(qh.to(BF), kh.to(BF), vh.to(BF),
                                          attn_mask=bias[:, None, None, :].to(BF)).transpose(1, 2).reshape(M, D)
    as_good_as(out, flow, ref, "sdpa_fwd_full")
    # backward: dense grid == compact work list, bit for bit, at full size
    dout = grnd(M, D, scale=0.1, seed=14)
    outs = []
    for work in (None, ops.kv_work_list(lens, T, DEV)):
        dq, dkv = torch.empty_like(q), torch.full_like(kv, float("nan"))
        delta = torch.empty(B, H2, N, device=DEV)
        ops.sdpa_bwd(q, kv[:, :D], kv[:, D:], B, N, T, H2, DH2, scale, bias, kvl, out, dout, lse, delta, dq, dkv[:, :D],
                     dkv[:, D:], work=work)
        outs.append((dq, dkv))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][1].float()).all()
    # rows of fully masked key tiles get exact zeros
    for b, L in enumerate(lens):
        first_dead = ((L + 63) // 64) * 64
        assert outs[1][1][b * T + first_dead:(b + 1) * T].abs().max().item() == 0.0


def test_dwconv_glu_full_size(ops):
    h = w = 32
    z = grnd(M, 2 * HC, seed=15)
    s = F.silu(z.float()).to(BF)
    wdw, bdw = grnd(2 * HC, 9, scale=1 / 3, seed=16), grnd(2 * HC, scale=0.1, seed=17)
    y = torch.empty(M, HC, dtype=BF, device=DEV)
    ops.dwconv_glu_fwd(s, B, h, w, HC, wdw, bdw, y)
    si = s.float().view(B, h, w, 2 * HC).permute(0, 3, 1, 2)
    u = F.conv2d(si, wdw.float().view(2 * HC, 1, 3, 3), bdw.float(), padding=1, groups=2 * HC).to(BF).float()
    a, g = torch.chunk(u, 2, dim=1)
    ref = (a * F.silu(g).to(BF).float()).permute(0, 2, 3, 1).reshape(M, HC)
    close_big(y, ref.to(BF), "dwconv_glu_fwd_full")
    # backward: whole batch == two half batches for dz (bit for bit); weight gradients add up
    dy = grnd(M, HC, scale=0.1, seed=18)
    ws = torch.empty(ops.dwconv_glu_bwd_workspace_bytes(B, h, w, HC), dtype=torch.uint8, device=DEV)
    dz, dw, db = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
    ops.dwconv_glu_bwd(s, z, B, h, w, HC, wdw, bdw, dy, dz, dw, db, ws)
    dz2, dw2, db2 = torch.empty_like(z), torch.empty_like(wdw), torch.empty_like(bdw)
    hb = B // 2
    for i, acc in ((0, False), (1, True)):
        r = slice(i * hb * N, (i + 1) * hb * N)
        ops.dwconv_glu_bwd(s[r], z[r], hb, h, w, HC, wdw, bdw, dy[r], dz2[r], dw2, db2, ws, accumulate=acc)
    assert torch.equal(dz, dz2)
    # (two bf16-rounded partial sums added in bf16 vs one rounding of the whole sum: one extra rounding)
    assert rel(dw2, dw) <= 5e-3 and rel(db2, db) <= 5e-3, (rel(dw2, dw), rel(db2, db))


def test_optimizer_full_size_halves_equal_whole(ops):
    n = 64 * 1024 * 1024                        # 64 M parameters per array (the step has 1.6 B; elementwise -> any size)
    g = torch.Generator(device=DEV).manual_seed(19)

    def mk():
        return [(torch.randn(n, generator=g, device=DEV) * sc).to(BF) for sc in (0.02, 1e-3, 1e-3, 1e-6)]
    p, gr, m, v = mk()
    v.abs_()
    whole = [t.clone() for t in (p, gr, m, v)]
    coef = torch.full((1,), 0.37, dtype=torch.float32, device=DEV)
    ops.adamw_step(*whole, coef, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, zero_grad=False)
    halves = [t.clone() for t in (p, gr, m, v)]
    k = n // 2
    for sl in (slice(0, k), slice(k, n)):
        ops.adamw_step(*(t[sl] for t in halves), coef, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, zero_grad=False)
    for a, b_ in zip(whole, halves):
        assert torch.equal(a, b_)
    bg = [t.clone() for t in (p, gr, m, v)]      # the 48-VGPR background variant (one workgroup per CU): same bits
    ops.adamw_step(*bg, coef, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, zero_grad=False, background=256)
    for a, b_ in zip(whole, bg):
        assert torch.equal(a, b_)
    assert not torch.equal(whole[0], p)         # the step moved the parameters


def test_pixart_self_attention_full_size(ops):
    """PixArt-Sigma attn1 at 1024 px: 16 heads x 72 over N = T = 4096 tokens, q | k | v = the column blocks of the fused
    projection, 128-query workgroups (the long-sequence variant).  Forward and all three gradients against torch (fp32
    truth and torch's own bf16 kernel as the yardstick) on two images; ragged N (4096 - 40) exercises the tail masking."""
    Bp, Hp, dh = 2, 16, 72
    Dp = Hp * dh
    scale = 1 / math.sqrt(dh)
    for Np in (4096, 4096 - 40):
        qkv = grnd(Bp * Np, 3 * Dp, seed=31)
        dout = grnd(Bp * Np, Dp, scale=0.1, seed=32)
        out, lse = torch.empty(Bp * Np, Dp, dtype=BF, device=DEV), torch.empty(Bp, Hp, Np, device=DEV)
        zero = full = None                       # no key bias, as yat_amd/pixart.py launches attn1 (the no-bias instantiations)
        q, k, v = qkv[:, :Dp], qkv[:, Dp:2 * Dp], qkv[:, 2 * Dp:]
        ops.sdpa_fwd(q, k, v, Bp, Np, Np, Hp, dh, scale, zero, full, out, lse)
        dqkv = torch.full_like(qkv, float("nan"))
        ops.sdpa_bwd(q, k, v, Bp, Np, Np, Hp, dh, scale, zero, full, out, dout, lse, torch.empty(Bp, Hp, Np, device=DEV),
                     dqkv[:, :Dp], dqkv[:, Dp:2 * Dp], dqkv[:, 2 * Dp:])
        assert torch.isfinite(dqkv.float()).all()

        def torch_path(dt):
            t = qkv.to(dt).clone().requires_grad_(True)
            heads = lambda x: x.reshape(Bp, Np, Hp, dh).transpose(1, 2)
            o = F.scaled_dot_product_attention(heads(t[:, :Dp]), heads(t[:, Dp:2 * Dp]), heads(t[:, 2 * Dp:]))
            o = o.transpose(1, 2).reshape(Bp * Np, Dp)
            o.backward(dout.to(dt))
            return o.detach(), t.grad
        o32, g32 = torch_path(torch.float32)
        obf, gbf = torch_path(BF)
        as_good_as(out, obf, o32, f"pixart_sdpa_fwd N={Np}")
        for name, sl in (("dq", slice(0, Dp)), ("dk", slice(Dp, 2 * Dp)), ("dv", slice(2 * Dp, 3 * Dp))):
            as_good_as(dqkv[:, sl], gbf[:, sl], g32[:, sl], f"pixart_sdpa_{name} N={Np}")


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 (SD3.5-Medium at 1024 px) at its own token counts: joint attention over 4096 image + 333 text tokens,
# 24 heads x 64 (train_sd35.py:165-194 -> JointAttnProcessor2_0 [RECALL]); the head-dim instantiation <2, 4>, the
# 128 / 192-query workgroups, the XCD-contiguous order and the dense dK/dV grid only run at these sizes.
SD_B, SD_N, SD_T, SD_H, SD_DH = 2, 4096, 333, 24, 64


@pytest.mark.parametrize("L,tag", [(SD_N + SD_T, "joint 4096+333"), (SD_N, "image-only second attention 4096")])
def test_sd35_attention_full_size(ops, L, tag):
    """The joint attention (L = 4429) and the dual blocks' image-only attention (L = 4096) exactly as yat_amd/sd3.py launches
    them -- q | k | v = column blocks of the [B*L, 3D] joint buffer, no key bias -- forward, dQ, dK, dV against torch
    (fp32 truth, torch's own bf16 kernel as the yardstick) on two images."""
    Bs, Hs, dh = SD_B, SD_H, SD_DH
    Ds = Hs * dh
    scale = 1 / math.sqrt(dh)
    qkv = grnd(Bs * L, 3 * Ds, seed=41)
    dout = grnd(Bs * L, Ds, scale=0.1, seed=42)
    out, lse = torch.empty(Bs * L, Ds, dtype=BF, device=DEV), torch.empty(Bs, Hs, L, device=DEV)
    zero = full = None                           # no key bias (JointAttnProcessor2_0 passes no mask): the no-bias instantiations
    q, k, v = qkv[:, :Ds], qkv[:, Ds:2 * Ds], qkv[:, 2 * Ds:]
    ops.sdpa_fwd(q, k, v, Bs, L, L, Hs, dh, scale, zero, full, out, lse)
    dqkv = torch.full_like(qkv, float("nan"))
    ops.sdpa_bwd(q, k, v, Bs, L, L, Hs, dh, scale, zero, full, out, dout, lse, torch.empty(Bs, Hs, L, device=DEV),
                 dqkv[:, :Ds], dqkv[:, Ds:2 * Ds], dqkv[:, 2 * Ds:])
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(lse).all()

    def torch_path(dt):
        t = qkv.to(dt).clone().requires_grad_(True)
        heads = lambda x: x.reshape(Bs, L, Hs, dh).transpose(1, 2)
        o = F.scaled_dot_product_attention(heads(t[:, :Ds]), heads(t[:, Ds:2 * Ds]), heads(t[:, 2 * Ds:]))
        o = o.transpose(1, 2).reshape(Bs * L, Ds)
        o.backward(dout.to(dt))
        return o.detach(), t.grad
    o32, g32 = torch_path(torch.float32)
    obf, gbf = torch_path(BF)
    as_good_as(out, obf, o32, f"sd35_sdpa_fwd {tag}")
    for name, sl in (("dq", slice(0, Ds)), ("dk", slice(Ds, 2 * Ds)), ("dv", slice(2 * Ds, 3 * Ds))):
        as_good_as(dqkv[:, sl], gbf[:, sl], g32[:, sl], f"sd35_sdpa_{name} {tag}")
    # a second run is bit-identical (no atomics, fixed reduction order)
    out2, lse2, dqkv2 = torch.empty_like(out), torch.empty_like(lse), torch.empty_like(qkv)
    ops.sdpa_fwd(q, k, v, Bs, L, L, Hs, dh, scale, zero, full, out2, lse2)
    ops.sdpa_bwd(q, k, v, Bs, L, L, Hs, dh, scale, zero, full, out2, dout, lse2, torch.empty(Bs, Hs, L, device=DEV),
                 dqkv2[:, :Ds], dqkv2[:, Ds:2 * Ds], dqkv2[:, 2 * Ds:])
    assert torch.equal(out, out2) and torch.equal(lse, lse2) and torch.equal(dqkv, dqkv2)


def test_sd35_qknorm_concat_full_rows(ops):
    """``yat_qknorm_concat_fwd/bwd`` at B x 4429 rows (B = 8, the bench's batch): per-head RMSNorm of q | k of both streams +
    the row concatenation of JointAttnProcessor2_0 [RECALL], forward bit-exact against the bf16 restatement, backward against
    fp32 autograd of the same function (evaluated on the GPU: 163 M elements)."""
    Bq, N_, T_, H_, dh = 8, SD_N, SD_T, SD_H, SD_DH
    Dq, L, eps = H_ * dh, SD_N + SD_T, 1e-6
    qkv_i, qkv_t = grnd(Bq * N_, 3 * Dq, seed=51), grnd(Bq * T_, 3 * Dq, seed=52)
    ws = [(1.0 + 0.2 * torch.randn(dh, generator=torch.Generator().manual_seed(60 + j))).to(BF).to(DEV) for j in range(4)]
    joint = torch.empty(Bq * L, 3 * Dq, dtype=BF, device=DEV)
    rstd = torch.empty(Bq * L, 2 * H_, dtype=torch.float32, device=DEV)
    ops.qknorm_concat_fwd(qkv_i, qkv_t, Bq, N_, T_, H_, dh, eps, ws[0], ws[1], ws[2], ws[3], joint, rstd)

    def rms(x, w, dt):
        y = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)
        if dt == BF:
            y = y.to(BF)
        return (y * w.to(dt)).to(dt)

    def ref(dt):
        xi = qkv_i.detach().clone().to(dt).view(Bq, N_, 3, H_, dh).requires_grad_(True)
        xt = qkv_t.detach().clone().to(dt).view(Bq, T_, 3, H_, dh).requires_grad_(True)
        w = [t.detach().clone().to(dt).requires_grad_(True) for t in ws]
        qq = torch.cat([rms(xi[:, :, 0], w[0], dt), rms(xt[:, :, 0], w[2], dt)], 1)
        kk = torch.cat([rms(xi[:, :, 1], w[1], dt), rms(xt[:, :, 1], w[3], dt)], 1)
        vv = torch.cat([xi[:, :, 2], xt[:, :, 2]], 1)
        return torch.stack([qq, kk, vv], dim=2).reshape(Bq * L, 3 * Dq), (xi, xt), w
    with torch.no_grad():
        o_bf, _, _ = ref(BF)
    # torch's mean over the 64 head channels sums in another order than the kernel's butterfly: the fp32 statistics differ
    # in the last bit here and there, and over 163 M elements a few of those flip a bf16 rounding -- "an ulp or two on a
    # vanishing share of the elements" (bit-exact at the sizes of tests/test_sd3_gpu.py, on the CPU restatement)
    bad = (joint != o_bf)
    frac = bad.float().mean().item()
    assert frac <= 1e-4, frac
    if frac:
        a, b = joint[bad].float(), o_bf[bad].float()
        assert ((a - b).abs() <= 2.0 ** -6 * b.abs() + 1e-30).all()         # two roundings (normalise, then * weight): <= 2 ulps
    del o_bf, bad
    print(f"[parity] qknorm_concat {Bq * L} rows forward: {frac:.2e} of the elements differ from torch's restatement, none by more "
          f"than 2 bf16 ulps (fp32 summation order of the 64-wide mean square; bit-exact at the sizes of tests/test_sd3_gpu.py)")
    dj = grnd(Bq * L, 3 * Dq, seed=55)
    o32, parts, w32 = ref(torch.float32)
    o32.backward(dj.float())
    del o32
    dqi, dqt = torch.empty_like(qkv_i), torch.empty_like(qkv_t)
    dws = [torch.zeros(dh, dtype=BF, device=DEV) for _ in range(4)]
    wsb = torch.empty(ops.qknorm_concat_bwd_workspace_bytes(Bq, N_, T_, dh), dtype=torch.uint8, device=DEV)
    ops.qknorm_concat_bwd(qkv_i, qkv_t, Bq, N_, T_, H_, dh, ws[0], ws[1], ws[2], ws[3], rstd, dj, dqi, dqt, dws[0], dws[1],
                          dws[2], dws[3], wsb)
    e_i = rel(dqi, parts[0].grad.reshape(Bq * N_, 3 * Dq))
    e_t = rel(dqt, parts[1].grad.reshape(Bq * T_, 3 * Dq))
    e_w = [rel(dws[j], w32[j].grad) for j in range(4)]
    print(f"[parity] qknorm_concat B={Bq} N={N_} T={T_} H={H_} dh={dh} ({Bq * L} rows): fwd bit-exact; d_img {e_i:.3e} "
          f"d_txt {e_t:.3e} d_w {max(e_w):.3e}")
    assert e_i <= 4e-3 and e_t <= 4e-3 and max(e_w) <= 8e-3
