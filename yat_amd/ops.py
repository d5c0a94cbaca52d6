"""Thin torch-tensor wrappers over the C ABI (yat_amd.lib).  torch supplies device memory and the
stream; every computation happens in libyat_hip.so.  No CPU fallback: a CPU tensor is an error.

Each wrapper names the C entry point it calls; see include/yat_hip.h for the reference call site
that entry point replaces.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch

from . import lib as _l

BF16 = torch.bfloat16
ACT = {"none": 0, "silu": 1, "gelu_tanh": 2}

# Split-K workspace for the GEMM policy (fp32 partial slabs), one per device, allocated on first use.
_GEMM_WS = {}
GEMM_WS_BYTES = 96 << 20


def _gemm_ws(device):
    # one split-K workspace per (device, stream): GEMMs on different streams may run concurrently
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = torch.empty(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = ws
    return ws


# Optional live profiler for bench.py: when set to a list, every GEMM launch appends
# (flops, start_event, end_event) recorded on the launch stream.
GEMM_TIMER = None


class Recorder:
    """Launch-plan recording (yat_amd/flat.py ``planned``).  While one is installed every C-ABI call made through ``_lib()``
    is executed AND appended as ``[fn, args]``; stream / event operations and host callbacks are appended by the model code
    through FlatParamModule's helpers as tagged entries (yat_amd/plan.py compiles the list for the C-side replay).  A step over the same buffers is then replayed as a flat list of calls -- no tensor slicing, no
    stride arithmetic, no struct building, no stream look-ups on the host.  ``dynamic[name]`` lists (entry, argument) slots
    whose integer value changes from step to step (the length of the attention work list)."""

    _PURE = ("_workspace_bytes", "yat_version", "yat_gemm_epilogue_size", "yat_comm_")

    def __init__(self, real):
        self._real = real
        self.entries = []
        self.dynamic = {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if any(tag in name for tag in self._PURE):
            return fn
        entries = self.entries

        def call(*args):
            entries.append([fn, args])
            return fn(*args)
        return call

    def mark_dynamic(self, name, arg_index, mul=1, add=0):
        """The call just recorded takes a per-step integer at ``arg_index``: value(name) * mul + add."""
        self.dynamic.setdefault(name, []).append((len(self.entries) - 1, arg_index, mul, add))


RECORDER = None

# Packed text rows (yat_amd/sana.py forward_impl, kv_off): the number of text rows changes from batch to batch, and a launch
# plan must not be keyed by it (every new (bucket, row count) pair would be a fresh recording).  Inside ``with
# text_rows(n):`` the wrappers below mark the argument that carries the row count -- M of a forward / dgrad GEMM, K of a
# weight gradient, the row count of a norm / column sum / packed attention, the address of the last rows -- as the plan's
# dynamic integer "text_rows"; a replay patches them from ``plan_dynamic["text_rows"]``.  Scopes hold text-side calls only.
TEXT_ROWS = None


@contextlib.contextmanager
def text_rows(n):
    global TEXT_ROWS
    prev, TEXT_ROWS = TEXT_ROWS, n
    try:
        yield
    finally:
        TEXT_ROWS = prev


def _dyn_rows(arg_index, value, mul=1, add=0):
    if RECORDER is not None and TEXT_ROWS is not None:
        if value != TEXT_ROWS * mul + add:
            raise RuntimeError("text_rows scope around a call whose row count is not the text row count")
        RECORDER.mark_dynamic("text_rows", arg_index, mul, add)


def _lib():
    return RECORDER if RECORDER is not None else _l.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _l.YatLibraryError("yat_amd ops need device tensors (there is no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _chk_bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != BF16:
            raise TypeError(f"expected bf16, got {t.dtype}")


# ----------------------------------------------------------------------------------------------- GEMM
GEMM_STREAMS = 1


def gemm_concurrency(streams: int):
    """How many independent GEMM streams the caller keeps in flight from here on: host-side state of THIS module, handed to
    the library per call in the policy word of yat_gemm_bf16_ex (include/yat_hip.h) -- the library itself holds no policy
    state; a recorded launch plan carries the word of every call as recorded."""
    global GEMM_STREAMS
    if not 1 <= int(streams) <= 8:
        raise ValueError("gemm_concurrency: 1..8 streams")
    GEMM_STREAMS = int(streams)


def gemm(a, b, out, *, a_t=False, b_t=False, M, N, K, lda=None, ldb=None, ldc=None, bias=None, aux_out=None,
         activation="none", gate=None, ld_gate=0, residual=None, rows_per_batch=0, ld_aux=0, ld_residual=0, variant=0,
         glu_u=None, pre_add=None, dact_z=None, a_rowsum=None, a_rowsum_accumulate=False, dyn=None, a2=None, b2=None,
         k2=0, a2_group_n=0):
    """yat_gemm_bf16.  out[M,N] = epilogue(A_op @ B_op); see the header for layouts.  ``dyn`` ("M" | "K"): inside a
    ``text_rows`` scope, the dimension that is the text row count.  ``a2`` / ``b2`` / ``k2``: the second operand pair of the
    forward layout (yat_gemm_epilogue.a2): views whose row strides are lda / ldb."""
    _chk_bf16(a, b, out, bias, aux_out, gate, residual, glu_u, pre_add, dact_z, a_rowsum, a2, b2)
    lda = lda if lda is not None else (M if a_t else K)
    ldb = ldb if ldb is not None else (N if b_t else K)
    ldc = ldc if ldc is not None else N
    ep = None
    if bias is not None or aux_out is not None or activation != "none" or gate is not None or residual is not None \
            or glu_u is not None or pre_add is not None or dact_z is not None or a_rowsum is not None or a2 is not None:
        if a2 is not None and (a2.stride(0) != lda or b2.stride(0) != ldb):
            raise ValueError("gemm: the second operand pair must have the row strides of A and B")
        ep = _l.GemmEpilogue(_p(bias), _p(aux_out), ACT[activation], _p(gate), _p(residual), ld_aux, ld_gate,
                             ld_residual, rows_per_batch, _p(glu_u), 0 if glu_u is None else glu_u.stride(0),
                             _p(pre_add), 0 if pre_add is None else pre_add.stride(0),
                             _p(dact_z), 0 if dact_z is None else dact_z.stride(0),
                             _p(a_rowsum), int(bool(a_rowsum_accumulate)), _p(a2), _p(b2), int(k2), int(a2_group_n))
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = _gemm_ws(out.device)
    rc = _lib().yat_gemm_bf16_ex(int(a_t), int(b_t), M, N, K, _p(a), lda, _p(b), ldb, _p(out), ldc,
                                 C.byref(ep) if ep is not None else None,
                                 variant + (10000 * GEMM_STREAMS if GEMM_STREAMS > 1 else 0), _p(ws), ws.numel(), _stream())
    if dyn is not None and TEXT_ROWS is not None:
        _dyn_rows(2 if dyn == "M" else 4, M if dyn == "M" else K)
    if timer is not None:
        e1.record()
        nbytes = 2.0 * (M * K + K * N + M * N * (1 + (residual is not None) + (aux_out is not None) +
                                                 3 * (glu_u is not None) + (dact_z is not None)))
        timer.append((2.0 * M * N * (K + k2), e0, e1, ("nt"[int(a_t)] + "nt"[int(b_t)], M, N, K, activation,
                      gate is not None, residual is not None, aux_out is not None), nbytes))
    _l.check(rc, "yat_gemm_bf16")
    return out


def linear_fwd(x2d, w, bias=None, out=None, **ep):
    """y = x W^T (+bias ...): x2d [M,K], w [N,K] (nn.Linear layout)."""
    M, K = x2d.shape
    N = w.shape[0]
    out = out if out is not None else torch.empty(M, N, dtype=BF16, device=x2d.device)
    return gemm(x2d, w, out, M=M, N=N, K=K, bias=bias, dyn="M", **ep)


def linear_dgrad(dy2d, w, out=None, **ep):
    """dx = dy W : dy [M,N], w [N,K] -> [M,K]."""
    M, N = dy2d.shape
    K = w.shape[1]
    out = out if out is not None else torch.empty(M, K, dtype=BF16, device=dy2d.device)
    return gemm(dy2d, w, out, b_t=True, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, dyn="M", **ep)


def linear_dgrad_glu(dy2d, w, u, du):
    """GLUMBConv conv_point dgrad with the GLU backward in its epilogue: d = dy W ([M, Hc]) never reaches memory;
    du[:, :Hc] = d * SiLU(u_g), du[:, Hc:] = (d * u_a) * SiLU'(u_g) with u = [u_a | u_g] kept by the forward."""
    M, N = dy2d.shape
    K = w.shape[1]
    assert u.shape == (M, 2 * K) and du.shape == (M, 2 * K)
    return gemm(dy2d, w, du, b_t=True, M=M, N=K, K=N, lda=N, ldb=K, ldc=2 * K, glu_u=u)


def linear_dgrad_act(dy2d, w, z, act, out=None):
    """dz = (dy W) * act'(z): the activation backward in the epilogue of the dgrad GEMM that produces the activation's output
    gradient (bit-identical to linear_dgrad followed by act_bwd; the intermediate never reaches memory)."""
    M, N = dy2d.shape
    K = w.shape[1]
    assert z.shape == (M, K)
    out = out if out is not None else torch.empty(M, K, dtype=BF16, device=dy2d.device)
    return gemm(dy2d, w, out, b_t=True, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, activation=act, dact_z=z)


def wgrad_fuses_bias(N, K):
    """Does ``linear_wgrad`` take the bias gradient out of the weight-gradient GEMM itself ([N, K] weight)?  Only where the
    shape policy would not split K anyway (>= 96 tiles of 256 x 256); otherwise it is a separate column-sum pass."""
    return ((N + 255) // 256) * ((K + 255) // 256) >= 96


def linear_wgrad(dy2d, x2d, out, accumulate=False, bias_grad=None, colsum_ws=None):
    """dW = dy^T x : dy [M,N], x [M,K] -> out [N,K] (optionally += for gradient accumulation).  ``bias_grad`` [N]: the bias
    gradient (column sums of dy) from the same launch (yat_gemm_epilogue.a_rowsum_out) -- or, for the shapes whose K is split
    (``wgrad_fuses_bias``), from the separate yat_colsum_bf16 pass (``colsum_ws``: its workspace)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    # fused only where the shape policy would not split K anyway (>= 96 tiles of 256 x 256: csrc/gemm.hip est_time_256) -- the
    # fused form is one workgroup per tile over the whole K, and a 25-tile, K = 32768 weight gradient (PixArt's D x D) left
    # unsplit makes the weight-gradient stream the critical path (PixArt 228 -> 244 ms when it was fused unconditionally)
    fused = bias_grad is not None and wgrad_fuses_bias(N, K)
    r = gemm(dy2d, x2d, out, a_t=True, b_t=True, M=N, N=K, K=M, lda=dy2d.stride(0), ldb=K, ldc=K,
             residual=out if accumulate else None, a_rowsum=bias_grad if fused else None, a_rowsum_accumulate=accumulate,
             dyn="K")
    if bias_grad is not None and not fused:
        if colsum_ws is None:                                   # (the models pass their arena buffer)
            colsum_ws = torch.empty(int(_lib().yat_colsum_workspace_bytes(M, N)), dtype=torch.uint8, device=dy2d.device)
        colsum(dy2d, bias_grad, colsum_ws, accumulate=accumulate)
    return r


def wgrad_grouped(items, accumulate=False):
    """yat_gemm_grouped_bf16 for a set of weight gradients: items = [(dy [M,N], x [M,K], out [N,K][, bias_grad [N]]), ...], all
    dW = dy^T x (optionally += out) in ONE launch of 256x256 tiles (see include/yat_hip.h); a 4th member gets the bias
    gradient (column sums of dy) from the same launch."""
    n = len(items)
    probs = (_l.GemmProblem * n)()
    eps = (_l.GemmEpilogue * n)()
    flops = 0.0
    for i, item in enumerate(items):
        dy, x, out = item[:3]
        bias_grad = item[3] if len(item) > 3 else None      # optional 4th member: the bias gradient [N]
        _chk_bf16(dy, x, out, bias_grad)
        M, N = dy.shape
        K = x.shape[1]
        pr = probs[i]
        pr.M, pr.N, pr.K = N, K, M
        pr.A, pr.lda, pr.B, pr.ldb, pr.C, pr.ldc = _p(dy), dy.stride(0), _p(x), x.stride(0), _p(out), K
        if accumulate or bias_grad is not None:
            eps[i] = _l.GemmEpilogue(None, None, 0, None, _p(out) if accumulate else None, 0, 0, K, 0, None, 0, None, 0,
                                     None, 0, _p(bias_grad), int(bool(accumulate)))
            pr.epilogue = C.pointer(eps[i])
        flops += 2.0 * M * N * K
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib().yat_gemm_grouped_bf16(1, 1, n, probs, _stream())
    if timer is not None:
        e1.record()
        timer.append((flops, e0, e1, ("tt", 0, 0, 0, f"grouped x{n}", False, accumulate, False)))
    _l.check(rc, "yat_gemm_grouped_bf16")


def lokr_delta(w1, w2a, w2b, scale, delta2d):
    """delta2d [out, in] (a view with row stride) <- kron(w1, w2a @ w2b) * scale, bf16 op by op (yat_lokr_delta)."""
    (out_l, in_m), (out_k, r), in_n = w1.shape, w2a.shape, w2b.shape[1]
    assert delta2d.shape == (out_l * out_k, in_m * in_n)
    rc = _lib().yat_lokr_delta(out_l, out_k, in_m, in_n, r, _p(w1), _p(w2a), _p(w2b), float(scale), _p(delta2d),
                               delta2d.stride(0), _stream())
    _l.check(rc, "yat_lokr_delta")


def lokr_project(w1, w2a, w2b, scale, d_delta2d, dw1, dw2a, dw2b, workspace):
    (out_l, in_m), (out_k, r), in_n = w1.shape, w2a.shape, w2b.shape[1]
    rc = _lib().yat_lokr_project(out_l, out_k, in_m, in_n, r, _p(w1), _p(w2a), _p(w2b), float(scale), _p(d_delta2d),
                                 d_delta2d.stride(0), _p(dw1), _p(dw2a), _p(dw2b), _p(workspace), _stream())
    _l.check(rc, "yat_lokr_project")


def hadamard_scale(a2d, b2d, scale, out2d):
    """out = bf16(bf16(a * b) * scale) on [rows, cols] views (yat_hadamard_scale: LoHa's delta_w)."""
    _chk_bf16(a2d, b2d, out2d)
    rows, cols = a2d.shape
    rc = _lib().yat_hadamard_scale(rows, cols, _p(a2d), a2d.stride(0), _p(b2d), b2d.stride(0), float(scale), _p(out2d),
                                   out2d.stride(0), _stream())
    _l.check(rc, "yat_hadamard_scale")
    return out2d


def hadamard_bwd(dd2d, a1, a2, scale, t1, t2):
    """g = bf16(dd * scale); t1 = bf16(g * a2); t2 = bf16(g * a1) (yat_hadamard_bwd)."""
    _chk_bf16(dd2d, a1, a2, t1, t2)
    rows, cols = dd2d.shape
    rc = _lib().yat_hadamard_bwd(rows, cols, _p(dd2d), dd2d.stride(0), _p(a1), a1.stride(0), _p(a2), a2.stride(0), float(scale),
                                 _p(t1), t1.stride(0), _p(t2), t2.stride(0), _stream())
    _l.check(rc, "yat_hadamard_bwd")


def dora_delta(w, lw, mag, scaling, delta, s_buf, n_buf):
    """delta_j = s_j (W_j + scaling lw_j) - W_j with s_j = mag_j / ||W_j + scaling lw_j|| (yat_dora_delta); s, n -> fp32 buffers."""
    _chk_bf16(w, lw, mag, delta)
    rows, cols = w.shape
    rc = _lib().yat_dora_delta(rows, cols, _p(w), w.stride(0), _p(lw), lw.stride(0), _p(mag), float(scaling), _p(delta),
                               delta.stride(0), _p(s_buf), _p(n_buf), _stream())
    _l.check(rc, "yat_dora_delta")


def dora_bwd(dd, w, lw, scaling, s_buf, n_buf, t1, dmag):
    """dmag_j = (sum_l dd[j,l] u[j,l]) / n_j; t1 = bf16(bf16(s_j scaling) * dd) (yat_dora_bwd)."""
    _chk_bf16(dd, w, lw, t1, dmag)
    rows, cols = w.shape
    rc = _lib().yat_dora_bwd(rows, cols, _p(dd), dd.stride(0), _p(w), w.stride(0), _p(lw), lw.stride(0), float(scaling),
                             _p(s_buf), _p(n_buf), _p(t1), t1.stride(0), _p(dmag), _stream())
    _l.check(rc, "yat_dora_bwd")


def lokr_rows_fwd(x2d, wb, t1):
    """t1[rows, R] = x2d[rows, N] wb^T (wb [R, N]) -- the T1 product of the factored LoKr path."""
    _chk_bf16(x2d, wb, t1)
    rows, N = x2d.shape
    R = wb.shape[0]
    if wb.shape[1] != N or t1.shape != (rows, R) or not (x2d.is_contiguous() and wb.is_contiguous() and t1.is_contiguous()):
        raise ValueError("lokr_rows_fwd: shape mismatch")
    _l.check(_lib().yat_lokr_rows(rows, N, R, 0, _p(wb), _p(x2d), _p(t1), _stream()), "yat_lokr_rows")
    return t1


def lokr_rows_fwd_flat(x2d, wb, t1_flat, in_m):
    """t1_flat[m, j R + q] = (x2d wb^T)[m in_m + j, q]: T1 of the factored LoKr path written as [M, in_m R] columns of a
    row-strided view -- the layout the base GEMM takes as its second operand (``gemm(..., a2=t1_flat)``)."""
    _chk_bf16(x2d, wb, t1_flat)
    rows, N = x2d.shape
    R = wb.shape[0]
    if wb.shape[1] != N or rows % in_m or t1_flat.shape != (rows // in_m, in_m * R) or t1_flat.stride(1) != 1 \
            or not (x2d.is_contiguous() and wb.is_contiguous()):
        raise ValueError("lokr_rows_fwd_flat: shape mismatch")
    _l.check(_lib().yat_lokr_rows_fwd_flat(rows, N, R, in_m, _p(wb), _p(x2d), _p(t1_flat), t1_flat.stride(0), _stream()),
             "yat_lokr_rows_fwd_flat")
    return t1_flat


def lokr_rows_bwd(h2d, wb, dx2d):
    """dx2d[rows, N] += h2d[rows, R] wb (product rounded to bf16 first, like a GEMM's residual epilogue)."""
    _chk_bf16(h2d, wb, dx2d)
    rows, R = h2d.shape
    N = wb.shape[1]
    if wb.shape[0] != R or dx2d.shape != (rows, N) or not (h2d.is_contiguous() and wb.is_contiguous() and dx2d.is_contiguous()):
        raise ValueError("lokr_rows_bwd: shape mismatch")
    _l.check(_lib().yat_lokr_rows(rows, N, R, 1, _p(wb), _p(h2d), _p(dx2d), _stream()), "yat_lokr_rows")
    return dx2d


def dropout(x, p, seed, out=None):
    """out = bf16(x * keep / (1 - p)) with the counter-based mask of (seed, element index) -- yat_dropout."""
    _chk_bf16(x)
    out = torch.empty_like(x) if out is None else out
    if not (x.is_contiguous() and out.is_contiguous()):
        raise ValueError("dropout: contiguous tensors only")
    _l.check(_lib().yat_dropout(x.numel(), float(p), int(seed), 0, _p(x), _p(out), _stream()), "yat_dropout")
    return out


def dropout_bwd_add(g, p, seed, io):
    """io += dropout-mask(g) / (1 - p): the gradient through the mask of (seed, element index), accumulated."""
    _chk_bf16(g, io)
    if g.shape != io.shape or not (g.is_contiguous() and io.is_contiguous()):
        raise ValueError("dropout_bwd_add: shape mismatch")
    _l.check(_lib().yat_dropout(g.numel(), float(p), int(seed), 1, _p(g), _p(io), _stream()), "yat_dropout")
    return io


def lora_scatter_b(table, entries, max_out, R, scale, src_flat, dst_flat):
    """Every adapter's scaling * lora_B into the first R columns of its target's rows of ``dst_flat`` (a zero-filled shadow of the
    model's flat weights): the b2 operands of the base GEMMs (yat_lora_scatter_b; ``table``: int64 [entries, 4] on the device)."""
    _chk_bf16(src_flat, dst_flat)
    _l.check(_lib().yat_lora_scatter_b(int(entries), int(max_out), int(R), float(scale), _p(table), _p(src_flat), _p(dst_flat),
                                       _stream()), "yat_lora_scatter_b")


def rank_expand(h2d, w, io2d, scale=1.0, residual=False):
    """io[rows, N] = bf16(bf16(h w) * scale), or bf16(bf16(h w) + io) with residual -- h2d [rows, R], w [R, N]; io2d may be a
    column block of a wider matrix (include/yat_hip.h: yat_rank_expand)."""
    _chk_bf16(h2d, w, io2d)
    rows, R = h2d.shape
    N = w.shape[1]
    if w.shape[0] != R or io2d.shape != (rows, N) or not (h2d.is_contiguous() and w.is_contiguous()) or io2d.stride(1) != 1:
        raise ValueError("rank_expand: shape mismatch")
    _l.check(_lib().yat_rank_expand(rows, N, R, _p(w), _p(h2d), _p(io2d), io2d.stride(0), float(scale), int(residual), _stream()),
             "yat_rank_expand")
    return io2d


_SW_WS = {}


def lokr_small_wgrad(a2d, x2d, out2d, workspace=None, accumulate=False, scale=1.0):
    """out[q, n] (+)= bf16(scale * sum_row a[row, q] * x[row, n]), q < out.shape[0]; x2d / out2d may be column blocks of wider
    matrices (row strides) -- include/yat_hip.h: yat_lokr_small_wgrad.  workspace None: a cached one per (device, stream)."""
    _chk_bf16(a2d, x2d, out2d)
    rows, R = a2d.shape
    N = x2d.shape[1]
    if x2d.shape[0] != rows or out2d.shape[1] != N or not a2d.is_contiguous() or x2d.stride(1) != 1 or out2d.stride(1) != 1:
        raise ValueError("lokr_small_wgrad: shape mismatch")
    need = int(_lib().yat_lokr_small_wgrad_workspace_bytes(rows, R, N))
    if workspace is None:
        key = (a2d.device, torch.cuda.current_stream().cuda_stream)
        workspace = _SW_WS.get(key)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(max(need, 8 << 20), dtype=torch.uint8, device=a2d.device)
            _SW_WS[key] = workspace
    elif workspace.numel() < need:
        raise ValueError("lokr_small_wgrad: workspace too small")
    _l.check(_lib().yat_lokr_small_wgrad(rows, R, N, out2d.shape[0], _p(a2d), _p(x2d), x2d.stride(0), _p(out2d),
                                         out2d.stride(0), float(scale), int(accumulate), _p(workspace), _stream()),
             "yat_lokr_small_wgrad")
    return out2d


def colsum(x2d, out, workspace, accumulate=False):
    """yat_colsum_bf16: out[c] (+)= sum_r x[r,c]."""
    rows, cols = x2d.shape
    rc = _lib().yat_colsum_bf16(rows, cols, _p(x2d), x2d.stride(0), _p(out), int(accumulate), _p(workspace), _stream())
    _dyn_rows(0, rows)
    _l.check(rc, "yat_colsum_bf16")
    return out


# ----------------------------------------------------------------------------------------------- norms / modulation
def modulation_fwd(table, tmod, slot_stride, out=None):
    S, D = table.shape
    B = tmod.shape[0]
    out = out if out is not None else torch.empty(B, S, D, dtype=BF16, device=table.device)
    rc = _lib().yat_modulation_fwd(B, S, D, _p(table), _p(tmod), tmod.stride(0), slot_stride, _p(out), _stream())
    _l.check(rc, "yat_modulation_fwd")
    return out


def modulation_bwd(dmod_f32, dtable, dtmod_acc_f32, slot_stride, accumulate_table=False):
    B, S, D = dmod_f32.shape
    rc = _lib().yat_modulation_bwd(B, S, D, _p(dmod_f32), _p(dtable), int(accumulate_table), _p(dtmod_acc_f32),
                                   dtmod_acc_f32.stride(0), slot_stride, _stream())
    _l.check(rc, "yat_modulation_bwd")


def ln_modulate_fwd(x2d, shift, scale, mod_ld, rows_per_batch, eps, y=None, mean=None, rstd=None):
    M, D = x2d.shape
    dev = x2d.device
    y = y if y is not None else torch.empty_like(x2d)
    mean = mean if mean is not None else torch.empty(M, dtype=torch.float32, device=dev)
    rstd = rstd if rstd is not None else torch.empty(M, dtype=torch.float32, device=dev)
    rc = _lib().yat_ln_modulate_fwd(M, D, rows_per_batch, eps, _p(x2d), _p(shift), _p(scale), mod_ld, _p(y), _p(mean),
                                    _p(rstd), _stream())
    _l.check(rc, "yat_ln_modulate_fwd")
    return y, mean, rstd


def ln_bwd_workspace_bytes(M, D, rpb):
    return int(_lib().yat_ln_bwd_workspace_bytes(M, D, rpb))


def ln_modulate_bwd(x2d, mean, rstd, scale, mod_ld, rows_per_batch, dy, dres, dx, dshift_acc, dscale_acc, acc_ld,
                    workspace, parts=3):
    """parts: 1 = dx only, 2 = dshift/dscale accumulators only, 3 = both (see include/yat_hip.h)."""
    M, D = x2d.shape
    rc = _lib().yat_ln_modulate_bwd(M, D, rows_per_batch, _p(x2d), _p(mean), _p(rstd), _p(scale), mod_ld, _p(dy),
                                    _p(dres), _p(dx), _p(dshift_acc), _p(dscale_acc), acc_ld, _p(workspace), parts,
                                    _stream())
    _l.check(rc, "yat_ln_modulate_bwd")
    return dx


def rmsnorm_fwd(x2d, w, eps, y=None, rstd=None):
    M, D = x2d.shape
    y = y if y is not None else torch.empty_like(x2d)
    rstd = rstd if rstd is not None else torch.empty(M, dtype=torch.float32, device=x2d.device)
    rc = _lib().yat_rmsnorm_fwd(M, D, eps, _p(x2d), _p(w), _p(y), _p(rstd), _stream())
    _dyn_rows(0, M)
    _l.check(rc, "yat_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(x2d, w, rstd, dy, dx, dw, workspace, accumulate_dw=False):
    M, D = x2d.shape
    rc = _lib().yat_rmsnorm_bwd(M, D, _p(x2d), _p(w), _p(rstd), _p(dy), _p(dx), _p(dw), int(accumulate_dw),
                                _p(workspace), _stream())
    _dyn_rows(0, M)
    _l.check(rc, "yat_rmsnorm_bwd")


def gate_bwd(dout, lin, gate, gate_ld, rows_per_batch, dlin, dgate_acc, acc_ld, workspace, dbias=None,
             accumulate_bias=False):
    """dlin = gate * dout, dgate += sum dout * lin; ``dbias`` (optional) (+)= column sum of dlin in the same pass."""
    M, D = dout.shape
    rc = _lib().yat_gate_bwd(M, D, rows_per_batch, _p(dout), _p(lin), _p(gate), gate_ld, _p(dlin), _p(dgate_acc), acc_ld,
                             _p(dbias), int(accumulate_bias), _p(workspace), _stream())
    _l.check(rc, "yat_gate_bwd")


# ----------------------------------------------------------------------------------------------- attention
def linear_attn_workspace_bytes(B, N, H):
    return int(_lib().yat_linear_attn_workspace_bytes(B, N, H))


def linear_attn_fwd(qkv2d, B, N, H, k_off, v_off, out, workspace):
    rc = _lib().yat_linear_attn_fwd(B, N, H, _p(qkv2d), qkv2d.stride(0), k_off, v_off, _p(out), out.stride(0),
                                    _p(workspace), _stream())
    _l.check(rc, "yat_linear_attn_fwd")
    return out


def linear_attn_bwd(qkv2d, B, N, H, k_off, v_off, dout, dqkv, workspace, state=None):
    rc = _lib().yat_linear_attn_bwd(B, N, H, _p(qkv2d), qkv2d.stride(0), k_off, v_off, _p(dout), dout.stride(0),
                                    _p(dqkv), dqkv.stride(0), _p(state), _p(workspace), _stream())
    _l.check(rc, "yat_linear_attn_bwd")
    return dqkv


def sdpa_fwd(q2d, k2d, v2d, B, N, T, H, dh, scale, key_bias, kv_len, out, lse, kv_off=None):
    """k2d / v2d may be column slices of one fused [B*T, 2*H*dh] projection (same row stride).  ``kv_off`` (device int32, one
    row offset per image): packed keys -- k2d / v2d are the whole [rows, .] matrices without padding rows
    (include/yat_hip.h: yat_sdpa_fwd_packed).  ``key_bias`` None (then ``kv_len`` None too): plain attention over all T
    keys -- self-attention; same result as a zero bias, on the kernels' no-bias instantiations."""
    assert k2d.stride(0) == v2d.stride(0)
    if key_bias is None and (kv_len is not None or kv_off is not None):
        raise ValueError("sdpa: no key bias means every key of every image attends (no kv_len, no packed keys)")
    if kv_off is not None:
        rc = _lib().yat_sdpa_fwd_packed(B, N, T, H, dh, scale, _p(q2d), q2d.stride(0), _p(k2d), _p(v2d), k2d.stride(0),
                                        _p(kv_off), k2d.shape[0], _p(key_bias), _p(kv_len), _p(out), out.stride(0), _p(lse),
                                        _stream())
        _dyn_rows(12, k2d.shape[0])
        _l.check(rc, "yat_sdpa_fwd_packed")
        return out
    rc = _lib().yat_sdpa_fwd(B, N, T, H, dh, scale, _p(q2d), q2d.stride(0), _p(k2d), _p(v2d), k2d.stride(0),
                             _p(key_bias), _p(kv_len), _p(out), out.stride(0), _p(lse), _stream())
    _l.check(rc, "yat_sdpa_fwd")
    return out


def kv_work_list(lens, T, device):
    """Compact (batch, key tile) list for yat_sdpa_bwd from HOST-known key lengths (0 = all T keys attend)."""
    pairs = [(b, t) for b, L in enumerate(lens) for t in range(((L if L > 0 else T) + 63) // 64)]
    return torch.tensor(pairs, dtype=torch.int32).to(device, non_blocking=True)


def sdpa_bwd(q2d, k2d, v2d, B, N, T, H, dh, scale, key_bias, kv_len, out, dout, lse, delta, dq, dk, dv, work=None,
             parts=3, kv_off=None):
    """parts: 1 = dQ + delta, 2 = dK/dV (after part 1, possibly on another stream), 3 = both.  ``kv_off``: packed keys, as in
    ``sdpa_fwd`` (dk / dv share the packed row layout; rows outside the images' ranges are left alone).  ``key_bias`` None:
    as in ``sdpa_fwd``."""
    assert k2d.stride(0) == v2d.stride(0) and dk.stride(0) == dv.stride(0)
    if key_bias is None and (kv_len is not None or kv_off is not None or work is not None):
        raise ValueError("sdpa: no key bias means every key of every image attends (no kv_len, no work list, no packed keys)")
    if kv_off is not None:
        rc = _lib().yat_sdpa_bwd_packed(B, N, T, H, dh, scale, _p(q2d), q2d.stride(0), _p(k2d), _p(v2d), k2d.stride(0),
                                        _p(kv_off), k2d.shape[0], _p(key_bias), _p(kv_len), _p(out), out.stride(0), _p(dout),
                                        dout.stride(0), _p(lse), _p(delta), _p(dq), dq.stride(0), _p(dk), _p(dv), dk.stride(0),
                                        _p(work), 0 if work is None else work.shape[0], parts, _stream())
        if RECORDER is not None and work is not None:
            RECORDER.mark_dynamic("n_work", 27)
        _dyn_rows(12, k2d.shape[0])
        _l.check(rc, "yat_sdpa_bwd_packed")
        return
    rc = _lib().yat_sdpa_bwd(B, N, T, H, dh, scale, _p(q2d), q2d.stride(0), _p(k2d), _p(v2d), k2d.stride(0),
                             _p(key_bias), _p(kv_len), _p(out), out.stride(0), _p(dout), dout.stride(0), _p(lse),
                             _p(delta), _p(dq), dq.stride(0), _p(dk), _p(dv), dk.stride(0), _p(work),
                             0 if work is None else work.shape[0], parts, _stream())
    if RECORDER is not None and work is not None:
        RECORDER.mark_dynamic("n_work", 25)              # the work list keeps its buffer; its length is per batch
    _l.check(rc, "yat_sdpa_bwd")


# ----------------------------------------------------------------------------------------------- MMDiT joint attention glue
def qknorm_concat_fwd(qkv_img, qkv_txt, B, N, T, H, dh, eps, wq_img, wk_img, wq_txt, wk_txt, joint, rstd):
    """joint [B*(N+T), 3D] <- [RMSNorm_head(q) | RMSNorm_head(k) | v] of the image rows, then of the text rows, per image;
    qkv_txt None (T = 0): self-attention over the image tokens only (include/yat_hip.h: yat_qknorm_concat_fwd)."""
    _chk_bf16(qkv_img, qkv_txt, wq_img, wk_img, wq_txt, wk_txt, joint)
    rc = _lib().yat_qknorm_concat_fwd(B, N, T, H, dh, eps, _p(qkv_img), qkv_img.stride(0), _p(qkv_txt),
                                      0 if qkv_txt is None else qkv_txt.stride(0), _p(wq_img), _p(wk_img), _p(wq_txt),
                                      _p(wk_txt), _p(joint), joint.stride(0), _p(rstd), _stream())
    _l.check(rc, "yat_qknorm_concat_fwd")
    return joint


def qknorm_concat_bwd_workspace_bytes(B, N, T, dh):
    return int(_lib().yat_qknorm_concat_bwd_workspace_bytes(B, N, T, dh))


def qknorm_concat_bwd(qkv_img, qkv_txt, B, N, T, H, dh, wq_img, wk_img, wq_txt, wk_txt, rstd, d_joint, dqkv_img, dqkv_txt,
                      dwq_img, dwk_img, dwq_txt, dwk_txt, workspace, accumulate_dw=False):
    _chk_bf16(qkv_img, qkv_txt, d_joint, dqkv_img, dqkv_txt, dwq_img, dwk_img, dwq_txt, dwk_txt)
    rc = _lib().yat_qknorm_concat_bwd(B, N, T, H, dh, _p(qkv_img), qkv_img.stride(0), _p(qkv_txt),
                                      0 if qkv_txt is None else qkv_txt.stride(0), _p(wq_img), _p(wk_img), _p(wq_txt),
                                      _p(wk_txt), _p(rstd), _p(d_joint), d_joint.stride(0), _p(dqkv_img), dqkv_img.stride(0),
                                      _p(dqkv_txt), 0 if dqkv_txt is None else dqkv_txt.stride(0), _p(dwq_img), _p(dwk_img),
                                      _p(dwq_txt), _p(dwk_txt), int(accumulate_dw), _p(workspace), _stream())
    _l.check(rc, "yat_qknorm_concat_bwd")


def joint_rows(joint, img, txt, B, N, T, to_joint):
    """Rows between the joint layout [B*(N+T), C] and the per-stream layouts [B*N, C], [B*T, C] (txt may be None)."""
    _chk_bf16(joint, img, txt)
    C_ = img.shape[1]
    rc = _lib().yat_joint_rows(B, N, T, C_, _p(joint), joint.stride(0), _p(img), img.stride(0), _p(txt),
                               0 if txt is None else txt.stride(0), int(to_joint), _stream())
    _l.check(rc, "yat_joint_rows")


# ----------------------------------------------------------------------------------------------- GLUMBConv middle
def dwconv_glu_fwd(s, B, h, w, Hc, wdw, bdw, y, u_out=None):
    """s = bf16(SiLU(conv_inverted output)) as written by the GEMM epilogue; ``u_out`` (optional [M, 2Hc]) keeps the conv
    output for the backward (see ``linear_dgrad_glu`` / ``dwconv_glu_bwd(du=...)``)."""
    rc = _lib().yat_dwconv_glu_fwd(B, h, w, Hc, _p(s), _p(wdw), _p(bdw), _p(y), _p(u_out), _stream())
    _l.check(rc, "yat_dwconv_glu_fwd")
    return y


def dwconv_glu_bwd_workspace_bytes(B, h, w, Hc):
    return int(_lib().yat_dwconv_glu_bwd_workspace_bytes(B, h, w, Hc))


def dwconv_glu_bwd(s, z, B, h, w, Hc, wdw, bdw, dy, dz, dwdw, dbdw, workspace, accumulate=False, dz_colsum=None, du=None):
    """``dz_colsum`` (optional, [2Hc]) (+)= column sum of dz (the conv_inverted bias gradient) in the same pass.
    ``du`` (optional [M, 2Hc]): the GLU backward already applied (``linear_dgrad_glu``); ``dy`` may then be None."""
    rc = _lib().yat_dwconv_glu_bwd(B, h, w, Hc, _p(s), _p(z), _p(wdw), _p(bdw), _p(dy), _p(dz), _p(dwdw), _p(dbdw),
                                   _p(dz_colsum), int(accumulate), _p(workspace), _p(du), _stream())
    _l.check(rc, "yat_dwconv_glu_bwd")


# ----------------------------------------------------------------------------------------------- elementwise / recipe
def act_fwd(x, act, y=None):
    y = y if y is not None else torch.empty_like(x)
    _l.check(_lib().yat_act_fwd(x.numel(), ACT[act], _p(x), _p(y), _stream()), "yat_act_fwd")
    return y


def act_bwd(x, dy, act, dx=None):
    dx = dx if dx is not None else torch.empty_like(x)
    rc = _lib().yat_act_bwd(x.numel(), ACT[act], _p(x), _p(dy), _p(dx), _stream())
    if TEXT_ROWS is not None:
        _dyn_rows(0, x.numel(), mul=x.numel() // x.shape[0])
    _l.check(rc, "yat_act_bwd")
    return dx


def zero_(t):
    """t.zero_() as a C-ABI call (yat_memset_zero): recordable in a launch plan."""
    if not t.is_contiguous():
        raise ValueError("zero_: contiguous tensors only")
    _l.check(_lib().yat_memset_zero(_p(t), t.numel() * t.element_size(), _stream()), "yat_memset_zero")
    return t


def zero_last_rows(x2d, n):
    """Zero the last ``n`` rows of a contiguous [rows, cols] matrix (inside a ``text_rows`` scope the address follows the row
    count: a launch plan replays it for another number of rows)."""
    rows = x2d.shape[0]
    if not x2d.is_contiguous() or rows < n:
        raise ValueError("zero_last_rows: contiguous matrix with at least n rows")
    row_bytes = x2d.shape[1] * x2d.element_size()
    ptr = x2d.data_ptr() + (rows - n) * row_bytes
    rc = _lib().yat_memset_zero(C.c_void_p(ptr), n * row_bytes, _stream())
    _dyn_rows(0, ptr, mul=row_bytes, add=x2d.data_ptr() - n * row_bytes)
    _l.check(rc, "yat_memset_zero")


def add_bf16(a, b, out=None):
    out = out if out is not None else torch.empty_like(a)
    _l.check(_lib().yat_add_bf16(a.numel(), _p(a), _p(b), _p(out), _stream()), "yat_add_bf16")
    return out


def f32_to_bf16(x, y=None):
    y = y if y is not None else torch.empty(x.shape, dtype=BF16, device=x.device)
    _l.check(_lib().yat_f32_to_bf16(x.numel(), _p(x), _p(y), _stream()), "yat_f32_to_bf16")
    return y


def transpose(x3d, out=None):
    """yat_transpose_bf16: [B, R, C] -> [B, C, R]."""
    B, R, Cc = x3d.shape
    out = out if out is not None else torch.empty(B, Cc, R, dtype=BF16, device=x3d.device)
    _l.check(_lib().yat_transpose_bf16(B, R, Cc, _p(x3d), _p(out), _stream()), "yat_transpose_bf16")
    return out


def timestep_embed(t_f32, dim=256, out=None):
    B = t_f32.numel()
    out = out if out is not None else torch.empty(B, dim, dtype=BF16, device=t_f32.device)
    _l.check(_lib().yat_timestep_embed_fwd(B, dim, _p(t_f32), _p(out), _stream()), "yat_timestep_embed_fwd")
    return out


def pad_mask(src_cat, offsets_i32, B, T, Cdim, dst, mask_i64, key_bias_f32, kv_len_i32):
    rc = _lib().yat_pad_mask(B, T, Cdim, _p(src_cat), _p(offsets_i32), _p(dst), _p(mask_i64), _p(key_bias_f32),
                             _p(kv_len_i32), _stream())
    _l.check(rc, "yat_pad_mask")


def pack_mask(src_cat, offsets_i32, B, T, Cdim, dst_packed, mask_i64, key_bias_f32, kv_len_i32):
    """The text rows WITHOUT padding rows: dst_packed [rows_padded, C] = the source rows, then zero rows; mask / key bias /
    kv_len as ``pad_mask`` writes them (include/yat_hip.h: yat_pack_mask)."""
    rc = _lib().yat_pack_mask(B, T, Cdim, dst_packed.shape[0], _p(src_cat), _p(offsets_i32), _p(dst_packed), _p(mask_i64),
                              _p(key_bias_f32), _p(kv_len_i32), _stream())
    _l.check(rc, "yat_pack_mask")


def flow_mix(x, noise, sigma_bf16, noisy=None, target=None):
    B = x.shape[0]
    per = x.numel() // B
    noisy = noisy if noisy is not None else torch.empty_like(x)
    target = target if target is not None else torch.empty_like(x)
    rc = _lib().yat_flow_mix(B, per, _p(x), _p(noise), _p(sigma_bf16), _p(noisy), _p(target), _stream())
    _l.check(rc, "yat_flow_mix")
    return noisy, target


def mse_fwd_bwd(pred, target, loss_f32, dpred, workspace_f32, gscale=1.0):
    rc = _lib().yat_mse_fwd_bwd(pred.numel(), _p(pred), _p(target), gscale, _p(loss_f32), _p(dpred), _p(workspace_f32),
                                _stream())
    _l.check(rc, "yat_mse_fwd_bwd")
    return loss_f32


def patch_rearrange(src, dst, B, C, H, W, p, channel_major, to_tokens):
    """NCHW <-> rows of p x p patches (include/yat_hip.h: yat_patch_rearrange)."""
    _chk_bf16(src, dst)
    if src.numel() != B * C * H * W or dst.numel() != src.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("patch_rearrange: shape mismatch")
    _l.check(_lib().yat_patch_rearrange(B, C, H, W, p, int(channel_major), int(to_tokens), _p(src), _p(dst), _stream()),
           "patch_rearrange")
    return dst


def add_pos_embed(x2d, pos_f32, out=None):
    _chk_bf16(x2d)
    out = x2d if out is None else out
    N, D = pos_f32.shape
    if pos_f32.dtype != torch.float32 or x2d.shape[1] != D or x2d.shape[0] % N or not pos_f32.is_contiguous() \
            or not x2d.is_contiguous() or not out.is_contiguous():
        raise ValueError("add_pos_embed: shape mismatch")
    _l.check(_lib().yat_add_pos_embed(x2d.shape[0], N, D, _p(x2d), _p(pos_f32), _p(out), _stream()), "add_pos_embed")
    return out


def ddpm_add_noise(x, noise, sqrt_alpha_prod_bf16, sqrt_one_minus_bf16, noisy=None):
    _chk_bf16(x, noise, sqrt_alpha_prod_bf16, sqrt_one_minus_bf16)
    B = x.shape[0]
    if noise.shape != x.shape or sqrt_alpha_prod_bf16.numel() != B or sqrt_one_minus_bf16.numel() != B:
        raise ValueError("ddpm_add_noise: shape mismatch")
    noisy = torch.empty_like(x) if noisy is None else noisy
    _l.check(_lib().yat_ddpm_add_noise(B, x.numel() // B, _p(x), _p(noise), _p(sqrt_alpha_prod_bf16), _p(sqrt_one_minus_bf16),
                                     _p(noisy), _stream()), "ddpm_add_noise")
    return noisy


def mse_bf16_chunk(pred, target, loss_f32, dpred, workspace_f32, gscale=1.0):
    """pred [B, C2, H, W] of which channels [: target.shape[1]] are compared with target [B, C, H, W] (bf16 MSELoss)."""
    _chk_bf16(pred, target)
    B = pred.shape[0]
    stride, used = pred.numel() // B, target.numel() // B
    if target.shape[0] != B or used > stride or pred.shape[2:] != target.shape[2:] or workspace_f32.numel() < 256 \
            or not (pred.is_contiguous() and target.is_contiguous()) or (dpred is not None and dpred.shape != pred.shape):
        raise ValueError("mse_bf16_chunk: shape mismatch")
    _l.check(_lib().yat_mse_bf16_chunk(B, used, stride, _p(pred), _p(target), float(gscale), _p(loss_f32),
                                     _p(dpred) if dpred is not None else None, _p(workspace_f32), _stream()), "mse_bf16_chunk")
    return loss_f32


# ----------------------------------------------------------------------------------------------- optimizer
def gradnorm_workspace_bytes(n, nseg):
    return int(_lib().yat_gradnorm_workspace_bytes(n, nseg))


def gradnorm_clip(grad_flat, seg_start_i64, max_norm, norm_out, clip_coef, workspace):
    nseg = seg_start_i64.numel() - 1
    rc = _lib().yat_gradnorm_clip(grad_flat.numel(), _p(grad_flat), nseg, _p(seg_start_i64), max_norm, _p(norm_out),
                                  _p(clip_coef), _p(workspace), _stream())
    _l.check(rc, "yat_gradnorm_clip")


def gradnorm_pieces_partial(grad_flat, piece_start_i64, chunk_base_i32, max_piece_chunks, owned_u8, partial_f32):
    """Sums of squares per (piece, 2^18-element chunk) into ``partial_f32[chunk_base[p] + c]``; pieces with ``owned_u8[p] == 0``
    get zeros (``owned_u8`` None: all of them).  See include/yat_hip.h on partition invariance."""
    rc = _lib().yat_gradnorm_pieces_partial(_p(grad_flat), piece_start_i64.numel() - 1, _p(piece_start_i64), _p(chunk_base_i32),
                                            int(max_piece_chunks), _p(owned_u8), _p(partial_f32), _stream())
    _l.check(rc, "yat_gradnorm_pieces_partial")


def gradnorm_pieces_finish(tensor_first_piece_i32, chunk_base_i32, partial_f32, max_norm, norm_out, clip_coef):
    rc = _lib().yat_gradnorm_pieces_finish(tensor_first_piece_i32.numel() - 1, _p(tensor_first_piece_i32), _p(chunk_base_i32),
                                           _p(partial_f32), float(max_norm), _p(norm_out), _p(clip_coef), _stream())
    _l.check(rc, "yat_gradnorm_pieces_finish")


def adamw_step(param, grad, exp_avg, exp_avg_sq, clip_coef, lr, beta1, beta2, eps, weight_decay, step, zero_grad=True,
               ema_shadow=None, ema_decay=0.0, background=0):
    rc = _lib().yat_adamw_step(param.numel(), _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(clip_coef), lr, beta1,
                               beta2, eps, weight_decay, step, int(zero_grad), _p(ema_shadow), ema_decay, int(background),
                               _stream())
    _l.check(rc, "yat_adamw_step")
