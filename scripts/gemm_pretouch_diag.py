#!/usr/bin/env python3
"""What would the step's GEMMs cost if their operands were already in the Infinity Cache?  Diagnostic, not a product path.

Runs bench.py's serialized roofline pass with every GEMM launch preceded (same stream, outside the GEMM's HIP events) by a
read pass over the operands named in PRETOUCH:  a | b | ab | none | dup (the launch itself run twice, the second one timed) | dummy (an unrelated GEMM first) | abdummy.  The per-shape table (--gemm-detail) then shows, for the
launches of the real step in their real order, the in-step time with operand A / B / both brought back into the 256 MB
Infinity Cache right before the launch -- the prize a deeper DMA lookahead or an operand warmer can win at most.

    PRETOUCH=ab python scripts/gemm_pretouch_diag.py --steps 4 --warmup 3 --no-cpu-baseline --gemm-detail out.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import ops

MODE = os.environ.get("PRETOUCH", "ab")
_gemm = ops.gemm
_sink = []


def _touch(t, rows, cols, ld):
    """read [rows, cols] of a row-major operand with leading dimension ld (one pass; result discarded)"""
    v = torch.as_strided(t, (rows, cols), (ld, 1), storage_offset=t.storage_offset()) if t.dim() != 2 or t.stride(0) != ld \
        else t[:rows, :cols]
    _sink.append(v.view(torch.int16).sum(dtype=torch.int32) if v.is_contiguous() else v.sum(dtype=torch.float32))
    if len(_sink) > 64:
        del _sink[:]


_scratch = {}


def _dummy():
    if not _scratch:
        g = torch.Generator(device="cuda").manual_seed(5)
        _scratch["a"] = (torch.randn(8192, 2240, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        _scratch["b"] = (torch.randn(4480, 2240, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        _scratch["c"] = torch.empty(8192, 4480, dtype=torch.bfloat16, device="cuda")
    timer, ops.GEMM_TIMER = ops.GEMM_TIMER, None
    _gemm(_scratch["a"], _scratch["b"], _scratch["c"], M=8192, N=4480, K=2240)
    ops.GEMM_TIMER = timer


def gemm(a, b, out, *, a_t=False, b_t=False, M, N, K, lda=None, ldb=None, **kw):
    if ops.GEMM_TIMER is not None and MODE == "dup":
        # the launch itself, untimed, right before the timed one: operands, output, TLBs and code all as hot as they get
        res = kw.get("residual")
        if not kw.get("a_rowsum_accumulate") and (res is None or res.data_ptr() != out.data_ptr()):
            timer, ops.GEMM_TIMER = ops.GEMM_TIMER, None
            _gemm(a, b, out, a_t=a_t, b_t=b_t, M=M, N=N, K=K, lda=lda, ldb=ldb, **kw)
            ops.GEMM_TIMER = timer
    elif ops.GEMM_TIMER is not None and MODE != "none":
        la = lda if lda is not None else (M if a_t else K)
        lb = ldb if ldb is not None else (N if b_t else K)
        if "a" in MODE.replace("dummy", ""):
            _touch(a, K if a_t else M, M if a_t else K, la)
        if "b" in MODE.replace("dummy", ""):
            _touch(b, K if b_t else N, N if b_t else K, lb)
        if MODE.endswith("dummy"):
            _dummy()                      # an unrelated GEMM on scratch operands right before the timed one: the chip is in its
            #                               GEMM power state, none of the timed launch's data was touched by it
    return _gemm(a, b, out, a_t=a_t, b_t=b_t, M=M, N=N, K=K, lda=lda, ldb=ldb, **kw)


ops.gemm = gemm
import bench  # noqa: E402

bench.main()
