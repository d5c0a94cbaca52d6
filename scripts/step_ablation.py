#!/usr/bin/env python3
"""Sensitivity of the SANA step to each non-GEMM kernel family: bench.py with that family's launches SKIPPED -- WRONG results,
only ms/step is read -- to see how much of the step each family's chip time is worth on this schedule (a kernel made x us
faster alone buys anything between 0 and x in the step: they run beside GEMMs of another stream).  A scripts/ monkeypatch of
yat_amd.ops on purpose: the product modules contain no switch that skips kernels (round-3 review, hygiene).

    ABLATE=dwfwd,la python scripts/step_ablation.py --steps 16 --warmup 4 --no-cpu-baseline --no-gemm-timer

Families: ln (LN+modulate fwd/bwd), gate (gate backward), la (linear attention), sdpa, dwfwd, dwbwd, adamw (clip + AdamW)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yat_amd import flat, ops

ABLATE = frozenset(x for x in os.environ.get("ABLATE", "").split(",") if x)
# family -> {wrapper name: what the skipped call returns (index of the positional / name of the keyword argument, or None)}
FAMILIES = {
    "ln": {"ln_modulate_fwd": "fwd3", "ln_modulate_bwd": "dx"},
    "gate": {"gate_bwd": None},
    "la": {"linear_attn_fwd": "out", "linear_attn_bwd": "dqkv"},
    "sdpa": {"sdpa_fwd": "out", "sdpa_bwd": None},
    "dwfwd": {"dwconv_glu_fwd": "y"},
    "dwbwd": {"dwconv_glu_bwd": None},
    "adamw": {"gradnorm_clip": None, "adamw_step": None},
}


def _skip(name, ret):
    import inspect
    real = getattr(ops, name)
    sig = inspect.signature(real)

    def skipped(*a, **k):
        if ret is None:
            return None
        b = sig.bind_partial(*a, **k)
        if ret == "fwd3":            # ln_modulate_fwd returns (y, mean, rstd): hand back whatever buffers the caller passed
            x = b.arguments["x2d"]
            y = b.arguments.get("y")
            M = x.shape[0]
            mk = lambda v: v if v is not None else torch.empty(M, dtype=torch.float32, device=x.device)
            return (y if y is not None else torch.empty_like(x)), mk(b.arguments.get("mean")), mk(b.arguments.get("rstd"))
        return b.arguments.get(ret)
    return skipped


unknown = ABLATE - set(FAMILIES)
if unknown:
    raise SystemExit(f"unknown families {sorted(unknown)}; known: {sorted(FAMILIES)}")
for fam in ABLATE:
    for name, ret in FAMILIES[fam].items():
        setattr(ops, name, _skip(name, ret))
if ABLATE:
    ops.DIAGNOSTIC_INVALID = f"ABLATE={','.join(sorted(ABLATE))}: kernels skipped, wrong results, timing diagnostic only"
    flat.ARENA_INIT = lambda t: t.normal_(0.0, 0.5) if t.is_floating_point() else None    # operand statistics without the kernels
    print(f"[step_ablation] {ops.DIAGNOSTIC_INVALID}", file=sys.stderr, flush=True)
import bench  # noqa: E402

bench.main()
