#!/usr/bin/env python3
"""Round 6: the B = 32 LoKr step in peft's rounding order (pair=False, the pre_add form) was not run-to-run deterministic
(tests/test_fulldepth_gpu.py::test_lokr_config5_batch32_step_properties[False]).  Which launch?  Every adapted Linear shape of a
SANA block at M = 32768 rows, the adapter term through forward_term() + the pre_add epilogue, repeated; outputs compared bit
for bit between repetitions, with and without a competing stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from yat_amd import ops
from yat_amd.lokr import LoKrAdapters, adapted_linear
BF, DEV = torch.bfloat16, "cuda"
M = int(os.environ.get("M", 32768))
shapes = {"qkv": (3, 2240, 2240), "to_out.0": (1, 2240, 2240), "conv_inverted": (1, 11200, 2240), "conv_point": (1, 2240, 5600)}
g = torch.Generator().manual_seed(1)
for name, (nblk, out, inn) in shapes.items():
    names = ["blk.to_q", "blk.to_k", "blk.to_v"] if nblk == 3 else ["blk." + name]
    flat = (torch.randn(nblk * out * inn, generator=g) * inn ** -0.5).to(BF).to(DEV)
    model = SimpleNamespace(P={n + ".weight": flat[i * out * inn:(i + 1) * out * inn].view(out, inn) for i, n in enumerate(names)},
                            flat_param=flat, flat_grad=torch.zeros_like(flat))
    for pair in (False, True):
        ad = LoKrAdapters(model, ["to_q", "to_k", "to_v", "to_out.0", "conv_inverted", "conv_point"], r=8, alpha=8.0, pair=pair)
        for e in ad.entries:
            w1 = ad._views(e, ad.flat_param)[0]
            w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
        x = torch.randn(M, inn, generator=g).to(BF).to(DEV)
        w = flat.view(nblk * out, inn)
        bias = torch.randn(nblk * out, generator=g).to(BF).to(DEV)
        outs, terms = [], []
        for rep in range(6):
            ad.materialize(True)
            y = adapted_linear(ad, x, w, bias, out=torch.empty(M, nblk * out, dtype=BF, device=DEV))
            if not pair:
                terms.append(ad.forward_term(x, w).clone())
            torch.cuda.synchronize()
            outs.append(y.clone())
        same = [torch.equal(outs[0], o) for o in outs[1:]]
        tsame = [torch.equal(terms[0], t) for t in terms[1:]] if terms else []
        print(f"{name:14s} pair={pair!s:5s} M={M}: outputs equal to the first run {same}; adapter term alone {tsame}; "
              f"finite {bool(torch.isfinite(outs[0].float()).all())}", flush=True)
        if not pair and not all(same):
            d = (outs[0].float() - outs[[i for i, s_ in enumerate(same) if not s_][0] + 1].float())
            rows = d.abs().amax(1).nonzero().flatten()
            cols = d.abs().amax(0).nonzero().flatten()
            print(f"   differing rows {rows.numel()} (first {rows[:6].tolist()}, last {rows[-3:].tolist()}), cols {cols.numel()} "
                  f"(first {cols[:6].tolist()}, last {cols[-3:].tolist()}), max |d| {d.abs().max().item():.3e}")
