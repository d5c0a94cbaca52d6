"""LoKr adapters on the HIP path (BASELINE config 5: ``lora_algo: lokr``; the reference wraps the transformer with peft's
``LoKrConfig(r, alpha, module_dropout, target_modules)`` at common/trainer.py:212-238 and hands every parameter to AdamW).

Arithmetic [RECALL peft/tuners/lokr/layer.py -- parity unpinned, see oracle/lokr_ref.py for the restatement]: for a target
weight W [out, in], ``result = base(x) + F.linear(x, kron(w1, w2_a @ w2_b) * alpha / r)`` in bf16, every op rounding.

MI355X mapping: the adapter set owns one flat bf16 buffer (w1 | w2_a | w2_b per target, 8-element aligned) with a flat
gradient twin -- so clip + AdamW and the data-parallel all-reduce are the same single-launch machinery as for the full
model, over ~0.6 M parameters instead of 1.6 B -- and a ``delta`` buffer laid out exactly like the model's flat weights:
target t's dense delta_w sits at W_t's offset, so the fused [3D, D] QKV view of the base weights has a matching fused view
of the deltas.  Per step: ``materialize()`` rebuilds every delta_w (one small launch per target, or zeros when peft's
module dropout drops the adapter for this step), the forward adds ``x delta_w^T`` through the GEMM's ``pre_add`` epilogue,
the backward adds ``dy delta_w`` to every input gradient, the ordinary weight-gradient GEMMs now produce d_delta_w in the
model's flat gradient buffer (the base weights are frozen: nobody reads those slots as weight gradients), and
``project()`` folds each d_delta_w into (d_w1, d_w2_a, d_w2_b).
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import ops

BF16 = torch.bfloat16


def factorization(dimension: int, factor: int = -1):
    """[RECALL peft lycoris_utils.factorization] divisor pair (m <= n) of ``dimension`` with the smallest m + n."""
    if factor > 0 and dimension % factor == 0:
        return factor, dimension // factor
    if factor == -1:
        factor = dimension
    m, n = 1, dimension
    length = m + n
    while m < n:
        new_m = m + 1
        while dimension % new_m != 0:
            new_m += 1
        new_n = dimension // new_m
        if new_m + new_n > length or new_m > factor:
            break
        m, n = new_m, new_n
    if m > n:
        m, n = n, m
    return m, n


def is_target(module_name: str, targets) -> bool:
    return any(module_name == t or module_name.endswith("." + t) for t in targets)


class LoKrAdapters:
    def __init__(self, model, targets, r: int, alpha: float, module_dropout: float = 0.0):
        self.model, self.r, self.alpha, self.scale = model, int(r), float(alpha), float(alpha) / int(r)
        self.targets, self.module_dropout = list(targets), float(module_dropout)
        dev = model.flat_param.device
        self.entries, off, segs = [], 0, [0]

        def take(n):
            nonlocal off
            o = off
            off += (n + 7) & ~7
            segs.append(o + n)              # segment = the tensor itself (the pad belongs to nobody)
            segs.append(off)
            return o
        base_ptr = model.flat_param.data_ptr()
        for key, w in model.P.items():
            if not key.endswith(".weight") or w.dim() < 2 or not is_target(key[:-7], self.targets):
                continue
            out_dim, in_dim = w.shape[0], w.numel() // w.shape[0]
            (out_l, out_k), (in_m, in_n) = factorization(out_dim), factorization(in_dim)
            if not (self.r < max(out_k, in_n) / 2):
                raise NotImplementedError(f"{key}: full lokr_w2 (r >= max(out_k, in_n)/2) is not built")
            e = dict(module=key[:-7], key=key, out=out_dim, inn=in_dim, out_l=out_l, out_k=out_k, in_m=in_m, in_n=in_n,
                     w_off=(w.data_ptr() - base_ptr) // 2, o1=take(out_l * in_m), oa=take(out_k * self.r),
                     ob=take(self.r * in_n), active=True)
            self.entries.append(e)
        if not self.entries:
            raise ValueError("no module matches lora_target_modules")
        self.numel_flat = off
        self.flat_param = torch.zeros(off, dtype=BF16, device=dev)
        self.flat_grad = torch.zeros(off, dtype=BF16, device=dev)
        # zero-length segments are fine for the norm kernel; keep them strictly increasing by dropping duplicates
        self.seg_start = torch.tensor(sorted(set(segs)), dtype=torch.int64)
        self.bucket_bounds = [(0, off)]
        self.param_events = None
        self.grad_ready = None              # HipDDP hook: called once, after project()
        self.delta = torch.zeros_like(model.flat_param)
        ws = max(ops._lib().yat_lokr_project_workspace_bytes(e["out_l"], e["out_k"], e["in_n"]) for e in self.entries)
        self._ws = torch.empty(int(ws), dtype=torch.uint8, device=dev)
        self.reset_parameters()
        model.adapters = self

    # ---- views
    def _views(self, e, flat):
        return (flat[e["o1"]:e["o1"] + e["out_l"] * e["in_m"]].view(e["out_l"], e["in_m"]),
                flat[e["oa"]:e["oa"] + e["out_k"] * self.r].view(e["out_k"], self.r),
                flat[e["ob"]:e["ob"] + self.r * e["in_n"]].view(self.r, e["in_n"]))

    def delta_like(self, w):
        """The view of the delta buffer that mirrors weight view ``w`` (same offset, shape and strides)."""
        off = (w.data_ptr() - self.model.flat_param.data_ptr()) // 2
        return torch.as_strided(self.delta, w.size(), w.stride(), off)

    def reset_parameters(self):
        """peft init_weights=True: w1 zeros, w2_a / w2_b kaiming_uniform(a=sqrt(5)) drawn on the CPU, then cast."""
        for e in self.entries:
            w1, wa, wb = self._views(e, self.flat_param)
            w1.zero_()
            for t in (wa, wb):
                init = torch.empty(t.shape, dtype=torch.float32)
                torch.nn.init.kaiming_uniform_(init, a=math.sqrt(5))
                t.copy_(init.to(BF16))

    def join_pending_update(self):
        pev, self.param_events = self.param_events, None
        if pev is not None:
            cur = torch.cuda.current_stream()
            for ev in pev:
                cur.wait_event(ev)

    # ---- per step
    def materialize(self, training=True):
        """Rebuild every delta_w from the current adapter parameters (zeros where module dropout drops the adapter)."""
        self.join_pending_update()
        for e in self.entries:
            e["active"] = (not training) or self.module_dropout <= 0.0 or bool(torch.rand(1) > self.module_dropout)
            d2 = self.delta[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
            if e["active"]:
                w1, wa, wb = self._views(e, self.flat_param)
                ops.lokr_delta(w1, wa, wb, self.scale, d2)
            else:
                d2.zero_()

    def project(self):
        """d_delta_w (the model's flat gradient slots of the frozen target weights) -> adapter gradients."""
        G = self.model.flat_grad
        for e in self.entries:
            g1, ga, gb = self._views(e, self.flat_grad)
            if not e["active"]:
                g1.zero_(); ga.zero_(); gb.zero_()
                continue
            w1, wa, wb = self._views(e, self.flat_param)
            dd = G[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
            ops.lokr_project(w1, wa, wb, self.scale, dd, g1, ga, gb, self._ws)
        if self.grad_ready is not None:
            self.grad_ready(0)

    # ---- checkpoint (peft layout: adapter_model.safetensors + adapter_config.json)
    def state_dict(self):
        self.join_pending_update()
        sd = {}
        for e in self.entries:
            w1, wa, wb = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            sd[pre + "lokr_w1"], sd[pre + "lokr_w2_a"], sd[pre + "lokr_w2_b"] = w1, wa, wb
        return sd

    def load_state_dict(self, sd):
        for e in self.entries:
            pre = f"base_model.model.{e['module']}."
            for t, name in zip(self._views(e, self.flat_param), ("lokr_w1", "lokr_w2_a", "lokr_w2_b")):
                t.copy_(sd[pre + name].to(device=t.device, dtype=BF16))

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "adapter_model.safetensors"))
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LOKR", "r": self.r, "alpha": self.alpha, "module_dropout": self.module_dropout,
                       "target_modules": self.targets, "decompose_both": False, "decompose_factor": -1,
                       "init_weights": True, "rank_dropout": 0.0, "use_effective_conv2d": False}, f, indent=2)

    def num_parameters(self):
        return sum(e["out_l"] * e["in_m"] + e["out_k"] * self.r + self.r * e["in_n"] for e in self.entries)
