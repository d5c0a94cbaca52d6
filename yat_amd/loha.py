"""LoHa adapters on the HIP path (``lora_algo: loha`` -- the reference wraps the transformer with peft's
``LoHaConfig(r, alpha, module_dropout, target_modules)`` at common/trainer.py:220-224 and hands every parameter to AdamW).

Arithmetic [RECALL peft/tuners/loha/layer.py -- parity unpinned, see oracle/loha_ref.py for the restatement]: for a target
Linear / 1x1 Conv with weight W [out, in] (``use_effective_conv2d=False``):
    delta_w = ((hada_w1_a @ hada_w1_b) * (hada_w2_a @ hada_w2_b)) * (alpha / r)        (HadaWeight.forward, bf16 op by op)
    result  = base_layer(x) + F.linear(x, delta_w);  the adapter is dropped for a call when rand(1) <= module_dropout
with w1_a, w1_b, w2_a kaiming-uniform(a=sqrt(5)) and w2_b zeros (``init_weights=True``: LoHaLayer.reset_adapter_parameters
zeroes hada_w2_b -- the factor that receives the first gradients is therefore w2_b), and HadaWeight's hand-written
backward: g = d_delta * scale; t1 = g * (w2a w2b); d_w1a = t1 w1b^T; d_w1b = w1a^T t1; t2 = g * (w1a w1b); d_w2a = t2 w2b^T;
d_w2b = w2a^T t2.

MI355X mapping -- the *dense* application of yat_amd/lokr.py with a different ``delta_w`` builder (a Hadamard product has no
factored shortcut: (A1 * A2) x is not a chain of skinny products): per step ``materialize()`` builds A1 = w1a w1b and
A2 = w2a w2b (rank-R GEMMs) and delta_w (``yat_hadamard_scale``) at the target weight's offset of three shadow buffers laid
out like the model's flat weights (so the fused q|k|v view has a fused delta view); the forward folds x delta_w^T into the base
GEMM through the ``pre_add`` epilogue, the backward adds dy delta_w to every input gradient, the ordinary weight-gradient
GEMMs leave d_delta_w in the frozen weights' gradient slots, and ``project()`` turns each into the four factor gradients
(``yat_hadamard_bwd`` + four rank-R GEMMs).  The adapter set owns one flat bf16 parameter / gradient buffer (rank padded to 8:
the padding rows / columns are zero and stay zero), so clip + AdamW and the data-parallel all-reduce are the usual launches;
an adapter dropped for a whole accumulation window is skipped by the optimizer like a ``grad is None`` parameter.
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import ops
from .lokr import is_target

BF16 = torch.bfloat16


class LoHaAdapters:
    def __init__(self, model, targets, r: int, alpha: float, module_dropout: float = 0.0):
        self.model, self.r, self.alpha, self.scale = model, int(r), float(alpha), float(alpha) / int(r)
        self.targets, self.module_dropout = list(targets), float(module_dropout or 0.0)
        self.R = R = (self.r + 7) // 8 * 8
        dev = model.flat_param.device
        self.entries, off, segs = [], 0, [0]
        base_ptr = model.flat_param.data_ptr()
        for key, w in model.P.items():
            if not key.endswith(".weight") or w.dim() < 2 or not is_target(key[:-7], self.targets):
                continue
            out_dim, in_dim = w.shape[0], w.numel() // w.shape[0]
            if out_dim % 8 or in_dim % 8:
                raise NotImplementedError(f"{key}: LoHa needs layer widths that are multiples of 8")
            e = dict(module=key[:-7], key=key, out=out_dim, inn=in_dim, w_off=(w.data_ptr() - base_ptr) // 2, o=off,
                     active=True, has_grad=False, steps=0)
            # w1a [out, R] | w1b [R, in] | w2a [out, R] | w2b [R, in]
            for n in (out_dim * R, R * in_dim, out_dim * R, R * in_dim):
                off += n
                segs.append(off)
            e["span"] = (e["o"], off)
            self.entries.append(e)
        if not self.entries:
            raise ValueError("no module matches lora_target_modules")
        self.numel_flat = off
        self.flat_param = torch.zeros(off, dtype=BF16, device=dev)
        self.flat_grad = torch.zeros(off, dtype=BF16, device=dev)
        self.seg_start = torch.tensor(sorted(set(segs)), dtype=torch.int64)
        self.bucket_bounds = [(0, off)]
        self.param_events = None
        self.grad_ready = None
        self.active_override = None
        # delta_w, A1 = w1a w1b, A2 = w2a w2b of every target at the target weight's offset (shadows of the flat weights)
        self.delta = torch.zeros_like(model.flat_param)
        self.A1 = torch.zeros_like(model.flat_param)
        self.A2 = torch.zeros_like(model.flat_param)
        big = max(e["out"] * e["inn"] for e in self.entries)
        self._t1 = torch.empty(big, dtype=BF16, device=dev)
        self._t2 = torch.empty(big, dtype=BF16, device=dev)
        self._lookup = {}
        self.reset_parameters()
        model.adapters = self

    # ---- views (padded to R): w1a [out, R], w1b [R, in], w2a [out, R], w2b [R, in]
    def _views(self, e, flat):
        o, R, out, inn = e["o"], self.R, e["out"], e["inn"]
        a = flat[o:o + out * R].view(out, R)
        b = flat[o + out * R:o + out * R + R * inn].view(R, inn)
        o2 = o + out * R + R * inn
        c = flat[o2:o2 + out * R].view(out, R)
        d = flat[o2 + out * R:o2 + out * R + R * inn].view(R, inn)
        return a, b, c, d

    def _shadow(self, e, buf):
        return buf[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])

    def delta_like(self, w):
        off = (w.data_ptr() - self.model.flat_param.data_ptr()) // 2
        return torch.as_strided(self.delta, w.size(), w.stride(), off)

    def lookup(self, t, base):
        off, n = (t.data_ptr() - base.data_ptr()) // 2, t.numel()
        hit = self._lookup.get((off, n))
        if hit is None:
            hit = [(e, (e["w_off"] - off) // e["inn"]) for e in self.entries if off <= e["w_off"] < off + n]
            self._lookup[(off, n)] = hit
        return hit

    def reset_parameters(self):
        """peft init_weights=True (LoHaLayer.reset_adapter_parameters [RECALL]): hada_w1_a, hada_w1_b, hada_w2_a
        kaiming_uniform(a=sqrt(5)) in that order (on the CPU, then cast), hada_w2_b zeros."""
        self.flat_param.zero_()
        r = self.r
        for e in self.entries:
            w1a, w1b, w2a, _ = self._views(e, self.flat_param)
            for t, shape in ((w1a[:, :r], (e["out"], r)), (w1b[:r], (r, e["inn"])), (w2a[:, :r], (e["out"], r))):
                init = torch.empty(shape, dtype=torch.float32)
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
        self.join_pending_update()
        first_micro = not getattr(self.model, "accumulate_grads", False)
        for e in self.entries:
            if first_micro:
                e["has_grad"] = False
            e["active"] = (not training) or self.module_dropout <= 0.0 or bool(torch.rand(1) > self.module_dropout)
            if training and self.active_override is not None:
                e["active"] = bool(self.active_override(e["module"]))
            d = self._shadow(e, self.delta)
            if not e["active"]:
                d.zero_()
                continue
            w1a, w1b, w2a, w2b = self._views(e, self.flat_param)
            a1, a2 = self._shadow(e, self.A1), self._shadow(e, self.A2)
            out, inn, R = e["out"], e["inn"], self.R
            ops.gemm(w1a, w1b, a1, b_t=True, M=out, N=inn, K=R, lda=R, ldb=inn, ldc=inn)          # w1a @ w1b
            ops.gemm(w2a, w2b, a2, b_t=True, M=out, N=inn, K=R, lda=R, ldb=inn, ldc=inn)          # w2a @ w2b
            ops.hadamard_scale(a1, a2, self.scale, d)

    def forward_term(self, x, w):
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return None
        tmp = torch.empty(x.shape[0], w.shape[0], dtype=BF16, device=x.device)
        if sum(e["out"] for e, _ in ents) != w.shape[0]:
            tmp.zero_()
            for e, row0 in ents:
                ops.gemm(x, self._shadow(e, self.delta), tmp[:, row0:row0 + e["out"]], M=x.shape[0], N=e["out"], K=e["inn"],
                         ldc=w.shape[0])
            return tmp
        return ops.linear_fwd(x, self.delta_like(w), None, out=tmp)          # (inactive entries: zero rows of delta)

    def dgrad_term(self, dy, w, dx):
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return {}
        if sum(e["out"] for e, _ in ents) == w.shape[0]:
            ops.linear_dgrad(dy, self.delta_like(w), out=dx, residual=dx)
        else:
            for e, row0 in ents:
                ops.gemm(dy[:, row0:row0 + e["out"]], self._shadow(e, self.delta), dx, b_t=True, M=dy.shape[0], N=e["inn"],
                         K=e["out"], lda=dy.stride(0), ldb=e["inn"], ldc=e["inn"], residual=dx)
        return {}

    def wgrad(self, dy, x, gw, accumulate=False, hs=None):
        """d_delta_w of the target(s) behind ``gw`` into their flat-gradient slots (the base weights are frozen)."""
        M, ld = dy.shape[0], dy.stride(0)
        acc_all = accumulate
        for e, row0 in self.lookup(gw, self.model.flat_grad):
            if not e["active"]:
                continue
            accumulate = acc_all and e["has_grad"]        # first active micro-step of a window overwrites (see lokr.py)
            e["has_grad"] = True
            g = self._shadow(e, self.model.flat_grad)
            ops.gemm(dy[:, row0:row0 + e["out"]], x, g, a_t=True, b_t=True, M=e["out"], N=e["inn"], K=M, lda=ld, ldb=e["inn"],
                     ldc=e["inn"], residual=g if accumulate else None)

    def project(self):
        """d_delta_w -> (d_w1a, d_w1b, d_w2a, d_w2b): HadaWeight.backward."""
        R = self.R
        for e in self.entries:
            g1a, g1b, g2a, g2b = self._views(e, self.flat_grad)
            if not e["active"]:
                if not e["has_grad"]:
                    for t in (g1a, g1b, g2a, g2b):
                        t.zero_()
                continue
            w1a, w1b, w2a, w2b = self._views(e, self.flat_param)
            out, inn = e["out"], e["inn"]
            dd = self._shadow(e, self.model.flat_grad)
            t1, t2 = self._t1[:out * inn].view(out, inn), self._t2[:out * inn].view(out, inn)
            ops.hadamard_bwd(dd, self._shadow(e, self.A1), self._shadow(e, self.A2), self.scale, t1, t2)
            ops.gemm(t1, w1b, g1a, M=out, N=R, K=inn, lda=inn, ldb=inn, ldc=R)                               # t1 @ w1b^T
            ops.gemm(w1a, t1, g1b, a_t=True, b_t=True, M=R, N=inn, K=out, lda=R, ldb=inn, ldc=inn)           # w1a^T @ t1
            ops.gemm(t2, w2b, g2a, M=out, N=R, K=inn, lda=inn, ldb=inn, ldc=R)
            ops.gemm(w2a, t2, g2b, a_t=True, b_t=True, M=R, N=inn, K=out, lda=R, ldb=inn, ldc=inn)
        if self.grad_ready is not None:
            self.grad_ready(0)

    def update_ranges(self):
        """[(lo, hi, step)] of the entries that have a gradient this window (see LoKrAdapters.update_ranges)."""
        out = []
        for e in self.entries:
            if not e["has_grad"]:
                continue
            e["steps"] += 1
            lo, hi = e["span"]
            if out and out[-1][1] == lo and out[-1][2] == e["steps"]:
                out[-1] = (out[-1][0], hi, e["steps"])
            else:
                out.append((lo, hi, e["steps"]))
        return out

    # ---- checkpoint (peft layout)
    def state_dict(self):
        self.join_pending_update()
        sd, r = {}, self.r
        for e in self.entries:
            w1a, w1b, w2a, w2b = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            sd[pre + "hada_w1_a"], sd[pre + "hada_w1_b"] = w1a[:, :r].contiguous(), w1b[:r].contiguous()
            sd[pre + "hada_w2_a"], sd[pre + "hada_w2_b"] = w2a[:, :r].contiguous(), w2b[:r].contiguous()
        return sd

    def load_state_dict(self, sd):
        r = self.r
        for e in self.entries:
            w1a, w1b, w2a, w2b = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            for t, name in ((w1a[:, :r], "hada_w1_a"), (w1b[:r], "hada_w1_b"), (w2a[:, :r], "hada_w2_a"), (w2b[:r], "hada_w2_b")):
                t.copy_(sd[pre + name].to(device=t.device, dtype=BF16))

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "adapter_model.safetensors"))
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LOHA", "r": self.r, "alpha": self.alpha, "module_dropout": self.module_dropout,
                       "target_modules": self.targets, "init_weights": True, "rank_dropout": 0.0,
                       "use_effective_conv2d": False}, f, indent=2)

    def num_parameters(self):
        return sum(2 * self.r * (e["out"] + e["inn"]) for e in self.entries)
