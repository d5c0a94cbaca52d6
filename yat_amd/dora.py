"""DoRA adapters on the HIP path (``lora_algo: lora`` + ``lora_use_dora`` -- the reference wraps the transformer with peft's
``LoraConfig(r, lora_dropout, target_modules, lora_alpha, use_dora=True)`` at common/trainer.py:215-220).

Arithmetic [RECALL peft/tuners/lora/dora.py -- parity unpinned, oracle/dora_ref.py is the restatement]: per target with weight
W [out, in], the LoRA pair (A [r, in], B [out, r], scaling = alpha / r) and a trainable magnitude m [out] (init: row norms of W):
    n = ||W + scaling B A||_2 per row (detached),  s = m / n
    result = base(x) + (s - 1) (x W^T) + s scaling (x A^T) B^T        =  base(x) + x delta^T,   delta = s (W + scaling B A) - W.

MI355X mapping -- the *dense* application shared with yat_amd/loha.py: per step ``materialize()`` builds lw = B A (a rank-R
GEMM) and, in one pass per output row (``yat_dora_delta``), the norm, s and the delta row, into shadow buffers laid out like the
model's flat weights (a fused q|k|v view has a fused delta view); the forward folds x delta^T into the base GEMM through the
``pre_add`` epilogue, the backward adds dy delta to every input gradient, the ordinary weight-gradient GEMMs leave d_delta in
the frozen weights' gradient slots, and ``project()`` turns each into d_m (a row dot product with W + scaling B A, divided by n)
and the scaled gradient whose two rank-R GEMMs are d_B and d_A (``yat_dora_bwd``).  One flat bf16 parameter / gradient buffer
(rank padded to 8, magnitudes padded to 8 per target), so clip + AdamW and the data-parallel all-reduce are the usual launches.
``lora_dropout`` > 0 changes peft's formula (the base product is recomputed on the dropped input) and is refused.
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import ops
from .lokr import is_target

BF16 = torch.bfloat16


class DoRAAdapters:
    def __init__(self, model, targets, r: int, alpha: float, dropout: float = 0.0, use_rslora: bool = False):
        if dropout and dropout > 0:
            raise NotImplementedError("DoRA with lora_dropout > 0 is not built (peft then recomputes the base product on the "
                                      "dropped input)")
        self.model, self.r, self.alpha = model, int(r), float(alpha)
        self.use_rslora = bool(use_rslora)
        self.scale = float(alpha) / (math.sqrt(r) if use_rslora else int(r))
        self.targets = list(targets)
        self.R = R = (self.r + 7) // 8 * 8
        dev = model.flat_param.device
        self.entries, off, segs, noff = [], 0, [0], 0
        base_ptr = model.flat_param.data_ptr()
        for key, w in model.P.items():
            if not key.endswith(".weight") or w.dim() < 2 or not is_target(key[:-7], self.targets):
                continue
            if w.dim() == 4 and w.shape[2] * w.shape[3] != 1:
                raise NotImplementedError(f"{key}: DoRA on a {w.shape[2]}x{w.shape[3]} convolution is not built; name the linear "
                                          f"targets more narrowly")
            out_dim, in_dim = w.shape[0], w.numel() // w.shape[0]
            if out_dim % 8 or in_dim % 8:
                raise NotImplementedError(f"{key}: DoRA needs layer widths that are multiples of 8")
            e = dict(module=key[:-7], key=key, out=out_dim, inn=in_dim, w_off=(w.data_ptr() - base_ptr) // 2, o=off, n_off=noff,
                     active=True, has_grad=False, steps=0)
            # lora_B [out, R] | lora_A [R, in] | magnitude [out]
            for n in (out_dim * R, R * in_dim, out_dim):
                off += n
                segs.append(off)
            noff += out_dim
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
        # delta and lw = B A of every target at the target weight's offset (shadows of the flat weights); s and n per output row
        self.delta = torch.zeros_like(model.flat_param)
        self.lw = torch.zeros_like(model.flat_param)
        self.s_buf = torch.zeros(noff, dtype=torch.float32, device=dev)
        self.n_buf = torch.ones(noff, dtype=torch.float32, device=dev)
        big = max(e["out"] * e["inn"] for e in self.entries)
        self._t1 = torch.empty(big, dtype=BF16, device=dev)
        self._lookup = {}
        self.reset_parameters()
        model.adapters = self

    # ---- views (rank padded to R): lora_B [out, R], lora_A [R, in], magnitude [out]
    def _views(self, e, flat):
        o, R, out, inn = e["o"], self.R, e["out"], e["inn"]
        b = flat[o:o + out * R].view(out, R)
        a = flat[o + out * R:o + out * R + R * inn].view(R, inn)
        m = flat[o + out * R + R * inn:o + out * R + R * inn + out]
        return b, a, m

    def _shadow(self, e, buf):
        return buf[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])

    def _rows(self, e, buf):
        return buf[e["n_off"]:e["n_off"] + e["out"]]

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
        """peft: lora_A kaiming_uniform(a=sqrt(5)) on the CPU then cast, lora_B zeros, magnitude = ||W||_2 per row."""
        self.flat_param.zero_()
        r = self.r
        for e in self.entries:
            _, a, m = self._views(e, self.flat_param)
            init = torch.empty(r, e["inn"], dtype=torch.float32)
            torch.nn.init.kaiming_uniform_(init, a=math.sqrt(5))
            a[:r].copy_(init.to(BF16))
            m.copy_(torch.linalg.norm(self._shadow(e, self.model.flat_param).float(), dim=1).to(BF16))

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
        R = self.R
        for e in self.entries:
            if first_micro:
                e["has_grad"] = False
            b, a, m = self._views(e, self.flat_param)
            out, inn = e["out"], e["inn"]
            lw = self._shadow(e, self.lw)
            ops.gemm(b, a, lw, b_t=True, M=out, N=inn, K=R, lda=R, ldb=inn, ldc=inn)                       # lora_B @ lora_A
            ops.dora_delta(self._shadow(e, self.model.flat_param), lw, m, self.scale, self._shadow(e, self.delta),
                           self._rows(e, self.s_buf), self._rows(e, self.n_buf))

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
        return ops.linear_fwd(x, self.delta_like(w), None, out=tmp)

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
        """d_delta of the target(s) behind ``gw`` into their flat-gradient slots (the base weights are frozen)."""
        M, ld = dy.shape[0], dy.stride(0)
        acc_all = accumulate
        for e, row0 in self.lookup(gw, self.model.flat_grad):
            accumulate = acc_all and e["has_grad"]
            e["has_grad"] = True
            g = self._shadow(e, self.model.flat_grad)
            ops.gemm(dy[:, row0:row0 + e["out"]], x, g, a_t=True, b_t=True, M=e["out"], N=e["inn"], K=M, lda=ld, ldb=e["inn"],
                     ldc=e["inn"], residual=g if accumulate else None)

    def project(self):
        """d_delta -> (d_lora_B, d_lora_A, d_magnitude)."""
        R = self.R
        for e in self.entries:
            gb, ga, gm = self._views(e, self.flat_grad)
            if not e["has_grad"]:
                for t in (gb, ga, gm):
                    t.zero_()
                continue
            b, a, _ = self._views(e, self.flat_param)
            out, inn = e["out"], e["inn"]
            t1 = self._t1[:out * inn].view(out, inn)
            ops.dora_bwd(self._shadow(e, self.model.flat_grad), self._shadow(e, self.model.flat_param), self._shadow(e, self.lw),
                         self.scale, self._rows(e, self.s_buf), self._rows(e, self.n_buf), t1, gm)
            ops.gemm(t1, a, gb, M=out, N=R, K=inn, lda=inn, ldb=inn, ldc=R)                                  # t1 @ lora_A^T
            ops.gemm(b, t1, ga, a_t=True, b_t=True, M=R, N=inn, K=out, lda=R, ldb=inn, ldc=inn)              # lora_B^T @ t1
        if self.grad_ready is not None:
            self.grad_ready(0)

    # ---- checkpoint (peft layout)
    def state_dict(self):
        self.join_pending_update()
        sd, r = {}, self.r
        for e in self.entries:
            b, a, m = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            sd[pre + "lora_A.weight"], sd[pre + "lora_B.weight"] = a[:r].contiguous(), b[:, :r].contiguous()
            sd[pre + "lora_magnitude_vector.weight"] = m.clone()          # [RECALL] ModuleDict of DoraLinearLayer (peft >= 0.11)
        return sd

    def load_state_dict(self, sd):
        r = self.r
        for e in self.entries:
            b, a, m = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            a[:r].copy_(sd[pre + "lora_A.weight"].to(device=a.device, dtype=BF16).reshape(r, -1))
            b[:, :r].copy_(sd[pre + "lora_B.weight"].to(device=b.device, dtype=BF16).reshape(e["out"], r))
            key = pre + "lora_magnitude_vector.weight"
            if key not in sd:
                key = pre + "lora_magnitude_vector"                        # peft < 0.11 kept a ParameterDict
            m.copy_(sd[key].to(device=m.device, dtype=BF16).reshape(-1))

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "adapter_model.safetensors"))
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LORA", "r": self.r, "lora_alpha": self.alpha, "lora_dropout": 0.0, "use_dora": True,
                       "use_rslora": self.use_rslora, "target_modules": self.targets, "init_lora_weights": True,
                       "bias": "none"}, f, indent=2)

    def num_parameters(self):
        return sum(self.r * (e["out"] + e["inn"]) + e["out"] for e in self.entries)
