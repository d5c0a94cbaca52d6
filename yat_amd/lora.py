"""Plain LoRA adapters on the HIP path (``lora_algo: lora`` -- the reference wraps the transformer with peft's
``LoraConfig(r, lora_alpha, lora_dropout, target_modules, use_dora)`` at common/trainer.py:213-219 and hands every parameter
to AdamW :243-248).

Arithmetic [RECALL peft/tuners/lora/layer.py -- parity unpinned, see oracle/lora_ref.py for the restatement]: for a target
Linear / 1x1 Conv with weight W [out, in]:
    result = base_layer(x) + lora_B(lora_A(dropout(x))) * scaling,   scaling = lora_alpha / r
with lora_A [r, in] kaiming-uniform(a=sqrt(5)), lora_B [out, r] zeros, every op in the module dtype (bf16).

MI355X mapping (same interface as yat_amd/lokr.py, so yat_amd/sana.py's hooks do not change): the adapter set owns one flat
bf16 buffer (per target: A [R, in] then B^T [R, out], R = the rank padded to 8 so every skinny GEMM keeps 16-byte rows; the
padding rows stay exactly zero: their gradients are zero) with a flat gradient twin -- clip + AdamW and the data-parallel
all-reduce are the usual single launches.  Per target and step:
    forward   T = x A^T (GEMM, N = R);  adapter = bf16(bf16(T B^T) * scaling)  (yat_rank_expand: a K = R product is all epilogue,
              so it is a stream, not a GEMM) -> pre_add
    dgrad     dT = scaling * (dy B)     (GEMM, N = R, the scalar as a constant gate row);  dx += dT A   (yat_rank_expand)
    wgrad     d_B^T = scaling * T^T dy,  d_A = dT^T x  -- R x width outputs over a B*N-row reduction: yat_lokr_small_wgrad
The dense weight gradients of the frozen base are never computed.  ``scaling`` is applied to the small side of each product
(the reference rounds ``u * scaling`` and ``dy * scaling`` element-wise: identical when scaling is a power of two, one bf16
rounding apart otherwise).  ``lora_dropout`` > 0: ``yat_dropout`` with a counter-based mask (hash of a per-(step, layer) seed and
the element index) that forward, input gradient and weight gradient regenerate -- the mask *stream* cannot equal torch's
Philox stream, the arithmetic (x * mask / (1 - p), one rounding) does.  DoRA is not built.
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import ops
from .lokr import is_target, SlabColumns

BF16 = torch.bfloat16


class LoRAAdapters:
    def __init__(self, model, targets, r: int, alpha: float, dropout: float = 0.0, use_rslora: bool = False, seed: int = 0,
                 pair: bool = True):
        if not (0.0 <= float(dropout or 0.0) < 1.0):
            raise ValueError("lora_dropout must be in [0, 1)")
        self.model, self.r, self.alpha, self.use_rslora = model, int(r), float(alpha), bool(use_rslora)
        self.dropout, self._seed, self._step = float(dropout or 0.0), int(seed), 0
        # [RECALL peft] scaling = lora_alpha / r, or lora_alpha / sqrt(r) with use_rslora
        self.scale = float(alpha) / (math.sqrt(int(r)) if use_rslora else int(r))
        self.targets = list(targets)
        self.R = R = (self.r + 7) // 8 * 8
        if R > 16:
            raise NotImplementedError("LoRA rank > 16")
        dev = model.flat_param.device
        self.entries, off, segs = [], 0, [0]
        base_ptr = model.flat_param.data_ptr()
        for key, w in model.P.items():
            if not key.endswith(".weight") or w.dim() < 2 or not is_target(key[:-7], self.targets):
                continue
            out_dim, in_dim = w.shape[0], w.numel() // w.shape[0]
            if out_dim % 8 or in_dim % 8:
                raise NotImplementedError(f"{key}: LoRA needs layer widths that are multiples of 8")
            e = dict(module=key[:-7], key=key, out=out_dim, inn=in_dim, w_off=(w.data_ptr() - base_ptr) // 2,
                     oa=off, ob=off + R * in_dim, active=True, index=len(self.entries))
            off += R * (in_dim + out_dim)
            segs += [e["ob"], off]
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
        self._gate = torch.full((max(max(e["out"] for e in self.entries), R),), self.scale, dtype=BF16, device=dev)
        self._lookup = {}
        # forward through the base GEMM's second operand pair (forward_pair()): scaling * lora_B of every adapter in the first R
        # columns of its target's rows of a zero-filled shadow of the flat weights, rebuilt by one launch per step; T in slab columns
        # (per target: 64 columns must fit a row, and the rows of the shadow must be 16-byte aligned)
        for e in self.entries:
            e["pair_ok"] = bool(pair) and e["inn"] >= 64 and e["w_off"] % 8 == 0
        paired = [e for e in self.entries if e["pair_ok"]]
        self.pair = bool(paired)
        if self.pair:
            self._flatB = torch.zeros_like(model.flat_param)
            self._slabs = SlabColumns(dev)
            self._b_table = torch.tensor([[e["ob"], e["w_off"], e["out"], e["inn"]] for e in paired], dtype=torch.int64).to(dev)
            self._n_paired, self._max_out = len(paired), max(e["out"] for e in paired)
        self.reset_parameters()
        model.adapters = self

    # ---- views: A [R, in] (rows >= r zero), B^T [R, out] (rows >= r zero)
    def _views(self, e, flat):
        return (flat[e["oa"]:e["oa"] + self.R * e["inn"]].view(self.R, e["inn"]),
                flat[e["ob"]:e["ob"] + self.R * e["out"]].view(self.R, e["out"]))

    def reset_parameters(self):
        """peft init_lora_weights=True: lora_A kaiming_uniform(a=sqrt(5)) drawn on the CPU then cast, lora_B zeros."""
        self.flat_param.zero_()
        for e in self.entries:
            a, _ = self._views(e, self.flat_param)
            init = torch.empty(self.r, e["inn"], dtype=torch.float32)
            torch.nn.init.kaiming_uniform_(init, a=math.sqrt(5))
            a[:self.r].copy_(init.to(BF16))

    def join_pending_update(self):
        pev, self.param_events = self.param_events, None
        if pev is not None:
            cur = torch.cuda.current_stream()
            for ev in pev:
                cur.wait_event(ev)

    def lookup(self, t, base):
        off, n = (t.data_ptr() - base.data_ptr()) // 2, t.numel()
        hit = self._lookup.get((off, n))
        if hit is None:
            hit = [(e, (e["w_off"] - off) // e["inn"]) for e in self.entries if off <= e["w_off"] < off + n]
            self._lookup[(off, n)] = hit
        return hit

    # ---- per step (interface of yat_amd/lokr.py)
    def materialize(self, training=True):
        self.join_pending_update()
        if self.pair:                              # b2 of every target for this step; T takes the slab columns from the start
            ops.lora_scatter_b(self._b_table, self._n_paired, self._max_out, self.R, self.scale, self.flat_param, self._flatB)
            self._slabs.restart()
        # nn.Dropout on the adapter input: a fresh mask per layer and step in training, none in eval.  The mask is a hash of
        # (seed, element): forward, input gradient and weight gradient regenerate it from the per-(step, layer) seed.
        self._drop = self.dropout if (training and self.dropout > 0.0) else 0.0
        self._step += 1

    def _mask_seed(self, e):
        return (self._seed * 1000003 + self._step) * 4096 + e["index"]

    def _dropped(self, e, x):
        """dropout(x) for layer e (x itself when dropout is off)."""
        return ops.dropout(x, self._drop, self._mask_seed(e)) if self._drop > 0.0 else x

    def forward_term(self, x, w):
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return None
        M, rows, R = x.shape[0], w.shape[0], self.R
        tmp = torch.empty(M, rows, dtype=BF16, device=x.device)
        if sum(e["out"] for e, _ in ents) != rows:
            tmp.zero_()
        for e, row0 in ents:
            a, bt = self._views(e, self.flat_param)
            t = torch.empty(M, R, dtype=BF16, device=x.device)
            ops.gemm(self._dropped(e, x), a, t, M=M, N=R, K=e["inn"])                      # T = dropout(x) A^T
            ops.rank_expand(t, bt, tmp[:, row0:row0 + e["out"]], scale=self.scale)         # bf16(bf16(T B^T) * scaling)
            e["t"] = (x.data_ptr(), t)             # kept for d_B (see lokr.py on the lifetime)
        return tmp

    def forward_pair(self, x, w):
        """The adapter term of target view ``w`` as the second operand pair of the base GEMM (interface of yat_amd/lokr.py):
        a2 = T = dropout(x) A^T in the first R of 64 slab columns per adapter (row stride of x), b2 = scaling * lora_B in the
        shadow of the weights, k2 = 64.  base + adapter is accumulated in fp32 and rounded once (peft rounds u, u * scaling and
        the sum; with scaling a power of two the products are the same numbers)."""
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return None
        M, K, R = x.shape[0], x.shape[1], self.R
        e0 = ents[0][0]
        if not self.pair or not x.is_contiguous() or w.stride(0) != K or w.shape[0] != sum(e["out"] for e, _ in ents) \
                or any(not e["pair_ok"] or e["inn"] != K or e["out"] != e0["out"] for e, _ in ents) or len(ents) * 64 > K \
                or (len(ents) > 1 and e0["out"] % 320 and e0["out"] % 256):
            return None
        a2 = self._slabs.take(M, K, len(ents) * 64)
        for j, (e, row0) in enumerate(ents):
            assert row0 == j * e0["out"]
            a, _ = self._views(e, self.flat_param)
            # T is computed contiguous and copied into its slab columns (round-5 advisor: "write it straight into the slab"):
            # the weight gradient d_B = scaling T^T dy reads T through yat_lokr_small_wgrad, whose first operand has no row
            # stride -- an [M, 8] copy (0.5 MB at B = 32) is cheaper than a second T product.  Slots are fixed per target
            # (plain LoRA never returns "plain"), so the columns R..63 of a slot stay the zeros the slab was created with.
            t = torch.empty(M, R, dtype=BF16, device=x.device)
            ops.gemm(self._dropped(e, x), a, t, M=M, N=R, K=K)                                # T = dropout(x) A^T
            a2[:, j * 64:j * 64 + R].copy_(t)
            e["t"] = (x.data_ptr(), t)
        b2 = self._flatB[(w.data_ptr() - self.model.flat_param.data_ptr()) // 2:][:w.shape[0] * K].view(w.shape[0], K)[:, :64]
        return a2, b2, 64, (e0["out"] if len(ents) > 1 else 0)

    def dgrad_term(self, dy, w, dx):
        hs = {}
        M, R, ld = dy.shape[0], self.R, dy.stride(0)
        for e, row0 in self.lookup(w, self.model.flat_param):
            a, bt = self._views(e, self.flat_param)
            dt = torch.empty(M, R, dtype=BF16, device=dy.device)
            ops.gemm(dy[:, row0:row0 + e["out"]], bt, dt, M=M, N=R, K=e["out"], lda=ld, ldb=e["out"], ldc=R,
                     gate=self._gate, ld_gate=0, rows_per_batch=M)                          # dT = scaling * (dy B)
            if self._drop > 0.0:                                                            # dx += mask * (dT A) / (1 - p)
                g_ = ops.rank_expand(dt, a, torch.empty(M, e["inn"], dtype=BF16, device=dy.device))
                ops.dropout_bwd_add(g_, self._drop, self._mask_seed(e), dx)
            else:
                ops.rank_expand(dt, a, dx, residual=True)                                   # dx += dT A
            hs[id(e)] = dt
        return hs

    def wgrad(self, dy, x, gw, accumulate=False, hs=None):
        M, R, ld = dy.shape[0], self.R, dy.stride(0)
        for e, row0 in self.lookup(gw, self.model.flat_grad):
            a, bt = self._views(e, self.flat_param)
            ga, gbt = self._views(e, self.flat_grad)
            dyb = dy[:, row0:row0 + e["out"]]
            kept = e.get("t")
            if kept is not None and kept[0] == x.data_ptr() and kept[1].shape[0] == M:
                t = kept[1]
            else:
                t = torch.empty(M, R, dtype=BF16, device=x.device)
                ops.gemm(self._dropped(e, x), a, t, M=M, N=R, K=e["inn"])
            dt = hs.get(id(e)) if hs else None
            if dt is None:
                dt = torch.empty(M, R, dtype=BF16, device=dy.device)
                ops.gemm(dyb, bt, dt, M=M, N=R, K=e["out"], lda=ld, ldb=e["out"], ldc=R, gate=self._gate, ld_gate=0,
                         rows_per_batch=M)
            else:
                dt.record_stream(torch.cuda.current_stream())
            ops.lokr_small_wgrad(t, dyb, gbt[:self.r], accumulate=accumulate, scale=self.scale)   # d_B^T = scaling T^T dy
            ops.lokr_small_wgrad(dt, self._dropped(e, x), ga[:self.r], accumulate=accumulate)     # d_A = dT^T dropout(x)

    def project(self):
        if self.grad_ready is not None:            # gradients are complete as written; only the DDP hook remains
            self.grad_ready(0)

    # ---- checkpoint (peft layout)
    def state_dict(self):
        self.join_pending_update()
        sd = {}
        for e in self.entries:
            a, bt = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            sd[pre + "lora_A.weight"], sd[pre + "lora_B.weight"] = a[:self.r], bt[:self.r].t()
        return sd

    def load_state_dict(self, sd):
        for e in self.entries:
            a, bt = self._views(e, self.flat_param)
            pre = f"base_model.model.{e['module']}."
            a[:self.r].copy_(sd[pre + "lora_A.weight"].to(device=a.device, dtype=BF16).view(self.r, e["inn"]))
            bt[:self.r].copy_(sd[pre + "lora_B.weight"].to(device=a.device, dtype=BF16).view(e["out"], self.r).t())

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "adapter_model.safetensors"))
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LORA", "r": self.r, "lora_alpha": self.alpha, "lora_dropout": self.dropout,
                       "target_modules": self.targets, "use_dora": False, "use_rslora": self.use_rslora, "bias": "none",
                       "init_lora_weights": True}, f, indent=2)

    def num_parameters(self):
        return sum(self.r * (e["inn"] + e["out"]) for e in self.entries)
