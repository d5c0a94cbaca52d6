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

That "dense" application costs 5/3 of the full fine-tune's GEMM FLOPs.  The default is the **factored** application
(``mode='factored'``, per target when its dimensions allow 16-byte rows): the Kronecker structure is kept,
    adapter(x) = T1_flat P^T,   T1 = x' w2_b^T  (x' = x viewed [M*in_m, in_n]),   P = kron(w1, w2_a) * alpha/r  [out, in_m*r]
so a target costs two GEMMs with K = in_n and K = in_m*r (1/7 of the dense product at D = 2240, r = 8) in the forward, the
mirrored pair for the input gradient, ``d_P = dy^T T1_flat`` (again 1/7) + ``yat_lokr_project`` for (d_w1, d_w2_a) and one
streaming pass (``yat_lokr_small_wgrad``) for d_w2_b -- and the dense weight gradients of the frozen base are not computed at
all.  P is built by ``yat_lokr_delta`` with w2_b := identity, d_P projected by ``yat_lokr_project`` the same way.  The
adapter term is accumulated in fp32 and rounded once (peft rounds delta_w to bf16 first): at least as close to the fp32
truth as the dense path, checked by the same oracle test in both modes.
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


class SlabColumns:
    """Column slots of [rows, K] slabs for the small per-step products an adapter set hands to the base GEMMs as their second
    operand (row stride K = the row stride of the x they are computed from).  Slabs are zero-filled, kept from step to step and
    handed out in call order; ``restart()`` at the start of a step."""

    def __init__(self, device):
        self.device, self.slabs, self.next = device, {}, {}

    def restart(self):
        self.next = {}

    def take(self, M, K, width):
        slabs = self.slabs.setdefault(K, [])
        i, col = self.next.get(K, (0, 0))
        if col + width > K:
            i, col = i + 1, 0
        if i == len(slabs) or slabs[i].shape[0] < M:
            buf = torch.zeros(M, K, dtype=BF16, device=self.device)
            if i == len(slabs):
                slabs.append(buf)
            else:
                slabs[i] = buf                                  # (a larger batch than any before: this step's earlier slots
        self.next[K] = (i, col + width)                         #  live in the old buffer, which their views keep alive)
        return slabs[i][:M, col:col + width]


def adapted_linear(ad, x, w, bias=None, out=None, **ep):
    """Linear of a (possibly adapted) target, the ``lin`` hook of the models: base_layer(x) + adapter(x) (peft's wrap).  An
    adapter set that can hand its term over as the GEMM's second operand pair (``forward_pair``: the factored LoKr targets)
    costs one launch; otherwise the term is computed first and folded in through the ``pre_add`` epilogue."""
    if ad is None:
        return ops.linear_fwd(x, w, bias, out=out, **ep)
    pair = ad.forward_pair(x, w) if hasattr(ad, "forward_pair") else None
    if pair == "plain":
        return ops.linear_fwd(x, w, bias, out=out, **ep)
    if pair is not None:
        a2, b2, k2, group = pair
        return ops.linear_fwd(x, w, bias, out=out, a2=a2, b2=b2, k2=k2, a2_group_n=group, **ep)
    tmp = ad.forward_term(x, w)                                   # None: no adapter on this weight
    if tmp is None:
        return ops.linear_fwd(x, w, bias, out=out, **ep)
    return ops.linear_fwd(x, w, bias, out=out, pre_add=tmp, **ep)


class LoKrAdapters:
    def __init__(self, model, targets, r: int, alpha: float, module_dropout: float = 0.0, mode: str | None = None,
                 pair: bool = True):
        self.model, self.r, self.alpha, self.scale = model, int(r), float(alpha), float(alpha) / int(r)
        self.targets, self.module_dropout = list(targets), float(module_dropout)
        self.mode = mode or "factored"
        if self.mode not in ("factored", "dense"):
            raise ValueError(f"LoKr mode {self.mode!r}")
        self.R = (self.r + 7) // 8 * 8                      # rank padded to 16-byte rows (zero columns / rows)
        # factored targets in the forward: T1_flat P^T as the SECOND OPERAND PAIR of the base Linear's GEMM (one launch, no
        # [M, out] addend) instead of a GEMM of its own + the pre_add epilogue; see forward_pair()
        self.pair = bool(pair) and self.mode == "factored"
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
            if w.dim() == 4 and w.shape[2] * w.shape[3] != 1:
                # peft factorises the CHANNEL counts of a k x k convolution and hangs the kernel on lokr_w2: not this layout
                raise NotImplementedError(f"{key}: LoKr on a {w.shape[2]}x{w.shape[3]} convolution is not built; name the "
                                          f"linear targets more narrowly (e.g. 'net.0.proj' instead of 'proj')")
            out_dim, in_dim = w.shape[0], w.numel() // w.shape[0]
            (out_l, out_k), (in_m, in_n) = factorization(out_dim), factorization(in_dim)
            if not (self.r < max(out_k, in_n) / 2):
                raise NotImplementedError(f"{key}: full lokr_w2 (r >= max(out_k, in_n)/2) is not built")
            e = dict(module=key[:-7], key=key, out=out_dim, inn=in_dim, out_l=out_l, out_k=out_k, in_m=in_m, in_n=in_n,
                     w_off=(w.data_ptr() - base_ptr) // 2, o1=take(out_l * in_m), oa=take(out_k * self.r),
                     ob=take(self.r * in_n), active=True, has_grad=False, steps=0)
            e["span"] = (e["o1"], off)      # [o1, end of w2_b incl. pad): the entry's contiguous, 8-aligned parameter range
            # factored application needs 16-byte rows in every small GEMM: in_n, in_m*R, out multiples of 8
            e["factored"] = (self.mode == "factored" and in_n % 8 == 0 and in_n <= 128 and self.R <= 16 and out_dim % 8 == 0
                             and out_k * self.R * 4 <= 65536)
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
        self.active_override = None         # callable(module name) -> bool replacing the module-dropout draw (tests)
        R = self.R
        fact = [e for e in self.entries if e["factored"]]
        # dense-mode entries keep their delta_w in a shadow of the model's flat weights (allocated only when one exists)
        self.delta = torch.zeros_like(model.flat_param) if len(fact) < len(self.entries) else None
        if fact:
            # P = kron(w1, w2_a) * scale and its gradient, [out, in_m*R] per factored target, each in one flat buffer
            sizes = [e["out"] * e["in_m"] * R for e in fact]
            # P lives in a zero-filled shadow of the model's flat weights: target t's P = the first in_m*R columns of the rows of
            # W_t, with W_t's row stride -- what the GEMM's second operand pair wants (B2 with B's row stride; the fused q|k|v
            # view of the weights has its stacked P at the same place), the columns up to the next multiple of 64 stay zero
            self.pair = self.pair and all(e["in_m"] * R <= e["inn"] for e in fact)
            flatP = torch.zeros_like(model.flat_param) if self.pair else torch.zeros(sum(sizes), dtype=BF16, device=dev)
            self._flatP = flatP
            flatdP = torch.zeros(sum(sizes), dtype=BF16, device=dev)
            o = 0
            for e, n in zip(fact, sizes):
                k2 = e["in_m"] * R
                if self.pair:
                    e["P"] = flatP[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])[:, :k2]
                else:
                    e["P"] = flatP[o:o + n].view(e["out"], k2)
                e["dP"] = flatdP[o:o + n].view(e["out"], k2)
                o += n
                if R != self.r:                             # zero-padded copies of w2_a / w2_b and their gradients
                    e["wa_pad"] = torch.zeros(e["out_k"], R, dtype=BF16, device=dev)
                    e["wb_pad"] = torch.zeros(R, e["in_n"], dtype=BF16, device=dev)
                    e["ga_pad"] = torch.zeros(e["out_k"], R, dtype=BF16, device=dev)
            self._eye = torch.eye(R, dtype=BF16, device=dev)
            self._gb_dummy = torch.empty(R, R, dtype=BF16, device=dev)
        # projection workspaces, one per entry: an entry is projected right behind its weight gradient, on whatever stream
        # that runs (wgrad()), so two entries may be in flight at once
        sizes = [(int(ops._lib().yat_lokr_project_workspace_bytes(e["out_l"], e["out_k"], R if e["factored"] else e["in_n"]))
                  + 255) & ~255 for e in self.entries]
        ws_all = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        o = 0
        for e, n in zip(self.entries, sizes):
            e["ws"], o = ws_all[o:o + n], o + n
        self._lookup = {}
        self._slabs = SlabColumns(dev)                 # T1 of forward_pair(): columns of [rows, in] slabs
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

    def lookup(self, t, base):
        """Adapter entries whose target weight lies inside ``t`` (a view of the model's flat parameter or gradient buffer
        ``base``; a fused q|k|v view holds three) -> [(entry, first row of the target inside the view)]."""
        off, n = (t.data_ptr() - base.data_ptr()) // 2, t.numel()
        hit = self._lookup.get((off, n))
        if hit is None:
            hit = [(e, (e["w_off"] - off) // e["inn"]) for e in self.entries if off <= e["w_off"] < off + n]
            self._lookup[(off, n)] = hit
        return hit

    def _w2(self, e):
        """(w2_a [out_k, R], w2_b [R, in_n]) with the rank padded to R."""
        _, wa, wb = self._views(e, self.flat_param)
        return (wa, wb) if self.R == self.r else (e["wa_pad"], e["wb_pad"])

    # ---- application through the model's hooks (yat_amd/sana.py: lin / dgrad / weight-gradient emission)
    def forward_term(self, x, w):
        """x delta_w^T for the (possibly fused) target view ``w`` -> bf16 [M, rows(w)] for the GEMM's pre_add, or None."""
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return None
        M, rows, R = x.shape[0], w.shape[0], self.R
        tmp = torch.empty(M, rows, dtype=BF16, device=x.device)
        if all(not e["factored"] for e, _ in ents):
            ops.linear_fwd(x, self.delta_like(w), None, out=tmp)
            return tmp
        if sum(e["out"] for e, _ in ents) != rows:
            tmp.zero_()
        for e, row0 in ents:
            blk = tmp[:, row0:row0 + e["out"]]
            if not e["active"]:
                blk.zero_()
            elif not e["factored"]:
                d = self.delta[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
                ops.gemm(x, d, blk, M=M, N=e["out"], K=e["inn"], ldc=rows)
            else:
                im, n_ = e["in_m"], e["in_n"]
                t1 = ops.lokr_rows_fwd(x.view(M * im, n_), self._w2(e)[1], torch.empty(M * im, R, dtype=BF16, device=x.device))
                ops.gemm(t1.view(M, im * R), e["P"], blk, M=M, N=e["out"], K=im * R, ldb=e["P"].stride(0), ldc=rows)
                # kept for the weight gradient (same x): the reference lives until the next forward replaces it, i.e.
                # past every stream that reads it in the backward (which the main stream joins before the optimizer)
                e["t1"] = (x.data_ptr(), t1.view(M, im * R))
        return tmp

    def forward_pair(self, x, w):
        """The adapter term of target view ``w`` as the second operand pair of the base GEMM: (a2 [M, n k2] with x's row
        stride, b2 [rows(w), k2] with w's, k2, columns of the output per block of a2) -- or None when the view has no adapter,
        when every one of its adapters is dropped this step ("plain"), or when this form does not apply (then
        forward_term()).  T1 = x' w2_b^T of each entry is written straight into its columns of a slab; a dropped entry of a
        fused view gets zeros there.  The sum base + adapter is rounded ONCE (peft rounds the two terms and their sum)."""
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return None
        M, K, R = x.shape[0], x.shape[1], self.R
        e0 = ents[0][0]
        kr = e0["in_m"] * R
        k2 = (kr + 63) // 64 * 64
        applies = not (not self.pair or not x.is_contiguous() or w.stride(0) != K
                       or w.shape[0] != sum(e["out"] for e, _ in ents)
                       or any(not e["factored"] or e["in_m"] * R != kr or e["inn"] != K or e["out"] != e0["out"]
                              for e, _ in ents)
                       or len(ents) * k2 > K or (len(ents) > 1 and e0["out"] % 320 and e0["out"] % 256))
        # the slot is taken BEFORE the dropped-target return (round-5 advisor): slots are handed out in call order, so a
        # target that module dropout drops this step must still consume its slot -- every target then owns the same columns
        # step after step, and the padding columns kr..k2 of a slot (never written, zero from the slab's creation) can
        # never hold another target's stale T1
        a2 = self._slabs.take(M, K, len(ents) * k2) if applies else None
        if not any(e["active"] for e, _ in ents):
            return "plain"
        if not applies:
            return None
        for j, (e, row0) in enumerate(ents):
            assert row0 == j * e0["out"]
            t1 = a2[:, j * k2:j * k2 + kr]
            if e["active"]:
                ops.lokr_rows_fwd_flat(x.view(M * e["in_m"], e["in_n"]), self._w2(e)[1], t1, e["in_m"])
                e["t1"] = (x.data_ptr(), t1)                    # kept for the weight gradient, as in forward_term()
            else:
                t1.zero_()
        b2 = self._flatP[(w.data_ptr() - self.model.flat_param.data_ptr()) // 2:][:w.shape[0] * K].view(w.shape[0], K)[:, :k2]
        return a2, b2, k2, (e0["out"] if len(ents) > 1 else 0)

    def dgrad_term(self, dy, w, dx):
        """dx += dy delta_w for the target view ``w`` (dy [M, rows(w)], dx [M, in] contiguous).  Returns {id(entry): H}
        with H = dy_block P of every factored entry -- the weight gradient of the same dy needs it too."""
        hs = {}
        ents = self.lookup(w, self.model.flat_param)
        if not ents:
            return hs
        if all(not e["factored"] for e, _ in ents):
            ops.linear_dgrad(dy, self.delta_like(w), out=dx, residual=dx)
            return hs
        M, R, ld = dy.shape[0], self.R, dy.stride(0)
        for e, row0 in ents:
            if not e["active"]:
                continue
            dyb = dy[:, row0:row0 + e["out"]]
            if not e["factored"]:
                d = self.delta[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
                ops.gemm(dyb, d, dx, b_t=True, M=M, N=e["inn"], K=e["out"], lda=ld, ldb=e["inn"], ldc=e["inn"], residual=dx)
                continue
            im, n_ = e["in_m"], e["in_n"]
            h = torch.empty(M, im * R, dtype=BF16, device=dy.device)
            ops.gemm(dyb, e["P"], h, b_t=True, M=M, N=im * R, K=e["out"], lda=ld, ldb=e["P"].stride(0), ldc=im * R)
            ops.lokr_rows_bwd(h.view(M * im, R), self._w2(e)[1], dx.view(M * im, n_))
            hs[id(e)] = h
        return hs

    def wgrad(self, dy, x, gw, accumulate=False, hs=None):
        """Adapter-side weight gradient of the target(s) behind the gradient view ``gw``: dense entries get d_delta_w in
        their flat-gradient slot (as before), factored ones d_P and d_w2_b; a non-target weight (frozen, no adapter) gets
        nothing.  dy [M, rows(gw)] (row stride allowed), x [M, in] contiguous; ``hs``: the H products ``dgrad_term`` already
        made for this dy."""
        M, R, ld = dy.shape[0], self.R, dy.stride(0)
        acc_all = accumulate
        for e, row0 in self.lookup(gw, self.model.flat_grad):
            if not e["active"]:
                continue
            # gradient accumulation x module dropout: an entry dropped on the earlier micro-steps of this window holds
            # the PREVIOUS window's sums in d_P / d_w2_b / its dense slot -- its first active micro-step overwrites
            accumulate = acc_all and e["has_grad"]
            e["has_grad"] = True
            dyb = dy[:, row0:row0 + e["out"]]
            if not e["factored"]:
                g = self.model.flat_grad[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
                ops.gemm(dyb, x, g, a_t=True, b_t=True, M=e["out"], N=e["inn"], K=M, lda=ld, ldb=e["inn"], ldc=e["inn"],
                         residual=g if accumulate else None)
                self._project_entry(e)
                continue
            im, n_ = e["in_m"], e["in_n"]
            x2 = x.view(M * im, n_)
            kept = e.get("t1")
            if kept is not None and kept[0] == x.data_ptr() and kept[1].shape[0] == M:
                t1 = kept[1]                                 # T1_flat [M, in_m R] of this very x, computed by the forward
            else:                                            # (a view with a row stride when it sits in a forward_pair() slab)
                t1 = ops.lokr_rows_fwd(x2, self._w2(e)[1], torch.empty(M * im, R, dtype=BF16, device=x.device)).view(M, im * R)
            ops.gemm(dyb, t1, e["dP"], a_t=True, b_t=True, M=e["out"], N=im * R, K=M, lda=ld, ldb=t1.stride(0),
                     ldc=im * R, residual=e["dP"] if accumulate else None)
            h = hs.get(id(e)) if hs else None
            if h is None:
                h = torch.empty(M, im * R, dtype=BF16, device=dy.device)
                ops.gemm(dyb, e["P"], h, b_t=True, M=M, N=im * R, K=e["out"], lda=ld, ldb=e["P"].stride(0), ldc=im * R)
            else:
                h.record_stream(torch.cuda.current_stream())     # produced on the chain's stream, read on this one
            _, _, gb = self._views(e, self.flat_grad)
            ops.lokr_small_wgrad(h.view(M * im, R), x2, gb, accumulate=accumulate)
            # d_P is complete: project it here, on the weight-gradient stream beside the chain's GEMMs -- at the end of the
            # backward the ~200 x 3 small launches of project() were a serial tail (~10 ms at B = 32) with the chip idle
            self._project_entry(e)

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
        self._slabs.restart()               # this step's T1 products take the slab columns from the start again
        first_micro = not getattr(self.model, "accumulate_grads", False)
        for e in self.entries:
            e["projected"] = False
            if first_micro:
                e["has_grad"] = False       # a new accumulation window: nothing has contributed yet
            e["active"] = (not training) or self.module_dropout <= 0.0 or bool(torch.rand(1) > self.module_dropout)
            if training and self.active_override is not None:     # tests: a chosen drop pattern instead of the draw
                e["active"] = bool(self.active_override(e["module"]))
            w1, wa, wb = self._views(e, self.flat_param)
            if e["factored"]:
                if e["active"]:
                    if self.R != self.r:
                        e["wa_pad"][:, :self.r].copy_(wa)
                        e["wb_pad"][:self.r].copy_(wb)
                    ops.lokr_delta(w1, self._w2(e)[0], self._eye, self.scale, e["P"])      # kron(w1, w2_a) * scale
                continue
            d2 = self.delta[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
            if e["active"]:
                ops.lokr_delta(w1, wa, wb, self.scale, d2)
            else:
                d2.zero_()

    def project(self):
        """d_delta_w (the model's flat gradient slots of the frozen target weights) -> adapter gradients: whatever wgrad() has
        not projected already, the zeros of dropped entries, then the data-parallel hook."""
        for e in self.entries:
            g1, ga, gb = self._views(e, self.flat_grad)
            if not e["active"]:
                # dropped on this micro-step: contributes nothing.  If an earlier micro-step of the window was active its
                # sums stay as they are; an entry that has not contributed at all reads as zero gradients (the optimizer
                # skips it entirely, update_ranges())
                if not e["has_grad"]:
                    g1.zero_(); ga.zero_(); gb.zero_()
                continue
            if not e.get("projected"):      # (wgrad() projects an entry right behind its weight gradient)
                self._project_entry(e)
        if self.grad_ready is not None:
            self.grad_ready(0)

    def _project_entry(self, e):
        """Entry e's d_P (factored) / d_delta_w (dense) -> (d_w1, d_w2_a[, d_w2_b]) on the current stream.  Idempotent: under
        gradient accumulation d_P holds the window's sum so far and every micro-step rewrites the projections from it."""
        g1, ga, gb = self._views(e, self.flat_grad)
        w1, wa, wb = self._views(e, self.flat_param)
        if e["factored"]:
            # d_P -> (d_w1, d_w2_a) by the autograd of kron(w1, w2_a) * scale; d_w2_b was written by wgrad()
            ga_r = ga if self.R == self.r else e["ga_pad"]
            ops.lokr_project(w1, self._w2(e)[0], self._eye, self.scale, e["dP"], g1, ga_r, self._gb_dummy, e["ws"])
            if ga_r is not ga:
                ga.copy_(ga_r[:, :self.r])
        else:
            dd = self.model.flat_grad[e["w_off"]:e["w_off"] + e["out"] * e["inn"]].view(e["out"], e["inn"])
            ops.lokr_project(w1, wa, wb, self.scale, dd, g1, ga, gb, e["ws"])
        e["projected"] = True

    def update_ranges(self):
        """Parameter ranges the optimizer step must touch, with each range's own step count: [(lo, hi, step)].  peft leaves
        a dropped adapter's ``.grad`` None, so torch.optim.AdamW skips it altogether -- no parameter, moment or weight-decay
        update, and its per-parameter ``step`` (the bias correction) does not advance.  Entries are contiguous in the flat
        buffer; neighbours with the same count share a launch."""
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
