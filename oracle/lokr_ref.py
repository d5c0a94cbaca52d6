"""Torch-CPU restatement of the PEFT LoKr wrap of BASELINE config 5 (oracle, test-only).

Follows /root/reference/common/trainer.py:212-241: ``LoKrConfig(r=lora_rank, module_dropout=lora_dropout,
target_modules=lora_target_modules, alpha=lora_alpha)`` -> ``get_peft_model(model, config).to(dtype)``, then every
``model.parameters()`` (frozen base included) goes to AdamW (:243-248; frozen ones never get a gradient).

The adapter arithmetic lives in the unpinned third-party ``peft`` (requirements.txt:15), absent from this container, so
everything below is [RECALL peft/tuners/lokr/layer.py, peft/tuners/lycoris_utils.py] -- PARITY UNPINNED for this module:

* ``factorization(dim, factor=-1)``: the divisor pair (m <= n, m*n = dim) with the smallest m + n;
* Linear / 1x1 Conv2d with weight [out, in]: (out_l, out_k) = factorization(out), (in_m, in_n) = factorization(in);
  ``decompose_both=False`` -> ``lokr_w1`` is a full [out_l, in_m] matrix; ``r < max(out_k, in_n) / 2`` ->
  ``lokr_w2 = lokr_w2_a [out_k, r] @ lokr_w2_b [r, in_n]`` (otherwise a full [out_k, in_n] matrix);
* init (``init_weights=True``): w1 zeros, w2 / w2_a / w2_b kaiming_uniform(a=sqrt(5));
* ``delta_w = kron(w1, w2) * (alpha / r)`` (the multiply is skipped when the scale is exactly 1), reshaped to the base
  weight's shape; forward: ``base(x) + F.linear(x, delta_w)`` (``F.conv2d`` for the conv); in training the adapter term
  is dropped for the whole call when ``torch.rand(1) <= module_dropout``;
* target match: the module's dotted name equals a target or ends with ``"." + target``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def factorization(dimension: int, factor: int = -1):
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


def is_target(name: str, targets) -> bool:
    return any(name == t or name.endswith("." + t) for t in targets)


class LoKrWrapped(nn.Module):
    def __init__(self, base: nn.Module, r: int, alpha: float, module_dropout: float = 0.0):
        super().__init__()
        self.base_layer = base
        for p in base.parameters():
            p.requires_grad_(False)
        if isinstance(base, nn.Conv2d):
            assert base.kernel_size == (1, 1), "only 1x1 convolutions are targeted in SANA"
            out_dim, in_dim = base.out_channels, base.in_channels
        else:
            out_dim, in_dim = base.out_features, base.in_features
        self.out_l, self.out_k = factorization(out_dim)
        self.in_m, self.in_n = factorization(in_dim)
        self.r, self.scale, self.module_dropout = r, alpha / r, module_dropout
        dt = base.weight.dtype
        self.lokr_w1 = nn.Parameter(torch.zeros(self.out_l, self.in_m, dtype=dt))
        self.full_w2 = not (r < max(self.out_k, self.in_n) / 2)
        if self.full_w2:
            self.lokr_w2 = nn.Parameter(torch.empty(self.out_k, self.in_n, dtype=dt))
            nn.init.kaiming_uniform_(self.lokr_w2, a=math.sqrt(5))
        else:
            self.lokr_w2_a = nn.Parameter(torch.empty(self.out_k, r, dtype=dt))
            self.lokr_w2_b = nn.Parameter(torch.empty(r, self.in_n, dtype=dt))
            nn.init.kaiming_uniform_(self.lokr_w2_a, a=math.sqrt(5))
            nn.init.kaiming_uniform_(self.lokr_w2_b, a=math.sqrt(5))

    def delta_weight(self):
        w2 = self.lokr_w2 if self.full_w2 else self.lokr_w2_a @ self.lokr_w2_b
        rebuild = torch.kron(self.lokr_w1, w2.contiguous())
        if self.scale != 1:
            rebuild = rebuild * self.scale
        return rebuild.reshape(self.base_layer.weight.shape)

    def forward(self, x):
        result = self.base_layer(x)
        if (not self.training) or torch.rand(1) > self.module_dropout:
            dw = self.delta_weight()
            xin = x.to(dw.dtype)
            result = result + (F.conv2d(xin, dw) if isinstance(self.base_layer, nn.Conv2d) else F.linear(xin, dw))
        return result


def apply_lokr(model: nn.Module, targets, r: int, alpha: float, module_dropout: float = 0.0):
    """Wrap every target module in place (get_peft_model); freezes ALL base parameters.  Returns {dotted name: wrapper}."""
    for p in model.parameters():
        p.requires_grad_(False)
    wrapped = {}
    for name, mod in list(model.named_modules()):
        if not isinstance(mod, (nn.Linear, nn.Conv2d)) or not is_target(name, targets):
            continue
        parent = model
        parts = name.split(".")
        for part in parts[:-1]:
            parent = parent[int(part)] if part.isdigit() else getattr(parent, part)
        w = LoKrWrapped(mod, r, alpha, module_dropout)
        if parts[-1].isdigit():
            parent[int(parts[-1])] = w
        else:
            setattr(parent, parts[-1], w)
        wrapped[name] = w
    return wrapped
