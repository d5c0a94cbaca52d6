"""Torch-CPU restatement of the PEFT LoHa wrap (oracle, test-only).

Follows /root/reference/common/trainer.py:220-224: ``LoHaConfig(r=lora_rank, module_dropout=lora_dropout,
target_modules=lora_target_modules, alpha=lora_alpha)`` -> ``get_peft_model(model, config).to(dtype)``.
The adapter arithmetic lives in the unpinned third-party ``peft`` (requirements.txt:15), absent from this container, so
everything below is [RECALL peft/tuners/loha/layer.py] -- PARITY UNPINNED for this module:

* Linear / 1x1 Conv2d (``use_effective_conv2d=False``) with weight [out, in]: ``hada_w1_a [out, r]``, ``hada_w1_b [r, in]``,
  ``hada_w2_a [out, r]``, ``hada_w2_b [r, in]``; init (``init_weights=True``, ``LoHaLayer.reset_adapter_parameters``): w1_a, w1_b,
  w2_a kaiming_uniform(a=sqrt(5)), w2_b zeros (so the initial delta is zero and w2_b receives the first gradients);
* ``delta_w = HadaWeight.apply(w1a, w1b, w2a, w2b, scale)`` with scale = alpha / r:
  forward ``((w1a @ w1b) * (w2a @ w2b)) * scale``; the hand-written backward
  ``g = grad * scale; t = g * (w2a @ w2b); d_w1a = t @ w1b.T; d_w1b = w1a.T @ t; t = g * (w1a @ w1b); d_w2a = t @ w2b.T;
  d_w2b = w2a.T @ t`` -- every op in the module dtype;
* forward: ``base(x) + F.linear(x, delta_w)`` (``F.conv2d`` for the conv); in training the adapter term is dropped for the
  whole call when ``torch.rand(1) <= module_dropout``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lokr_ref import is_target


class HadaWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w1a, w1b, w2a, w2b, scale):
        ctx.save_for_backward(w1a, w1b, w2a, w2b, scale)
        return ((w1a @ w1b) * (w2a @ w2b)) * scale

    @staticmethod
    def backward(ctx, grad_out):
        w1a, w1b, w2a, w2b, scale = ctx.saved_tensors
        grad_out = grad_out * scale
        temp = grad_out * (w2a @ w2b)
        grad_w1a = temp @ w1b.T
        grad_w1b = w1a.T @ temp
        temp = grad_out * (w1a @ w1b)
        grad_w2a = temp @ w2b.T
        grad_w2b = w2a.T @ temp
        return grad_w1a, grad_w1b, grad_w2a, grad_w2b, None


class LoHaWrapped(nn.Module):
    def __init__(self, base: nn.Module, r: int, alpha: float, module_dropout: float = 0.0):
        super().__init__()
        self.base_layer = base
        for p in base.parameters():
            p.requires_grad_(False)
        if isinstance(base, nn.Conv2d):
            assert base.kernel_size == (1, 1), "only 1x1 convolutions are targeted here"
            out_dim, in_dim = base.out_channels, base.in_channels
        else:
            out_dim, in_dim = base.out_features, base.in_features
        self.r, self.scale, self.module_dropout = r, alpha / r, module_dropout
        dt = base.weight.dtype
        self.hada_w1_a = nn.Parameter(torch.empty(out_dim, r, dtype=dt))
        self.hada_w1_b = nn.Parameter(torch.empty(r, in_dim, dtype=dt))
        self.hada_w2_a = nn.Parameter(torch.empty(out_dim, r, dtype=dt))
        self.hada_w2_b = nn.Parameter(torch.zeros(r, in_dim, dtype=dt))
        nn.init.kaiming_uniform_(self.hada_w1_a, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.hada_w1_b, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.hada_w2_a, a=math.sqrt(5))

    def delta_weight(self):
        w = HadaWeight.apply(self.hada_w1_a, self.hada_w1_b, self.hada_w2_a, self.hada_w2_b, torch.tensor(self.scale))
        return w.reshape(self.base_layer.weight.shape)

    def forward(self, x):
        result = self.base_layer(x)
        if (not self.training) or torch.rand(1) > self.module_dropout:
            dw = self.delta_weight()
            xin = x.to(dw.dtype)
            result = result + (F.conv2d(xin, dw) if isinstance(self.base_layer, nn.Conv2d) else F.linear(xin, dw))
        return result


def apply_loha(model: nn.Module, targets, r: int, alpha: float, module_dropout: float = 0.0):
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
        w = LoHaWrapped(mod, r, alpha, module_dropout)
        if parts[-1].isdigit():
            parent[int(parts[-1])] = w
        else:
            setattr(parent, parts[-1], w)
        wrapped[name] = w
    return wrapped
