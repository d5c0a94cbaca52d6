"""Torch-CPU restatement of the PEFT LoRA wrap (``lora_algo: lora``; oracle, test-only).

Follows /root/reference/common/trainer.py:213-219,237-241: ``LoraConfig(r=lora_rank, lora_dropout=lora_dropout,
target_modules=lora_target_modules, lora_alpha=lora_alpha, use_dora=lora_use_dora)`` -> ``get_peft_model(model,
config).to(dtype)``.  The adapter arithmetic lives in the unpinned third-party ``peft`` (requirements.txt:15), absent from this
container, so everything below is [RECALL peft/tuners/lora/layer.py] -- PARITY UNPINNED for this module:

* Linear / 1x1 Conv2d target with weight [out, in]: ``lora_A`` = Linear(in, r, bias=False) (a 1x1 Conv2d for the conv),
  ``lora_B`` = Linear(r, out, bias=False); init (``init_lora_weights=True``): A kaiming_uniform(a=sqrt(5)), B zeros;
* ``scaling = lora_alpha / r`` (``/ sqrt(r)`` with rsLoRA, not restated);
* forward: ``result = base_layer(x); result = result + lora_B(lora_A(dropout(x))) * scaling`` with ``dropout`` =
  nn.Dropout(lora_dropout) (Identity at 0), all in the module dtype;
* target match: the module's dotted name equals a target or ends with ``"." + target``; all base parameters are frozen.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lokr_ref import is_target


class LoRAWrapped(nn.Module):
    def __init__(self, base: nn.Module, r: int, alpha: float, dropout: float = 0.0):
        super().__init__()
        self.base_layer = base
        for p in base.parameters():
            p.requires_grad_(False)
        self.conv = isinstance(base, nn.Conv2d)
        if self.conv:
            # [RECALL] lora_A = Conv2d(in, r, kernel_size, stride, padding) of the base conv, lora_B = Conv2d(r, out, 1):
            # SANA's targets are 1x1; PixArt's PatchEmbed projection is k = s = 2 (lora_A held as [r, in*k*k], reshaped on use)
            assert base.kernel_size == base.stride and base.padding == (0, 0)
            self.k = base.kernel_size[0]
            out_dim, in_dim = base.out_channels, base.in_channels * self.k * self.k
        else:
            out_dim, in_dim = base.out_features, base.in_features
        dt = base.weight.dtype
        self.r, self.scaling = r, alpha / r
        self.lora_A = nn.Parameter(torch.empty(r, in_dim, dtype=dt))
        self.lora_B = nn.Parameter(torch.zeros(out_dim, r, dtype=dt))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()

    def forward(self, x):
        result = self.base_layer(x)
        xin = self.dropout(x.to(self.lora_A.dtype))
        if self.conv:
            a4 = self.lora_A.view(self.r, -1, self.k, self.k)
            u = F.conv2d(F.conv2d(xin, a4, stride=self.k), self.lora_B[:, :, None, None])
        else:
            u = F.linear(F.linear(xin, self.lora_A), self.lora_B)
        return result + u * self.scaling


def apply_lora(model: nn.Module, targets, r: int, alpha: float, dropout: float = 0.0):
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
        w = LoRAWrapped(mod, r, alpha, dropout)
        if parts[-1].isdigit():
            parent[int(parts[-1])] = w
        else:
            setattr(parent, parts[-1], w)
        wrapped[name] = w
    return wrapped
