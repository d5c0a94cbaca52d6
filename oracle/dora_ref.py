"""Torch-CPU restatement of the PEFT DoRA wrap (``lora_algo: lora`` with ``lora_use_dora``; oracle, test-only).

Follows /root/reference/common/trainer.py:213-219,237-241: ``LoraConfig(r, lora_dropout, target_modules, lora_alpha,
use_dora=params.lora_use_dora)`` -> ``get_peft_model(model, config).to(dtype)`` (``lora_use_dora`` is true when the key is
present in the YAML, common/training_parameters_reader.py:142,192).  The arithmetic lives in the unpinned third-party ``peft``
(requirements.txt:15), absent from this container: everything below is [RECALL peft/tuners/lora/dora.py + layer.py, peft
0.11 - 0.13] -- PARITY UNPINNED for this module:

* on top of the LoRA pair (oracle/lora_ref.py) a magnitude vector ``m`` [out] (``lora_magnitude_vector``), initialised to
  ``||W + scaling lora_B lora_A||_2`` per output row (= the row norms of W, lora_B being zero), trainable;
* forward (DoraLinearLayer.forward, dropout 0 so ``base_result`` is handed over):
      lora_weight = lora_B(lora_A(eye)).T                                  (the product through the layers, module dtype)
      weight_norm = ||W + scaling * lora_weight||_2 per row, .to(W.dtype), DETACHED
      mag_norm_scale = (m / weight_norm).view(1, -1)
      result = base_layer(x) + (mag_norm_scale - 1) * (base_layer(x) - bias) + mag_norm_scale * lora_B(lora_A(x)) * scaling
  all op by op in the module dtype; a 1x1 Conv2d target is the same arithmetic over channels (DoraConv2dLayer);
* with lora_dropout > 0 peft recomputes the base product on the dropped input -- not restated (the HIP side refuses it).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lokr_ref import is_target


class DoRAWrapped(nn.Module):
    def __init__(self, base: nn.Module, r: int, alpha: float):
        super().__init__()
        self.base_layer = base
        for p in base.parameters():
            p.requires_grad_(False)
        self.conv = isinstance(base, nn.Conv2d)
        if self.conv:
            assert base.kernel_size == (1, 1), "DoRA restated for Linear and 1x1 convolutions"
            out_dim, in_dim = base.out_channels, base.in_channels
        else:
            out_dim, in_dim = base.out_features, base.in_features
        dt = base.weight.dtype
        self.r, self.scaling = r, alpha / r
        self.lora_A = nn.Parameter(torch.empty(r, in_dim, dtype=dt))
        self.lora_B = nn.Parameter(torch.zeros(out_dim, r, dtype=dt))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        w = base.weight.detach().reshape(out_dim, in_dim)
        self.magnitude = nn.Parameter(torch.linalg.norm(w.float(), dim=1).to(dt))       # lora_B = 0 at init

    def weight_norm(self):
        w = self.base_layer.weight.detach().reshape(self.lora_B.shape[0], -1)
        lora_weight = F.linear(F.linear(torch.eye(w.shape[1], dtype=w.dtype), self.lora_A), self.lora_B).T.detach()
        return torch.linalg.norm(w + self.scaling * lora_weight, dim=1).to(w.dtype)

    def forward(self, x):
        result = self.base_layer(x)
        s = (self.magnitude / self.weight_norm().detach())
        bias = self.base_layer.bias
        if self.conv:
            s4 = s.view(1, -1, 1, 1)
            lora = F.conv2d(F.conv2d(x, self.lora_A[:, :, None, None]), self.lora_B[:, :, None, None])
            base_result = result if bias is None else result - bias.view(1, -1, 1, 1)
            return result + (s4 - 1) * base_result + s4 * lora * self.scaling
        s2 = s.view(1, -1)
        lora = F.linear(F.linear(x, self.lora_A), self.lora_B)
        base_result = result if bias is None else result - bias
        return result + (s2 - 1) * base_result + s2 * lora * self.scaling


def apply_dora(model: nn.Module, targets, r: int, alpha: float):
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
        w = DoRAWrapped(mod, r, alpha)
        if parts[-1].isdigit():
            parent[int(parts[-1])] = w
        else:
            setattr(parent, parts[-1], w)
        wrapped[name] = w
    return wrapped
