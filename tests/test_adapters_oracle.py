"""CPU checks of the adapter restatements (oracle/lora_ref.py, oracle/lokr_ref.py): what the wraps must satisfy by construction."""
import copy

import torch

from oracle.sana_ref import SanaConfig, SanaTransformerRef, init_like_pretrained
from oracle.lora_ref import apply_lora, LoRAWrapped
from oracle.lokr_ref import apply_lokr, factorization

TARGETS = ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"]


def _model():
    m = SanaTransformerRef(SanaConfig.tiny(num_layers=1))
    init_like_pretrained(m, 0)
    return m


def _inputs(cfg):
    g = torch.Generator().manual_seed(0)
    return (torch.randn(2, cfg.in_channels, 4, 4, generator=g), torch.randn(2, 8, cfg.caption_channels, generator=g),
            torch.tensor([500.0, 20.0]), torch.ones(2, 8, dtype=torch.long))


def test_fresh_adapters_are_the_identity_and_only_adapters_train():
    base = _model()
    x = _inputs(base.cfg)
    want = base(*x)
    for wrap, zero_name in ((apply_lora, "lora_B"), (apply_lokr, "lokr_w1")):
        m = copy.deepcopy(base)
        wrapped = wrap(m, TARGETS, r=2, alpha=4.0)
        assert len(wrapped) == 15 and all((getattr(w, zero_name) == 0).all() for w in wrapped.values())
        got = m(*x)
        assert torch.equal(got, want)                       # zero-initialised factor: the wrap changes nothing
        got.square().mean().backward()
        trainable = [n for n, p in m.named_parameters() if p.requires_grad]
        assert trainable and all(("lora_" in n or "lokr_" in n) for n in trainable)
        assert all(p.grad is None for n, p in m.named_parameters() if not p.requires_grad)


def test_lora_term_is_the_low_rank_product():
    lin = torch.nn.Linear(24, 16)
    w = LoRAWrapped(lin, r=4, alpha=8.0)
    with torch.no_grad():
        w.lora_B.normal_()
    x = torch.randn(5, 24)
    want = lin(x) + (x @ w.lora_A.T @ w.lora_B.T) * 2.0
    assert torch.allclose(w(x), want, atol=1e-6)
    conv = torch.nn.Conv2d(3, 8, kernel_size=2, stride=2)    # PixArt's PatchEmbed projection as a target
    wc = LoRAWrapped(conv, r=2, alpha=2.0)
    with torch.no_grad():
        wc.lora_B.normal_()
    img = torch.randn(1, 3, 4, 6)
    rows = torch.nn.functional.unfold(img, 2, stride=2).transpose(1, 2)                   # [1, 6, 12] patches
    want = conv(img) + (rows @ wc.lora_A.T @ wc.lora_B.T).transpose(1, 2).reshape(1, 8, 2, 3)
    assert torch.allclose(wc(img), want, atol=1e-5)


def test_factorization_known_answers():
    assert factorization(2240) == (40, 56) and factorization(11200) == (100, 112) and factorization(1152) == (32, 36)
    assert factorization(7) == (1, 7) and factorization(64) == (8, 8)
