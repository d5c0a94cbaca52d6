"""Oracle self-checks (CPU): the restatement against the known answers that exist for this path.
The reference has no tests or fixtures, so these pin what can be pinned: scheduler table
values quoted in SURVEY.md 8(c), the default-seed RNG stream of stock torch, analytic identities of each op."""
import math

import torch
import torch.nn.functional as F

from oracle.recipe_ref import FlowMatchSchedule, draw_recipe_randoms, optimize_ref, pad_embeddings
from oracle.sana_ref import SanaConfig, SanaTransformerRef, init_like_pretrained, timestep_sinusoid, RMSNorm


def test_flow_match_table_known_answers():
    s = FlowMatchSchedule(shift=3.0)
    for i, v in ((0, 1.0), (1, 0.99966639), (500, 0.75), (999, 0.00299401)):
        assert abs(s.sigmas[i].item() - v) < 1e-7
    assert abs(s.timesteps[999].item() - 2.99401212) < 1e-5
    assert s.sigmas.dtype == torch.float32 and s.sigmas.shape == (1000,)


def test_fresh_generator_stream_is_constant():
    """trainer.py:325 creates an unseeded torch.Generator() per step: same default seed, same draws (App. B-1)."""
    assert torch.Generator().initial_seed() == 67280421310721
    s = FlowMatchSchedule()
    a = draw_recipe_randoms((8, 32, 32, 32), 8, s, torch.Generator())
    b = draw_recipe_randoms((8, 32, 32, 32), 8, s, torch.Generator())
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a[1].tolist() == [118, 742, 713, 109, 512, 401, 795, 302]          # golden indices (noise drawn first)
    assert a[0].dtype == torch.bfloat16 and a[3].dtype == torch.bfloat16
    assert torch.equal(a[2], s.timesteps[a[1]])


def test_param_count_is_1p6b():
    from yat_amd.sana import _param_specs, SanaConfig as C
    assert sum(math.prod(s) for _, s in _param_specs(C())) == 1_604_462_752


def test_linear_attention_matches_explicit_formula():
    cfg = SanaConfig.tiny()
    m = SanaTransformerRef(cfg)
    init_like_pretrained(m, 1)
    attn = m.transformer_blocks[0].attn1
    x = torch.randn(2, 12, cfg.inner_dim)
    q = F.relu(attn.to_q(x)).view(2, 12, 2, 32)
    k = F.relu(attn.to_k(x)).view(2, 12, 2, 32)
    v = attn.to_v(x).view(2, 12, 2, 32)
    w = torch.einsum("bihc,bjhc->bhij", q, k)                     # relu(q_i) . relu(k_j)
    o = torch.einsum("bhij,bjhc->bihc", w, v) / (w.sum(-1).permute(0, 2, 1)[..., None] + 1e-15)
    ref = attn.to_out[0](o.reshape(2, 12, -1))
    assert torch.allclose(attn.linear_attention(x), ref, rtol=1e-4, atol=1e-5)


def test_masked_tail_equals_truncated_sequence():
    """SDPA with a -10000 bias on the padded tail == attention over the unpadded keys (fp32)."""
    cfg = SanaConfig.tiny()
    m = SanaTransformerRef(cfg)
    init_like_pretrained(m, 2)
    a2 = m.transformer_blocks[0].attn2
    x, enc = torch.randn(1, 10, cfg.inner_dim), torch.randn(1, 8, cfg.inner_dim)
    bias = torch.zeros(1, 1, 8)
    bias[..., 5:] = -10000.0
    assert torch.allclose(a2(x, enc, bias), a2(x, enc[:, :5], None), rtol=1e-4, atol=1e-5)


def test_norms_and_embeddings_known_answers():
    x = torch.full((2, 16), 3.0)
    assert torch.allclose(F.layer_norm(x, (16,), None, None, 1e-6), torch.zeros(2, 16), atol=1e-3)
    r = RMSNorm(16, eps=1e-5)
    assert torch.allclose(r(x), torch.ones(2, 16), atol=1e-5)
    e = timestep_sinusoid(torch.tensor([0.0, 1000.0]))
    assert e.shape == (2, 256)
    assert torch.allclose(e[0, :128], torch.ones(128)) and torch.allclose(e[0, 128:], torch.zeros(128))   # cos first
    assert abs(e[1, 0].item() - math.cos(1000.0)) < 1e-4


def test_pad_embeddings_and_optimize_shapes():
    embs = [torch.randn(3, 96), torch.randn(7, 96)]
    enc, mask = pad_embeddings(embs, 8)
    assert enc.shape == (2, 8, 96) and mask.dtype == torch.long
    assert mask.tolist() == [[1, 1, 1, 0, 0, 0, 0, 0], [1] * 7 + [0]]
    assert enc[0, 3:].abs().max() == 0
    cfg = SanaConfig.tiny()
    m = SanaTransformerRef(cfg)
    init_like_pretrained(m, 0)
    lat = torch.randn(2, 8, 4, 6) * 0.5
    l32, p32, t32 = optimize_ref(m, FlowMatchSchedule(), lat, embs, torch.Generator(), 8, torch.float32)
    mb = SanaTransformerRef(cfg)
    mb.load_state_dict(m.state_dict())
    lb, pb, _ = optimize_ref(mb.to(torch.bfloat16), FlowMatchSchedule(), lat, embs, torch.Generator(), 8, torch.bfloat16)
    assert p32.shape == lat.shape and pb.dtype == torch.bfloat16
    assert abs(lb.item() - l32.item()) < 2e-2 * l32.item()
    assert ((pb.float() - p32).norm() / p32.norm()).item() < 3e-2
