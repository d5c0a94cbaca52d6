"""CPU checks for the SD3.5 path: the oracle runs and is self-consistent, the host-side position-table crop equals the
oracle's crop of the full table, the shard format carries the pooled projection."""
import torch

BF = torch.bfloat16


def test_oracle_shapes_keys_and_fp32_bf16_agree():
    from oracle.sd3_ref import SD3Config, SD3TransformerRef, init_like_pretrained, optimize_ref
    from oracle.recipe_ref import FlowMatchSchedule
    cfg = SD3Config.tiny()
    m = SD3TransformerRef(cfg)
    init_like_pretrained(m, 0)
    keys = set(m.state_dict())
    assert "transformer_blocks.0.attn2.norm_q.weight" in keys and "transformer_blocks.2.attn2.to_q.weight" not in keys
    assert "transformer_blocks.2.attn.to_add_out.weight" not in keys and "transformer_blocks.2.ff_context.net.2.weight" not in keys
    assert m.state_dict()["transformer_blocks.0.norm1.linear.weight"].shape[0] == 9 * cfg.inner_dim
    assert m.state_dict()["transformer_blocks.2.norm1_context.linear.weight"].shape[0] == 2 * cfg.inner_dim
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 8, 12, 8, generator=g)
    pe, pool = torch.randn(2, 10, 96, generator=g), torch.randn(2, 64, generator=g)
    mb = SD3TransformerRef(cfg)
    mb.load_state_dict(m.state_dict())
    mb = mb.to(BF)
    m32 = SD3TransformerRef(cfg)
    m32.load_state_dict(mb.state_dict())
    m32 = m32.float()
    lb, pb, _ = optimize_ref(mb, FlowMatchSchedule(), lat, pe, pool, torch.Generator().manual_seed(1), BF)
    lt, pt, _ = optimize_ref(m32, FlowMatchSchedule(), lat, pe, pool, torch.Generator().manual_seed(1), torch.float32)
    assert pb.shape == lat.shape and pb.dtype == BF
    assert ((pb.float() - pt).norm() / pt.norm()).item() < 3e-2 and abs(lb.item() - lt.item()) < 3e-2 * lt.item()


def test_position_table_crop_matches_the_full_table():
    from oracle.sd3_ref import SD3Config, PatchEmbedMax
    from yat_amd.sd3 import sincos_crop
    cfg = SD3Config.tiny(pos_embed_max_size=20, sample_size=16)
    pe = PatchEmbedMax(cfg)
    for h, w in ((6, 4), (20, 20), (3, 17)):
        top, left = (20 - h) // 2, (20 - w) // 2
        mine = sincos_crop(cfg.inner_dim, 20, cfg.sample_size // cfg.patch_size, top, left, h, w)
        assert torch.equal(mine, pe.cropped(h, w)[0]), (h, w)


def test_shards_carry_the_pooled_projection(tmp_path):
    from yat_amd.common.shards import write_shard, read_shard
    g = torch.Generator().manual_seed(0)
    samples = [dict(__key__=f"{i:05d}", ratio="1.0", latent=torch.randn(4, 8, 8, generator=g).to(BF),
                    emb=torch.randn(5, 16, generator=g).to(BF), pooled=torch.randn(8, generator=g).to(BF)) for i in range(3)]
    p = str(tmp_path / "s.tar")
    write_shard(p, samples)
    back = list(read_shard(p))
    assert len(back) == 3
    for a, b in zip(samples, back):
        assert torch.equal(a["pooled"], b["pooled.pt"]) and torch.equal(a["emb"], b["emb.pt"])
