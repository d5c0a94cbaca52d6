"""Host-side logic that needs no GPU: schedule, LR warm-up, shard format, bucket sampler, model parameter layout,
FLOP accounting."""
import os

import pytest
import torch

from yat_amd.common.aspect_ratios import ASPECT_RATIO_1024_BIN, ASPECT_RATIO_512_BIN, table_for_resolution
from yat_amd.common.bucket_sampler import BucketSampler
from yat_amd.common.shards import read_shard, write_shard
from yat_amd.common.trainer import HipAccelerator, WarmupLR
from yat_amd.scheduler import FlowMatchSchedule


def test_schedule_equals_oracle_tables():
    from oracle.recipe_ref import FlowMatchSchedule as Ref, draw_recipe_randoms
    a, b = FlowMatchSchedule(shift=3.0), Ref(shift=3.0)
    assert torch.equal(a.sigmas, b.sigmas) and torch.equal(a.timesteps, b.timesteps)
    # same draws as the oracle when the noise is drawn first (reference order)
    g1, g2 = torch.Generator(), torch.Generator()
    torch.randn((4, 8, 4, 4), generator=g1, dtype=torch.bfloat16)
    idx, t, sig = a.sample(4, g1)
    _, ridx, rt, rsig = draw_recipe_randoms((4, 8, 4, 4), 4, b, g2)
    assert torch.equal(idx, ridx) and torch.equal(t, rt) and torch.equal(sig, rsig)


def test_warmup_matches_torch_lambdalr():
    class Opt:
        param_groups = [dict(lr=1e-3, initial_lr=1e-3)]
    w = WarmupLR(Opt(), 4)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: s / 4 if s < 4 else 1.0)
    for _ in range(7):
        assert abs(w.get_last_lr()[0] - ref.get_last_lr()[0]) < 1e-12
        opt.step()
        ref.step()
        w.step()


def test_aspect_tables():
    assert ASPECT_RATIO_1024_BIN["1.0"] == [1024.0, 1024.0] and len(ASPECT_RATIO_1024_BIN) == 33
    assert ASPECT_RATIO_512_BIN["0.25"] == [256.0, 1024.0]
    assert table_for_resolution(1024) is ASPECT_RATIO_1024_BIN
    for k, (h, w) in ASPECT_RATIO_1024_BIN.items():
        assert h % 32 == 0 and w % 32 == 0 and abs(h / w - float(k)) < 0.02


def _make_shards(tmp_path, n_shards=2, per=24, seed=0):
    g = torch.Generator().manual_seed(seed)
    ratios = ["1.0", "0.5", "2.0"]
    paths = []
    for s in range(n_shards):
        samples = []
        for i in range(per):
            r = ratios[(i + s) % 3]
            H, W = ASPECT_RATIO_1024_BIN[r]
            L = int(torch.randint(1, 20, (1,), generator=g))
            samples.append(dict(__key__=f"{s:03d}{i:05d}", ratio=r,
                                latent=torch.randn(8, int(H) // 256, int(W) // 256, generator=g).to(torch.bfloat16),
                                emb=torch.randn(L, 16, generator=g).to(torch.bfloat16)))
        p = str(tmp_path / f"shard-{s:06d}.tar")
        write_shard(p, samples)
        paths.append(p)
    return paths


def test_shard_roundtrip(tmp_path):
    paths = _make_shards(tmp_path, 1, 5)
    out = list(read_shard(paths[0]))
    assert len(out) == 5
    assert out[0]["ratio"] in (1.0, 0.5, 2.0) and out[0]["latent.pt"].dtype == torch.bfloat16
    assert out[0]["emb.pt"].shape[1] == 16


def test_legacy_cache_roundtrip_and_sampler(tmp_path):
    """cache/{idx}.npy tuples of the reference (common/cache.py:70-85): (ratio, latent, (emb padded to 300, mask))."""
    from yat_amd.common.shards import write_legacy_sample, read_legacy_sample, iter_legacy_cache
    g = torch.Generator().manual_seed(1)
    cache = tmp_path / "cache"
    cache.mkdir()
    want = []
    for i in range(10):
        r = ["1.0", "0.5"][i % 2]
        H, W = ASPECT_RATIO_1024_BIN[r]
        lat = torch.randn(8, int(H) // 256, int(W) // 256, generator=g).to(torch.bfloat16)
        emb = torch.randn(3 + i, 16, generator=g).to(torch.bfloat16)
        write_legacy_sample(str(cache / f"{i}.npy"), float(r), lat, emb, compress=(i % 3 == 0))
        want.append((float(r), lat, emb))
    # the file is exactly the tuple the reference's loader expects
    ratio, latent, (padded, mask) = torch.load(str(cache / "1.npy"), weights_only=False)
    assert padded.shape == (300, 16) and mask.shape == (300,) and int(mask.sum()) == 4 and ratio == 0.5
    assert torch.equal(padded[:4], want[1][2]) and padded[4:].abs().max() == 0
    got = list(iter_legacy_cache(str(cache)))
    assert [s["__key__"] for s in got] == [str(i) for i in range(10)]
    for s, (r, lat, emb) in zip(got, want):
        assert s["ratio"] == r and torch.equal(s["latent.pt"], lat) and torch.equal(s["emb.pt"], emb)
    assert [s["__key__"] for s in iter_legacy_cache(str(cache), rank=1, world=2)] == ["1", "3", "5", "7", "9"]
    assert torch.equal(read_legacy_sample(str(cache / "0.npy"))["emb.pt"], want[0][2])          # gzip variant
    model = type("M", (), {"aspect_ratios": ASPECT_RATIO_1024_BIN})()
    s = BucketSampler([], HipAccelerator(1, device="cpu"), batch_size=2, model=model, seed=0, legacy_cache_dir=str(cache))
    b = next(iter(s))
    assert b.vae_features.shape[0] == 2 and len(b.embeddings) == 2 and b.ratio in (1.0, 0.5)


def test_bucket_sampler_single_process(tmp_path):
    paths = _make_shards(tmp_path)
    model = type("M", (), {"aspect_ratios": ASPECT_RATIO_1024_BIN})()
    acc = HipAccelerator(1, device="cpu")
    s = BucketSampler([], acc, batch_size=4, model=model, seed=3, local_paths=paths)
    it = iter(s)
    seen = set()
    for _ in range(12):
        b = next(it)
        assert b.vae_features.shape[0] == 4 and len(b.embeddings) == 4
        H, W = ASPECT_RATIO_1024_BIN[str(b.ratio)]
        assert tuple(b.vae_features.shape[2:]) == (int(H) // 256, int(W) // 256)      # one ratio per batch
        seen.add(b.ratio)
    assert seen == {1.0, 0.5, 2.0}
    # deterministic for a given seed
    s2 = BucketSampler([], acc, batch_size=4, model=model, seed=3, local_paths=paths)
    b1, b2 = next(iter(BucketSampler([], acc, 4, model=model, seed=3, local_paths=paths))), next(iter(s2))
    assert b1.ratio == b2.ratio and torch.equal(b1.vae_features, b2.vae_features)


def test_model_parameter_layout_on_cpu():
    """Construction, state-dict keys and bucket bounds need no GPU (compute does)."""
    from oracle.sana_ref import SanaConfig as RC, SanaTransformerRef
    from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
    rc = RC.tiny()
    ref = SanaTransformerRef(rc)
    m = SanaTransformer2DModelHIP(SanaConfig(**{k: getattr(rc, k) for k in SanaConfig.__dataclass_fields__}), device="cpu")
    sd, rsd = m.state_dict(), ref.state_dict()
    assert set(sd) == set(rsd)
    assert all(sd[k].shape == rsd[k].shape for k in sd)
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in rsd.items()})
    assert torch.equal(m.state_dict()["proj_out.bias"], rsd["proj_out.bias"].to(torch.bfloat16))
    # to_q | to_k | to_v are contiguous in the flat buffer (one fused QKV GEMM)
    D = rc.inner_dim
    w, _ = m._fused("transformer_blocks.0.attn1.to_q.weight", 3 * D, D)
    assert torch.equal(w[D:2 * D], m.P["transformer_blocks.0.attn1.to_k.weight"])
    assert m.bucket_bounds[0][0] == 0 and m.bucket_bounds[-1][1] == m.numel_flat
    assert all(a[1] == b[0] for a, b in zip(m.bucket_bounds, m.bucket_bounds[1:]))
    assert all(p.grad is not None and p.grad.data_ptr() >= m.flat_grad.data_ptr() for p in m.parameters())
    with pytest.raises(Exception):            # the product path fails loudly without a GPU: no CPU fallback
        m(torch.zeros(1, 8, 4, 4), encoder_hidden_states=torch.zeros(1, 8, 96), timestep=torch.zeros(1))


def test_flop_accounting_matches_survey():
    import bench
    from yat_amd.sana import SanaConfig
    f = bench.train_flops_per_image(SanaConfig(), 1024, 512)
    assert abs(f / 9.285e12 - 1) < 2e-3                 # 9.285 TFLOP / image


def test_empty_embedding_layout_is_checked_per_recipe():
    """CFG dropout (common/trainer.py:306-308,319-323) substitutes ``empty_embeddings[0]`` per sample: SANA / PixArt expect a
    list with one [L, C] tensor, SD3.5 a (prompt [T, C], pooled [P]) pair -- a file in the wrong layout is refused by name
    instead of being sliced by rows."""
    import pytest
    from train_sd35 import SD35Trainer
    from yat_amd.common.trainer import Model
    t = torch.randn(5, 8)
    assert Model.check_empty_embeddings(None, t, "f.pt")[0] is t
    assert Model.check_empty_embeddings(None, [t], "f.pt")[0] is t
    with pytest.raises(ValueError, match="f.pt"):
        Model.check_empty_embeddings(None, {"x": 1}, "f.pt")
    pair = (torch.randn(333, 16), torch.randn(1, 32))
    for given in (pair, [pair], (pair[0][None], pair[1])):
        got = SD35Trainer.check_empty_embeddings(None, given, "e.pt")
        assert len(got) == 1 and got[0][0].shape == (333, 16) and got[0][1].shape == (32,)
    with pytest.raises(ValueError, match="e.pt"):
        SD35Trainer.check_empty_embeddings(None, t, "e.pt")
    with pytest.raises(ValueError, match="e.pt"):
        SD35Trainer.check_empty_embeddings(None, [t], "e.pt")


def test_host_draws_are_remembered_per_generator_state():
    """SanaRecipe._draw_cached: the reference hands every step a fresh torch.Generator() (common/trainer.py:325), i.e. always
    the same state; a generator arriving in a remembered state gets the remembered noise / timesteps and is left in the
    remembered end state -- exactly what drawing again does; any other state draws."""
    from yat_amd.recipe import SanaRecipe
    from yat_amd.scheduler import FlowMatchSchedule
    r = SanaRecipe.__new__(SanaRecipe)
    r.scheduler = FlowMatchSchedule()
    shape, B = (4, 8, 6, 10), 4

    def draw(gen):
        out = torch.empty(shape, dtype=torch.bfloat16)
        t, s = r._draw_cached(shape, B, gen, out)
        return out, t, s, torch.rand(3, generator=gen)          # the last item: where the generator was left

    g = torch.Generator()
    want_n = torch.randn(shape, generator=g, dtype=torch.bfloat16)
    _, want_t, want_s = r.scheduler.sample(B, g)
    want_next = torch.rand(3, generator=g)
    for _ in range(3):                                           # first call draws, the others are remembered
        n, t, s, nxt = draw(torch.Generator())
        assert torch.equal(n, want_n) and torch.equal(t, want_t) and torch.equal(s, want_s) and torch.equal(nxt, want_next)
    a, b, a2 = draw(torch.Generator().manual_seed(3)), draw(torch.Generator().manual_seed(4)), draw(torch.Generator().manual_seed(3))
    assert not torch.equal(a[0], b[0]) and torch.equal(a[0], a2[0]) and torch.equal(a[3], a2[3])
    g = torch.Generator()                                        # an advancing generator (exploration steps): no false hits
    x, y = draw(g), draw(g)
    assert not torch.equal(x[0], y[0])


def test_host_thread_cap():
    from yat_amd.common import host
    n = host.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        assert host.cap_host_threads() <= max(1, min(before, n, 16)) or "OMP_NUM_THREADS" in os.environ
    finally:
        torch.set_num_threads(before)


def test_carried_loss_arithmetic_in_bf16():
    """yat_amd/ddp.py on_loss / _harvest_loss: what survives of the logged loss when the slots and the reduction are bf16 --
    eight ranks, losses around 2 (where bf16 alone resolves 0.016), difference to last step's mean in a head + remainder pair."""
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    prev_mean = torch.tensor(2.031)
    losses = 2.0 + 0.2 * torch.randn(8, generator=g)
    y = losses - prev_mean
    hi = y.to(BF)
    lo = (y - hi.float()).to(BF)
    acc_hi, acc_lo = torch.zeros((), dtype=BF), torch.zeros((), dtype=BF)
    for r in range(8):                                   # a ring's running sums, rounded to bf16 at every hop
        acc_hi, acc_lo = (acc_hi.float() + hi[r].float()).to(BF), (acc_lo.float() + lo[r].float()).to(BF)
    got = prev_mean + (acc_hi.float() / 8).to(BF).float() + (acc_lo.float() / 8).to(BF).float()
    want = losses.mean()
    plain = torch.zeros((), dtype=BF)
    for r in range(8):
        plain = (plain.float() + losses[r].to(BF).float()).to(BF)
    assert abs(got - want) < 2e-3 * want                 # 0.2 % of the loss
    assert abs(got - want) < 0.5 * abs((plain.float() / 8) - want) + 1e-4        # and better than carrying the bf16 loss itself


def test_rescale_adapter_scale_and_the_whitelist_call_sites():
    """common/trainer.py:270-281,385-397: `rescale_adapter_scale` is peft's context manager; the reference calls it bare, which
    rescales nothing.  The helper itself must scale and restore; the trainer's call sites keep the reference's effective
    behaviour by default and switch the adapter per timestep with YAT_ADAPTER_RESCALE=1."""
    import types
    from yat_amd.common import trainer as T
    ad = types.SimpleNamespace(scale=0.5, _gate=torch.full((4,), 0.5))
    with T.rescale_adapter_scale(ad, 0.0):
        assert ad.scale == 0.0 and torch.all(ad._gate == 0.0)
    assert ad.scale == 0.5 and torch.all(ad._gate == 0.5)
    with pytest.raises(TypeError):
        with T.rescale_adapter_scale(ad, "1"):
            pass
    with pytest.raises(ValueError):
        with T.rescale_adapter_scale(None, 1.0):
            pass
    m = types.SimpleNamespace(adapters=ad, timesteps=[0, 500], _rescale_live=False, _rescale_cm=None)
    m._set_adapter_scale = types.MethodType(T.Model._set_adapter_scale, m)
    m._set_adapter_scale(0.0)
    assert ad.scale == 0.5                                   # the reference's bare call: no effect
    m._rescale_live = True
    m._set_adapter_scale(0.0)
    assert ad.scale == 0.0 and torch.all(ad._gate == 0.0)
    m._set_adapter_scale(1.0)
    assert ad.scale == 0.5 and m._rescale_cm is None
    m._set_adapter_scale(0.0)
    m._set_adapter_scale(0.0)                                # twice in a row: still the trained scaling times zero, restorable
    m._set_adapter_scale(1.0)
    assert ad.scale == 0.5 and torch.all(ad._gate == 0.5)


def test_legacy_cache_files_written_by_the_reference(tmp_path):
    """The two ``cache/{idx}.npy`` files under tests/golden/legacy_cache/ were written by the REFERENCE's own
    CacheLoadFeatures.run (common/cache.py:54-85; tests/golden/make_legacy_cache_golden.py) -- a 7-row prompt and one that fills
    all 300 rows.  This repo's reader must hand back exactly the inputs (tests/golden/legacy_cache_inputs.py), and its writer
    must produce the tuple layout the reference produced."""
    import gzip
    import shutil
    import sys
    from yat_amd.common.shards import iter_legacy_cache, read_legacy_sample, write_legacy_sample
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    try:
        from legacy_cache_inputs import sample
    finally:
        sys.path.remove(gold)
    cache = tmp_path / "cache"
    cache.mkdir()
    for idx in (0, 1):
        with gzip.open(os.path.join(gold, "legacy_cache", f"{idx}.npy.gz"), "rb") as f, open(cache / f"{idx}.npy", "wb") as g:
            shutil.copyfileobj(f, g)
    got = list(iter_legacy_cache(str(cache)))
    assert [s["__key__"] for s in got] == ["0", "1"]
    for idx, s in enumerate(got):
        ratio, lat, rows = sample(idx)
        emb = torch.cat(rows)                                                    # [L, 2304] bf16, small integers
        assert s["ratio"] == pytest.approx(float(ratio)) and isinstance(s["ratio"], float)
        assert s["latent.pt"].dtype == torch.bfloat16 and torch.equal(s["latent.pt"], lat[0])
        assert s["emb.pt"].shape == emb.shape and torch.equal(s["emb.pt"].to(torch.bfloat16), emb)
        # what the reference wrote, field by field: fp32 embeddings zero-padded to 300 rows, a float mask, the squeezed latent
        r_ref, l_ref, (e_ref, m_ref) = torch.load(str(cache / f"{idx}.npy"), weights_only=False)
        assert e_ref.shape == (300, 2304) and e_ref.dtype == torch.float32 and m_ref.shape == (300,) and m_ref.dtype == torch.float32
        assert int(m_ref.sum()) == emb.shape[0] and l_ref.shape == lat.shape[1:]
        # ... and this repo's writer produces the same tuple for the same sample (padding, mask, latent; embeddings in the
        # dtype it is handed: the reference's zero buffer makes them fp32)
        write_legacy_sample(str(tmp_path / "mine.npy"), float(ratio), lat[0], emb.float())
        r2, l2, (e2, m2) = torch.load(str(tmp_path / "mine.npy"), weights_only=False)
        assert torch.equal(e2, e_ref) and torch.equal(m2, m_ref) and torch.equal(l2, l_ref) and float(r2) == pytest.approx(float(r_ref))
    assert read_legacy_sample(str(cache / "1.npy"))["emb.pt"].shape[0] == 300


def test_reference_nccl_environment_is_not_inherited():
    """utils/set_nccl_vars.py exports six NCCL_* variables (tests/golden/nccl_vars.json: produced by importing the reference's
    module); yat_amd/ddp.py states for each why this build does not inherit it, and nothing in the package sets one."""
    import json
    import subprocess
    import sys
    from yat_amd.ddp import REFERENCE_NCCL_ENV_NOT_INHERITED as table
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "nccl_vars.json")) as f:
        ref = json.load(f)
    assert set(ref) == set(table) and ref["NCCL_P2P_DISABLE"] == "1"
    code = ("import os, sys; sys.path.insert(0, %r); import yat_amd.ddp, yat_amd.common.trainer as T; "
            "T.HipAccelerator(1, device='cpu'); print(sorted(k for k in os.environ if k.startswith('NCCL_')))" % root)
    env = {k: v for k, v in os.environ.items() if not k.startswith("NCCL_")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip().splitlines()[-1] == "[]"


def test_default_transport_and_group_backend(monkeypatch):
    """yat_amd/ddp.py: a multi-rank job sends its gradient buckets through torch.distributed's group (nccl on a GPU, gloo on
    the CPU) unless YAT_COMM=native asks for the library's own communicator -- never run at N > 1, so opt-in (round-4
    advisor) -- in which case the launcher-level group is gloo (one RCCL communicator per process); YAT_DIST_BACKEND wins."""
    from yat_amd import ddp
    for k in ("YAT_COMM", "YAT_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert ddp.default_transport() == "torch" and ddp.group_backend(on_gpu=False) == "gloo"
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    assert ddp.default_transport() == "torch" and ddp.group_backend() == "nccl"
    monkeypatch.setenv("YAT_COMM", "native")
    assert ddp.default_transport() == "native" and ddp.group_backend() == "gloo"
    monkeypatch.setenv("YAT_COMM", "torch")
    assert ddp.default_transport() == "torch" and ddp.group_backend() == "nccl"
    monkeypatch.delenv("YAT_COMM")
    monkeypatch.setenv("YAT_DIST_BACKEND", "gloo")                 # several ranks on one GPU: RCCL cannot be used at all
    assert ddp.default_transport() == "torch" and ddp.group_backend() == "gloo"
    monkeypatch.setenv("YAT_COMM", "native")
    monkeypatch.setenv("YAT_DIST_BACKEND", "nccl")
    assert ddp.default_transport() == "native" and ddp.group_backend() == "nccl"


def test_rccl_channel_policy(monkeypatch):
    """yat_amd/ddp.py apply_channel_policy (round-5 advisor): RCCL is left alone by default -- as the reference leaves it --;
    a cap is opt-in (argument or YAT_RCCL_CHANNELS); a site's own NCCL_MIN/MAX_NCHANNELS are never rewritten, a request that
    contradicts them raises."""
    import pytest
    from yat_amd import ddp
    for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "YAT_RCCL_CHANNELS"):
        monkeypatch.delenv(k, raising=False)
    assert ddp.apply_channel_policy(1) is None and "NCCL_MAX_NCHANNELS" not in os.environ
    assert ddp.apply_channel_policy(8) is None and "NCCL_MAX_NCHANNELS" not in os.environ            # default: RCCL's own
    assert ddp.apply_channel_policy(8, 0) is None and "NCCL_MAX_NCHANNELS" not in os.environ
    monkeypatch.setenv("YAT_RCCL_CHANNELS", str(ddp.RCCL_CHANNEL_CAP))                               # opt-in by environment
    assert ddp.apply_channel_policy(8) == ddp.RCCL_CHANNEL_CAP and os.environ["NCCL_MAX_NCHANNELS"] == "24"
    monkeypatch.delenv("YAT_RCCL_CHANNELS")
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "40")                                                   # the site's own choice
    assert ddp.apply_channel_policy(8) == 40 and os.environ["NCCL_MAX_NCHANNELS"] == "40"
    assert ddp.apply_channel_policy(8, 40) == 40
    with pytest.raises(ValueError, match="NCCL_MAX_NCHANNELS=40"):
        ddp.apply_channel_policy(8, 8)
    assert os.environ["NCCL_MAX_NCHANNELS"] == "40"                                                  # untouched
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "32")
    with pytest.raises(ValueError, match="NCCL_MIN_NCHANNELS=32"):
        ddp.apply_channel_policy(8, 8)
    assert os.environ["NCCL_MIN_NCHANNELS"] == "32" and "NCCL_MAX_NCHANNELS" not in os.environ       # untouched
    assert ddp.apply_channel_policy(8, 48) == 48 and os.environ["NCCL_MIN_NCHANNELS"] == "32"
