#!/usr/bin/env python3
"""Host-side cost of one SANA training step: Python launch path vs launch-plan replay (GPU idle when the clock starts)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yat_amd import ops
from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
from yat_amd.recipe import SanaRecipe
from yat_amd.optim import FlatAdamW

dev = torch.device("cuda", 0)
layers = int(os.environ.get("LAYERS", "20"))
model = SanaTransformer2DModelHIP(SanaConfig(num_layers=layers), device=dev).init_synthetic(0)
opt = FlatAdamW(model, lr=1e-5, overlap_update=True)
recipe = SanaRecipe(model, pad_to=512, device=dev)
B, T, Cc = 8, 512, 2304
g = torch.Generator(device=dev).manual_seed(0)
lat = (torch.randn(B, 32, 32, 32, generator=g, device=dev) * 0.5).bfloat16()
enc = torch.randn(B, T, Cc, generator=g, device=dev).bfloat16()
bias = torch.zeros(B, T, device=dev); kvl = torch.full((B,), 300, dtype=torch.int32, device=dev)
noise = torch.randn_like(lat); t = torch.full((B,), 500.0, device=dev); sig = torch.full((B,), 0.5, device=dev).bfloat16()
work = ops.kv_work_list([300] * B, T, dev)
loss = torch.zeros(1, device=dev)
for plans in (True, False):
    model.use_plans = plans
    for _ in range(4):
        recipe.train_step_device(lat, enc, (bias, kvl), noise, t, sig, loss, kv_work=work); opt.step()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        recipe.train_step_device(lat, enc, (bias, kvl), noise, t, sig, loss, kv_work=work)
        t1 = time.perf_counter(); opt.step(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    torch.cuda.synchronize()
    n = {k[0]: (v.n_entries, len(v.segments)) for k, v in model._plans.items()}
    print(f"plans={plans}: fwd+bwd host ms {[round(a, 2) for a, _ in ts]}, optimizer host ms {[round(b, 2) for _, b in ts]}; "
          f"plan entries {n}; replays {getattr(model, 'plan_replays', 0)}")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    model.use_plans = True
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    recipe.train_step_device(lat, enc, (bias, kvl), noise, t, sig, loss, kv_work=work)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
