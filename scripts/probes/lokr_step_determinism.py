#!/usr/bin/env python3
"""Round 6: where does the B = 32 LoKr step in the pre_add form (pair=False) stop being run-to-run deterministic?  A few blocks at
the real width, the same step several times, every kept activation of every block compared bit for bit with the first run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yat_amd.sana import SanaConfig, SanaTransformer2DModelHIP
from yat_amd.lokr import LoKrAdapters
from yat_amd.recipe import SanaRecipe
BF, DEV = torch.bfloat16, "cuda"
L, B = int(os.environ.get("LAYERS", 4)), int(os.environ.get("B", 32))
pair = os.environ.get("PAIR", "0") != "0"
hip = SanaTransformer2DModelHIP(SanaConfig(num_layers=L), device=DEV).init_synthetic(7)
ad = LoKrAdapters(hip, ["conv_inverted", "conv_point", "to_q", "to_k", "to_v", "to_out.0", "linear_1", "linear_2", "proj"], r=8,
                  alpha=8.0, module_dropout=float(os.environ.get("DROP", 0.05)), pair=pair)
g = torch.Generator().manual_seed(78)
for e in ad.entries:
    w1 = ad._views(e, ad.flat_param)[0]
    w1.copy_((torch.randn(w1.shape, generator=g) * 0.05).to(BF))
cfg = hip.cfg
lens = [int(x) for x in torch.randint(20, 301, (B,), generator=g)]
latents = (torch.randn(B, cfg.in_channels, 32, 32, generator=g) * 0.5).to(BF)
embs = [torch.randn(n, cfg.caption_channels, generator=g).to(BF) for n in lens]
recipe = SanaRecipe(hip, pad_to=512, device=DEV)
hip.train()
NAMES = ("h1", "qkv", "attn", "lin1", "x1", "q2", "kv2", "o2", "x2", "h2", "z", "s", "y", "lin3", "x3")
first = None
for rep in range(int(os.environ.get("REPS", 4))):
    torch.manual_seed(1234)
    ad.flat_grad.zero_()
    loss, pred, _ = recipe.optimize(latents, embs, torch.Generator().manual_seed(5), return_pred=True)
    S = hip._saved
    snap = {"x0": S.blocks[0].x_in.clone(), "encn": S.encn.clone()}
    for i, A in enumerate(S.blocks):
        for n in NAMES:
            t = getattr(A, n, None)
            if t is not None:
                snap[f"b{i}.{n}"] = t.clone()
    snap["pred"] = pred.detach().clone()
    loss.backward()
    torch.cuda.synchronize()
    snap["grads"] = ad.flat_grad.clone()
    if first is None:
        first = snap
        print(f"run 0: loss {loss.item():.7f}", flush=True)
        continue
    bad = [k for k in snap if not torch.equal(snap[k], first[k])]
    print(f"run {rep}: loss {loss.item():.7f}; differing from run 0: {bad[:12]} ({len(bad)} of {len(snap)})", flush=True)
    for k in bad[:3]:
        d = (snap[k].float() - first[k].float()).abs()
        if d.dim() == 2:
            rows, cols = d.amax(1).nonzero().flatten(), d.amax(0).nonzero().flatten()
            print(f"   {k}: shape {tuple(d.shape)} rows {rows.numel()} [{rows[:4].tolist()}..{rows[-2:].tolist()}] cols {cols.numel()} "
                  f"[{cols[:4].tolist()}..{cols[-2:].tolist()}] max {d.max().item():.3e}", flush=True)
