"""Inputs of the legacy-cache fixtures (tests/golden/legacy_cache/*.npy.gz), shared by the script that ran the REFERENCE writer on
them (make_legacy_cache_golden.py) and by the test that holds this repo's reader to the files it wrote.  Values are small
integers (exact in bf16 / fp32, and the pickles compress to a few KB)."""
import torch

C = 2304          # hard-coded in the reference writer (common/cache.py:74)


def sample(idx):
    """-> (ratio, latent [1, 32, h, w] bf16, embedding: list of L rows [1, C] bf16) as the extractor hands them to
    CacheLoadFeatures.run (common/cache.py:70-73: ``ratio, latent, embedding = item[0]``; ``torch.stack(embedding)`` then
    ``swapaxes(0, 1)``)."""
    L, (h, w), ratio = [(7, (32, 32), 1.0), (300, (24, 42), 0.57)][idx]
    lat = ((torch.arange(32 * h * w).reshape(1, 32, h, w) * (3 + idx)) % 17 - 8).to(torch.bfloat16)
    rows = [(((torch.arange(C) + 5 * r + idx) % 13) - 6).to(torch.bfloat16).reshape(1, C) for r in range(L)]
    return torch.tensor([ratio]), lat, rows
