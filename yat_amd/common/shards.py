"""Cached-feature shard format: reader and writer without the webdataset dependency.

Wire format = what the reference's FeaturesExtractor writes (common/features_extractor.py:83-88) and its
BucketSampler decodes (common/bucket_sampler.py:138-156): a POSIX tar whose members are grouped by key,
``<key>.ratio`` (ascii float), ``<key>.latent.pt`` (torch.save of the [C, h, w] bf16 latent) and ``<key>.emb.pt``
(torch.save of the UNPADDED [L, C] bf16 text embedding -- only mask-true rows, train_sana.py:92-94).
"""
from __future__ import annotations

import gzip
import io
import os
import tarfile

import torch


def write_shard(path, samples):
    """samples: iterable of dicts {__key__, ratio, latent (Tensor), emb (Tensor)[, pooled (Tensor)]}.  ``pooled`` -> an extra
    ``<key>.pooled.pt`` member: the pooled text projection SD3.5 conditions on (train_sd35.py:76-92 returns it beside the
    prompt embeddings; the reference's shard writer predates it and stores no such member -- this is the build's extension,
    read back by ``read_shard`` like any other ``.pt`` member and ignored by recipes that do not use it)."""
    with tarfile.open(path, "w") as tar:
        for s in samples:
            key = s["__key__"]
            members = [(".ratio", str(s["ratio"]).encode()), (".latent.pt", _save(s["latent"])), (".emb.pt", _save(s["emb"]))]
            if s.get("pooled") is not None:
                members.append((".pooled.pt", _save(s["pooled"])))
            for ext, payload in members:
                info = tarfile.TarInfo(key + ext)
                info.size = len(payload)
                tar.addfile(info, io.BytesIO(payload))


def _save(t):
    buf = io.BytesIO()
    torch.save(t.detach().cpu().contiguous(), buf)
    return buf.getvalue()


def read_shard(path):
    """Yield {__key__, ratio: float, 'latent.pt': Tensor, 'emb.pt': Tensor} in tar order (webdataset grouping:
    consecutive members sharing the basename up to the first dot form one sample)."""
    cur_key, cur = None, {}
    with tarfile.open(path, "r") as tar:
        for m in tar:
            if not m.isfile():
                continue
            base = m.name.split("/")[-1]
            key, _, ext = base.partition(".")
            key = m.name[: len(m.name) - len(base)] + key
            if cur_key is not None and key != cur_key:
                if _complete(cur):
                    yield cur
                cur = {}
            cur_key = key
            data = tar.extractfile(m).read()
            cur["__key__"] = key
            if ext == "ratio":
                cur["ratio"] = float(data.decode())
            elif ext.endswith("pt"):
                cur[ext] = torch.load(io.BytesIO(data), map_location="cpu", weights_only=True)
        if cur_key is not None and _complete(cur):
            yield cur


def _complete(s):
    return "ratio" in s and "latent.pt" in s and "emb.pt" in s


# ---- legacy cache: one file per sample ------------------------------------------------------------------------------
# common/cache.py:70-85 (CacheLoadFeatures.run) writes ``cache/{idx}.npy`` = torch.save((ratio, latent, (emb, mask)))
# with the embedding ZERO-PADDED to [300, C] and a float mask [300]; CacheFeaturesCompute (:28-45) keeps the same tuple
# gzip-compressed under ``datasets/<url>/<file>.npy``.  Despite the extension these are torch pickles, not numpy files.
LEGACY_PAD = 300


def write_legacy_sample(path, ratio, latent, emb, pad_to=LEGACY_PAD, compress=False):
    """Write one sample in the legacy tuple layout (emb: unpadded [L, C])."""
    L, C = emb.shape
    if L > pad_to:
        raise ValueError(f"embedding longer than the legacy pad length {pad_to}")
    padded = torch.zeros(pad_to, C, dtype=emb.dtype)
    padded[:L] = emb
    mask = torch.zeros(pad_to)
    mask[:L] = 1.0
    obj = (ratio, latent.detach().cpu(), (padded, mask))
    if compress:
        with gzip.open(path, "wb") as f:
            torch.save(obj, f)
    else:
        torch.save(obj, path)


def read_legacy_sample(path):
    """-> the same dict a shard sample decodes to: {__key__, ratio, 'latent.pt', 'emb.pt' (mask-true rows only)}."""
    with open(path, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"
    if gz:
        with gzip.open(path, "rb") as f:
            obj = torch.load(io.BytesIO(f.read()), map_location="cpu", weights_only=False)
    else:
        obj = torch.load(path, map_location="cpu", weights_only=False)
    ratio, latent, (emb, mask) = obj
    keep = int(mask.sum().item())
    return {"__key__": os.path.splitext(os.path.basename(path))[0], "ratio": float(ratio),
            "latent.pt": latent if latent.dim() == 3 else latent.squeeze(0), "emb.pt": emb[:keep].contiguous()}


def iter_legacy_cache(cache_dir, rank=0, world=1):
    """Samples ``{idx}.npy`` of a legacy cache directory in index order, strided over ranks (cache.py:57-63)."""
    idx = sorted(int(n[:-4]) for n in os.listdir(cache_dir) if n.endswith(".npy") and n[:-4].isdigit())
    for i in idx[rank::world]:
        yield read_legacy_sample(os.path.join(cache_dir, f"{i}.npy"))
