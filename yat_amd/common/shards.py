"""Cached-feature shard format: reader and writer without the webdataset dependency.

Wire format = what the reference's FeaturesExtractor writes (common/features_extractor.py:83-88) and its
BucketSampler decodes (common/bucket_sampler.py:138-156): a POSIX tar whose members are grouped by key,
``<key>.ratio`` (ascii float), ``<key>.latent.pt`` (torch.save of the [C, h, w] bf16 latent) and ``<key>.emb.pt``
(torch.save of the UNPADDED [L, C] bf16 text embedding -- only mask-true rows, train_sana.py:92-94).
"""
from __future__ import annotations

import io
import tarfile

import torch


def write_shard(path, samples):
    """samples: iterable of dicts {__key__, ratio, latent (Tensor), emb (Tensor)}."""
    with tarfile.open(path, "w") as tar:
        for s in samples:
            key = s["__key__"]
            for ext, payload in ((".ratio", str(s["ratio"]).encode()), (".latent.pt", _save(s["latent"])),
                                 (".emb.pt", _save(s["emb"]))):
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
