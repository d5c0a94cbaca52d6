"""Generate tests/golden/legacy_cache/{0,1}.npy.gz by running the REFERENCE's own legacy-cache writer.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_legacy_cache_golden.py

Imports /root/reference/common/cache.py (importable here: only tqdm / torch / gzip / json) and drives
``CacheLoadFeatures.run`` (common/cache.py:54-85) with a stub trainer whose extractor yields the two samples of
legacy_cache_inputs.py -- one 7-row prompt, one prompt that fills all 300 rows.  The files the reference writes
(``cache/{idx}.npy`` = ``torch.save((ratio, latent, (embeddings [300, 2304] fp32, mask [300] fp32)))``) are stored
gzip-compressed: data, not source.  One accommodation, stated because it touches the reference's namespace: the module does
``from tqdm import tqdm`` and then calls ``tqdm.tqdm(it, ...)`` (:65) -- an AttributeError on the class as written -- so the
script rebinds the module's global ``tqdm`` to a namespace whose ``.tqdm`` is a pass-through; every other line that runs is
the reference's."""
import gzip
import os
import shutil
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from common import cache as ref_cache                     # the reference's own module
    from legacy_cache_inputs import sample

    ref_cache.tqdm = types.SimpleNamespace(tqdm=lambda it, **kw: it)

    class _Model:
        def cpu(self):
            return self

    trainer = types.SimpleNamespace(
        model=_Model(), params=types.SimpleNamespace(cache_size=2),
        accelerator=types.SimpleNamespace(is_main_process=True, num_processes=1, process_index=0),
        data_extractor_iter=iter([[sample(0)], [sample(1)]]))
    out_dir = os.path.join(HERE, "legacy_cache")
    os.makedirs(out_dir, exist_ok=True)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "cache"))
        os.chdir(tmp)                                         # the writer's paths are relative: cache/{idx}.npy
        try:
            ref_cache.CacheLoadFeatures().run(trainer)
        finally:
            os.chdir(cwd)
        for idx in (0, 1):
            src = os.path.join(tmp, "cache", f"{idx}.npy")
            dst = os.path.join(out_dir, f"{idx}.npy.gz")
            with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
                shutil.copyfileobj(f, g)
            print("wrote", dst, os.path.getsize(src), "->", os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
