"""Build libyat_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m yat_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file (git-ignored, but it travels
to the GPU box with the gpurun snapshot).  A content hash of the sources is stored beside it so
repeated builds are no-ops.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libyat_hip.so")
SOURCES = ["gemm.hip", "gemm256.hip", "rowops.hip", "elementwise.hip", "optim.hip", "linear_attn.hip", "sdpa.hip", "dwconv_glu.hip", "lokr.hip", "pixart_ops.hip", "mmdit_ops.hip", "comm.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value"]
# Per-source additions.  sdpa.hip: keep the MFMA accumulators in ordinary VGPRs -- the softmax reads every score and rescales
# every output accumulator between MFMAs, and with the default AGPR form that was ~150 v_accvgpr_read/write moves per key tile
# per wave (more VALU cycles than the softmax itself); gfx950's MFMA takes VGPR operands directly.
EXTRA_FLAGS = {"sdpa.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/yat_hip.h"]:
        path = os.path.normpath(os.path.join(CSRC, name))
        if os.path.isfile(path):
            h.update(name.encode())
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".sha256"
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    resources = {}

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        # -Rpass-analysis=kernel-resource-usage: per-kernel VGPRs / spills / scratch as compiler remarks (free), kept in
        # build/resources.json.  A kernel that silently starts using scratch (an innocent-looking epilogue branch did that
        # to every 256x320 GEMM once: +11 ms per step) fails the build instead of the benchmark.
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c",
               os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = resources.setdefault(f"{src}:{m.group(1)}", {})
                continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|"
                          r"LDS Size \[bytes/block\]): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1)] = int(m.group(2))
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    with open(os.path.join(objdir, "resources.json"), "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    bad = {k: v for k, v in resources.items() if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0)}
    if bad:
        raise RuntimeError("kernels using scratch memory / spilling VGPRs (not allowed on this path): "
                           + json.dumps(bad, indent=1))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print(f"[yat_amd.build] built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
