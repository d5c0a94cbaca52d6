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
SOURCES = ["gemm.hip", "gemm256.hip", "rowops.hip", "elementwise.hip", "optim.hip", "linear_attn.hip", "sdpa.hip", "dwconv_glu.hip", "lokr.hip", "pixart_ops.hip", "mmdit_ops.hip", "comm.hip", "plan.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value"]
# Per-source additions.  sdpa.hip: keep the MFMA accumulators in ordinary VGPRs -- the softmax reads every score and rescales
# every output accumulator between MFMAs, and with the default AGPR form that was ~150 v_accvgpr_read/write moves per key tile
# per wave (more VALU cycles than the softmax itself); gfx950's MFMA takes VGPR operands directly.
# sdpa.hip, -fno-slp-vectorize: the SLP vectorizer pairs the per-score f32 multiplies / adds of the softmax loops into v_pk_*
# instructions (dearer than two plain ones beside MFMAs: MI355X_MICROARCH.md) and then shuffles the pairs back for the bf16 packs
EXTRA_FLAGS = {"sdpa.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


PLAN_GEN_VERSION = "1"

# Sources whose device assembly is kept (-save-temps) for a structural check after the compile: source -> checker name.
ASM_CHECKED = {"dwconv_glu.hip": "check_dwconv_stream_asm", "gemm256.hip": "check_gemm256_deep_asm"}

_VM_OTHER = re.compile(r"(buffer|global)_(load|store|atomic)|flat_|scratch_")


def _asm_functions(text, pattern):
    """[(mangled name, instruction-and-label lines)] of every function whose mangled name matches ``pattern``."""
    lines = text.split("\n")
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S*):", l)
        if m and re.search(pattern, m.group(1)):
            end = next((k for k in range(i, len(lines)) if ".amdhsa_kernel" in lines[k] or lines[k].startswith(".Lfunc_end")), len(lines))
            out.append((m.group(1), lines[i + 1:end]))
    return out


def _asm_loops(body):
    """The loops the compiler annotated in one function: [(header label, [instruction lines])].  A loop = its header block
    plus every block marked ``in Loop: Header=<label>`` (the annotation sits on the block's label line or the line after)."""
    blocks, cur = [], None                       # (label or None, comment text, [instructions])
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", t)
        if m or t.startswith("; %bb."):
            cur = [m.group(1) if m else None, m.group(2) if m else t, []]
            blocks.append(cur)
            continue
        if cur is None:
            continue
        if t.startswith(";"):
            cur[1] += " " + t                    # (continuation of the block's comment: "in Loop: Header=...")
        elif t and not t.startswith("."):
            cur[2].append(t)
    loops = []
    for lab, com, _ in blocks:
        if lab and "Loop Header" in com:
            key = lab.lstrip(".L")
            ins = []
            for l2, c2, i2 in blocks:
                if l2 == lab or re.search(rf"in Loop: Header={key}\b", c2):
                    ins += i2
            loops.append((lab, ins))
    return loops


def check_dwconv_stream_asm(asm_path: str) -> None:
    """csrc/dwconv_tile_fwd.inc, dwglu_stream_kernel<SSEG>: the LDS-DMA prefetch of the next rows is awaited with ``s_waitcnt
    vmcnt(3*SSEG)`` -- correct only while every wave issues EXACTLY 3*SSEG vector-memory operations (the u / y stores) after
    the prefetch in every step.  A compiler that spills, splits or merges a store, or adds any other vector-memory
    operation to the loop would make the wait return early and the kernel read stale ring rows without any test on this
    toolchain noticing (round-3 advisor finding).  Fail the BUILD instead: EVERY instantiation (SSEG read from its mangled
    name) must contain exactly 3*SSEG buffer stores, the counted wait, no scratch / flat operation, and no plain load after
    the first LDS-DMA."""
    funcs = _asm_functions(open(asm_path).read(), r"dwglu_stream_kernelILi\d+E")
    if not funcs:
        raise RuntimeError("dwglu_stream_kernel not found in " + asm_path)
    for name, body in funcs:
        sseg = int(re.search(r"dwglu_stream_kernelILi(\d+)E", name).group(1))
        ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
        stores = [l for l in ins if re.match(r"(buffer|global)_store", l)]
        bad = [l for l in ins if l.startswith(("scratch_", "flat_")) or re.match(r"(buffer|global)_atomic", l)]
        first_dma = next((i for i, l in enumerate(ins) if l.startswith("buffer_load") and " lds" in l), None)
        late_loads = [l for l in ins[first_dma:] if re.match(r"(buffer|global)_load", l) and " lds" not in l] if first_dma is not None else ["<no LDS-DMA>"]
        waits = [l for l in ins if re.match(rf"s_waitcnt vmcnt\({3 * sseg}\)", l)]
        if len(stores) != 3 * sseg or bad or late_loads or len(waits) != 1:
            raise RuntimeError(f"{name} no longer matches its `s_waitcnt vmcnt({3 * sseg})`: {len(stores)} stores "
                               f"(want {3 * sseg}), {len(waits)} counted wait(s) (want 1), scratch/flat/atomic ops {bad[:3]}, loads after "
                               f"the first LDS-DMA {late_loads[:3]} -- fix the kernel or fall back to vmcnt(0) there")


def check_gemm256_deep_asm(asm_path: str) -> None:
    """csrc/gemm256.hip, DEEP schedule: the K loop's FAST form waits for an LDS-DMA batch with ``s_waitcnt vmcnt(n)``, n = the
    pieces of the two younger batches -- correct only while a wave's loop iteration issues EXACTLY the pieces the source
    counts (PER_TILE = NAO + 2 NAH + 2 NB per K-tile) and no other vector-memory operation: one spilled register, one
    compiler-split load or a stray global access inside the loop and every gradient GEMM reads LDS bytes that have not
    landed, with no test on this one toolchain bound to notice (round-4 review).  Per instantiation (layout and tile width
    from the mangled name) the build therefore requires: exactly two loops with counted waits (wave group 0 and 1), each
    with exactly its PER_TILE LDS-DMA loads, wait values exactly those of wait_g0 / wait_g1, no other buffer / global / flat
    / scratch operation inside them; every other loop that issues LDS-DMA waits with vmcnt(0) only; no scratch anywhere."""
    funcs = _asm_functions(open(asm_path).read(), r"gemm256_(grouped_)?kernelILb[01]ELb[01]ELi[45]E")
    if not funcs:
        raise RuntimeError("no gemm256 kernel found in " + asm_path)
    for name, body in funcs:
        a_t, b_t, nt = re.search(r"kernelILb([01])ELb([01])ELi([45])E", name).groups()
        nao, nah = (0, 2) if a_t == "1" else (4, 0)
        want = []                                                      # (pieces per iteration, wait values) of group 0, 1
        for grp, nb in ((0, 2), (1, 3 if nt == "5" else 2)):
            per_tile = nao + 2 * nah + 2 * nb
            want.append((per_tile, {per_tile} if grp == 0 or not nao else {2 * nb + nao, 2 * nb}))
        if any(l.strip().startswith("scratch_") for l in body):
            raise RuntimeError(f"{name}: scratch access (a spill) in a gemm256 kernel")
        fast = []
        for label, ins in _asm_loops(body):
            dma = [l for l in ins if l.startswith("buffer_load") and " lds" in l]
            other = [l for l in ins if _VM_OTHER.match(l) and not (l.startswith("buffer_load") and " lds" in l)]
            counted = [int(m.group(1)) for l in ins for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", l)] if m and int(m.group(1)) > 0]
            if not dma:
                if counted:
                    raise RuntimeError(f"{name} {label}: counted vmcnt wait in a loop without LDS-DMA")
                continue
            if not counted:
                continue                          # the general form of the loop: vmcnt(0) everywhere
            if other:
                raise RuntimeError(f"{name} {label}: vector-memory operations other than LDS-DMA inside a loop with counted "
                                   f"waits: {other[:3]}")
            sig = (len(dma), set(counted))
            if sig not in want:
                raise RuntimeError(f"{name} {label}: {len(dma)} LDS-DMA loads per iteration with waits {sorted(set(counted))}; the "
                                   f"source counts on {[(k, sorted(v)) for k, v in want]} (pieces, waits) -- fix the kernel or "
                                   f"fall back to vmcnt(0)")
            fast.append(sig)
        if len(fast) != 2 or any(w not in fast for w in want):
            raise RuntimeError(f"{name}: expected one counted-wait loop per wave group {[(k, sorted(v)) for k, v in want]}, "
                               f"found {[(k, sorted(v)) for k, v in fast]}")


def generate_plan_dispatch(out_path: str) -> None:
    """include/yat_hip.h -> build/plan_dispatch.inc: one trampoline per int-returning entry point that unpacks a
    ``yat_plan_arg`` record into the typed call (float / double parameters travel in ``.d``, the rest in ``.i`` / ``.p``),
    the name table and the switch.  Deterministic in the header, so the library digest covers it."""
    text = open(os.path.join(ROOT, "include", "yat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    funcs = []
    for m in re.finditer(r"\bint\s+(yat_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        if name.startswith("yat_plan_") or name in ("yat_version",) or name.startswith("yat_comm_unique") or name == "yat_comm_init":
            continue
        plist = [] if params in ("", "void") else [" ".join(p.split()) for p in params.split(",")]
        casts = []
        for i, p in enumerate(plist):
            ptype = re.sub(r"\s*\w+$", "", p).strip() if not p.endswith("*") else p      # drop the parameter name
            if "*" in ptype or ptype == "yat_stream_t":
                casts.append(f"({ptype})a[{i}].p")
            elif ptype in ("float", "double"):
                casts.append(f"({ptype})a[{i}].d")
            else:
                casts.append(f"({ptype})a[{i}].i")
        if len(casts) > 30:
            raise RuntimeError(f"{name}: more than YAT_PLAN_MAX_ARGS parameters")
        funcs.append((name, casts))
    funcs.sort()
    lines = ["// GENERATED by yat_amd/build.py from include/yat_hip.h -- do not edit", "#pragma once",
             f"static const int kPlanCount = {len(funcs)};",
             "static const char* const kPlanNames[] = {" + ", ".join(f'"{n}"' for n, _ in funcs) + "};",
             "static int plan_dispatch(int op, const yat_plan_arg* a) {", "    switch (op) {"]
    for i, (n, casts) in enumerate(funcs):
        lines.append(f"        case {i}: return {n}({', '.join(casts)});")
    lines += ["        default: return YAT_EINVAL;", "    }", "}", ""]
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    new = "\n".join(lines)
    if not os.path.exists(out_path) or open(out_path).read() != new:
        with open(out_path, "w") as f:
            f.write(new)


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/yat_hip.h"]:
        path = os.path.normpath(os.path.join(CSRC, name))
        if os.path.isfile(path):
            h.update(name.encode())
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(PLAN_GEN_VERSION.encode())
    h.update(repr(sorted(ASM_CHECKED.items())).encode())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def parse_resource_remarks(stderr: str, src: str) -> dict:
    """-Rpass-analysis=kernel-resource-usage remarks of one compile -> {"<src>:<mangled kernel>": {field: value}}.  With
    -save-temps (the assembly-checked sources) every remark carries a file:line:col prefix; without, none."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:.*?\bFunction Name: (\S+)", line)
        if m:
            cur = out.setdefault(f"{src}:{m.group(1)}", {})
            continue
        m = re.search(r"remark:.*?\s(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".sha256"
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    generate_plan_dispatch(os.path.join(objdir, "plan_dispatch.inc"))

    resources = {}

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        # -Rpass-analysis=kernel-resource-usage: per-kernel VGPRs / spills / scratch as compiler remarks (free), kept in
        # build/resources.json.  A kernel that silently starts using scratch (an innocent-looking epilogue branch did that
        # to every 256x320 GEMM once: +11 ms per step) fails the build instead of the benchmark.
        temps = os.path.join(objdir, "temps_" + src.replace(".hip", ""))
        keep_asm = src in ASM_CHECKED
        if keep_asm:
            os.makedirs(temps, exist_ok=True)
            obj_out = os.path.join(temps, src.replace(".hip", ".o"))       # (-save-temps=obj: the temporaries land beside it)
        else:
            obj_out = obj
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage",
               *(["-save-temps=obj"] if keep_asm else []), "-c", os.path.join(CSRC, src), "-o", obj_out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if keep_asm:
            import shutil
            asm = [f for f in os.listdir(temps) if f.endswith(".s") and "amdgcn" in f]
            if len(asm) != 1:
                raise RuntimeError(f"{src}: expected one device assembly file in {temps}, found {asm}")
            globals()[ASM_CHECKED[src]](os.path.join(temps, asm[0]))
            shutil.copyfile(obj_out, obj)
            shutil.rmtree(temps, ignore_errors=True)
        resources.update(parse_resource_remarks(r.stderr, src))
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
