#!/usr/bin/env python3
"""Instruction histogram of the innermost loop(s) of a kernel in a hipcc -S listing:
    python scripts/asm_loop_hist.py yat_amd/build/asm/sdpa.s sdpa_fwd_kernelILi3ELi5ELi3
Prints, per loop body the compiler marked, the instruction counts and an issue-cycle estimate from the measured per-instruction
issue costs of MI355X_MICROARCH.md (MFMA 16x16x32: 16 cycles of pipe, holds the vector issue for 8; transcendental 8; other VALU 4)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])
body = text[start:end]
heads = [i for i, l in enumerate(body) if "Loop Header" in l]
print(f"{pat}: {end - start} lines, {len(heads)} loop header(s)")
for hi in heads:
    label = body[hi].split(":")[0]
    bb = label.lstrip(".L")                     # "BB6_40"
    # the loop = the header block + every later block the compiler marks "in Loop: Header=<bb>"
    blocks = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    member = [i for i in blocks if i == hi or f"Header={bb} " in body[i] or body[i].rstrip().endswith(f"Header={bb}")]
    if not member:
        continue
    lines_in = []
    for i in member:
        nxt = next((k for k in blocks if k > i), len(body))
        lines_in += body[i + 1:nxt]
    cnt = collections.Counter()
    for l in lines_in:
        l = l.strip()
        if l and not l.startswith((";", ".")):
            cnt[l.split()[0]] += 1
    trans = sum(v for k, v in cnt.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
    mfma = sum(v for k, v in cnt.items() if k.startswith("v_mfma"))
    valu = sum(v for k, v in cnt.items() if k.startswith("v_")) - trans - mfma
    lds = sum(v for k, v in cnt.items() if k.startswith("ds_"))
    print(f"\nloop {label} ({len(member)} blocks, conditional ones included): {sum(cnt.values())} instructions: {mfma} MFMA, "
          f"{trans} transcendental, {valu} other VALU, {lds} LDS, "
          f"{sum(v for k, v in cnt.items() if k.startswith('s_'))} scalar, {sum(v for k, v in cnt.items() if k.startswith('buffer_'))} buffer")
    print(f"  matrix pipe {16 * mfma} cycles; vector issue ~{8 * mfma + 8 * trans + 4 * valu} cycles (MFMA 8 + trans 8 + VALU 4)")
    for k, v in cnt.most_common(40):
        print(f"  {k:34s}{v}")
