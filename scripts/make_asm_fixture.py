#!/usr/bin/env python3
"""Cut the skeleton of one kernel out of a hipcc -save-temps listing for the build-time assembly checkers' unit tests
(tests/test_build_checks.py):  labels, block / loop annotations and every instruction the checkers look at (vector-memory,
LDS-DMA, s_waitcnt, s_barrier, branches, scratch / flat); all other instructions are dropped.

    python scripts/make_asm_fixture.py <listing.s> <mangled-name regex> <out.s>
"""
import re
import sys

src, pat, out = sys.argv[1:4]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and re.search(pat, l.split(":")[0]))
end = next(k for k in range(start, len(lines)) if ".amdhsa_kernel" in lines[k] or lines[k].startswith(".Lfunc_end"))
keep = re.compile(r"\s*((buffer|global|flat|scratch)_|s_waitcnt|s_barrier|s_cbranch|s_branch|s_endpgm)")
res = [lines[start].split(";")[0].rstrip()]
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        res.append(re.sub(r"%\S+", "%blk", t))                  # (long mangled block names are of no interest)
    elif t.startswith("; %bb."):
        res.append(t)
    elif t.startswith(";") and "in Loop:" in t:
        res.append("                                        " + t)
    elif keep.match(l):
        res.append("\t" + t)
last = max(i for i, l in enumerate(res) if "in Loop:" in l or "Loop Header" in l)
tail = next((i for i in range(last + 1, len(res)) if re.match(r"^\.LBB\d+_\d+:", res[i]) and "in Loop" not in res[i + 1 if i + 1 < len(res) else i]), len(res))
res = res[:min(len(res), tail + 40)]                            # (the epilogue behind the K loops: not what is checked)
res.append(".Lfunc_end_fixture:")
open(out, "w").write("\n".join(res) + "\n")
print(f"{out}: {len(res)} lines")
