#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_*) per kernel: counter means per dispatch."""
import csv, collections, glob, sys, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in out.items():
    n = max(len(v) for v in c.values())
    rows.append((k, n, {cn: sum(v) / len(v) for cn, v in c.items()}))
rows.sort(key=lambda r: -r[2].get("SQ_BUSY_CYCLES", 0) * r[1])
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]
print("kernel".ljust(60), "n".rjust(5), " ".join(c[-14:].rjust(14) for c in cols))
for k, n, m in rows[:40]:
    print(k.ljust(60), str(n).rjust(5), " ".join(("%.3g" % m[c]).rjust(14) if c in m else "-".rjust(14) for c in cols))
