#!/usr/bin/env python3
"""Per-kernel HBM roofline of the non-GEMM kernels of the SANA step from a SERIALIZED rocprofv3 kernel-stats csv
(scripts/gpu_round.sh profserial: one stream, so a kernel's duration is its own):

    python scripts/kernel_roofline.py profiles/r05_i_kernel_stats_serialized.csv [steps in the profile]

Algorithmic bytes per call are the tensors a kernel must read and write once at the bench's shapes (B = 8 images, mean
M = 8048 token rows over the four buckets, D = 2240, Hc = 5600, 1.6045e9 parameters); achieved = bytes / average duration,
against the 8 TB/s spec and the ~6.3 TB/s a streaming copy reaches on this chip (MI355X_MICROARCH.md).  Kernels whose work is
latency- or VALU-bound at these sizes show as a low fraction -- that is what the column is for."""
import csv
import re
import sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 11.0
M, D, Hc, NP = 8048.0, 2240.0, 5600.0, 1.6045e9
T = 1300.0            # packed text rows per batch (mean)
BYTES = [            # (regex on the kernel name, algorithmic bytes per call, what is counted)
    (r"adamw_kernel", 14 * NP, "p, g, m, v read; p, m, v written (bf16): 14 B / parameter, one launch"),
    (r"gradnorm(_pieces)?_partial", 2 * NP, "g read"),
    (r"dwglu_(tile|stream)_kernel", 10 * M * Hc, "s read; u (kept for the backward) and y written"),
    (r"dwglu_bwd2", 12 * M * Hc, "du, z read; dz written (s = SiLU(z) is recomputed since round 5)"),
    (r"ln_mod_fwd", 4 * M * D, "x read, h written"),
    (r"ln_mod_bwd_rows", 8 * M * D, "dy, x, incoming dx read; dx written"),
    (r"ln_mod_bwd_cols", 4 * M * D, "dy, x read (column statistics)"),
    (r"strip_kernel<1>", 6 * M * D, "gate backward: dout, lin read; dlin written"),
    (r"sdpa_fwd", 4 * M * D, "cross-attention: q read, o written (k, v: text rows, small)"),
    (r"sdpa_bwd_dq", 8 * M * D, "q, do, o read; dq written"),
    (r"sdpa_bwd_dkv", 4 * M * D, "q, do read (dk, dv: text rows)"),
    (r"la_state", 4 * M * D, "k, v read"),
    (r"la_fwd", 4 * M * D, "q read, out written"),
    (r"la_bwd_q", 6 * M * D, "q, dout read; dq written (the state and its gradient slabs are small)"),
    (r"la_bwd_kv", 8 * M * D, "k, v read; dk, dv written"),
]
rows = list(csv.DictReader(open(path)))
print(f"# {path}: {steps:g} steps; serialized stream; achieved = algorithmic bytes / average duration")
print(f"{'kernel':44s} {'calls/step':>10s} {'avg us':>8s} {'ms/step':>8s} {'MB/call':>9s} {'TB/s':>6s} {'of 8.0':>7s} {'of 6.3':>7s}  counted")
tot_ms = tot_floor = 0.0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    for pat, nbytes, what in BYTES:
        if re.search(pat, name):
            calls = int(r["Calls"]) / steps
            avg_us = float(r["AverageNs"]) / 1e3
            if "adamw" in pat and calls > 1.5:          # overlapped mode: one launch per bucket
                nbytes = nbytes / calls
            tbs = nbytes / avg_us / 1e6
            ms = int(r["TotalDurationNs"]) / 1e6 / steps
            tot_ms += ms
            tot_floor += calls * nbytes / 6.3e12 * 1e3
            short = name.split("(")[0][-44:]
            print(f"{short:44s} {calls:10.1f} {avg_us:8.1f} {ms:8.2f} {nbytes / 1e6:9.1f} {tbs:6.2f} {tbs / 8.0:7.2f} {tbs / 6.3:7.2f}  {what}")
            break
print(f"# listed kernels: {tot_ms:.2f} ms/step; at 6.3 TB/s their bytes take {tot_floor:.2f} ms/step")
