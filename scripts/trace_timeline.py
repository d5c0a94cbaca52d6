#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py: for the last full step, per HIP queue the busy time, the time
no kernel runs at all, how much of the step has >= 1 GEMM running, and the CU-time (duration x min(1, workgroups / 256)) of
the GEMM family against 256 CUs x step -- i.e. how full the chip is kept.  python scripts/trace_timeline.py TRACE.csv [steps]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"],
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) //
                 max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
# steps are delimited by the AdamW kernel launches of the bucketed optimizer: take the window between the first adamw of the
# second-to-last step and the first adamw of the last one
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3]]
# group adamw launches that are close in time into steps
groups = [[ad[0]]]
for i in ad[1:]:
    if rows[i][0] - rows[groups[-1][-1]][0] < 20e6:
        groups[-1].append(i)
    else:
        groups.append([i])
g0, g1 = groups[-3][0], groups[-2][0]
t0, t1 = rows[g0][0], rows[g1][0]
win = [r for r in rows if r[0] >= t0 and r[0] < t1]
step = (t1 - t0) / 1e6
print(f"step window {step:.2f} ms, {len(win)} kernels, queues {sorted(set(r[2] for r in win))}")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot / 1e6
allb = union([(r[0], min(r[1], t1)) for r in win])
print(f"some kernel running: {allb:.2f} ms ({100 * allb / step:.1f} %), idle {step - allb:.2f} ms")
for q in sorted(set(r[2] for r in win)):
    b = union([(r[0], min(r[1], t1)) for r in win if r[2] == q])
    print(f"  queue {q}: busy {b:.2f} ms ({100 * b / step:.1f} %), {sum(1 for r in win if r[2] == q)} kernels")
gem = [r for r in win if "gemm" in r[3] and "splitk" not in r[3]]
gb = union([(r[0], min(r[1], t1)) for r in gem])
cu = sum((min(r[1], t1) - r[0]) * min(1.0, r[4] / 256.0) for r in gem) / 1e6
print(f"GEMM running: {gb:.2f} ms ({100 * gb / step:.1f} %); sum of GEMM durations {sum(r[1]-r[0] for r in gem)/1e6:.2f} ms; "
      f"GEMM CU-time {cu:.2f} chip-ms ({100 * cu / step:.1f} % of the chip)")
non = collections.Counter()
for r in win:
    if "gemm" not in r[3] or "splitk" in r[3]:
        non[r[3][:60]] += (r[1] - r[0]) / 1e6
print("non-GEMM kernels, summed durations (overlapped run):")
for k, v in non.most_common(12):
    print(f"  {v:7.2f} ms  {k}")

# gaps on the busiest queue: where does the critical stream wait?
q = max(set(r[2] for r in win), key=lambda q_: sum(r[1] - r[0] for r in win if r[2] == q_))
seq = sorted([r for r in win if r[2] == q])
gaps = [(seq[i + 1][0] - seq[i][1], seq[i][3][:50], seq[i + 1][3][:50]) for i in range(len(seq) - 1)]
tot = sum(g[0] for g in gaps if g[0] > 0) / 1e6
print(f"queue {q}: {len(seq)} kernels, gaps between consecutive kernels total {tot:.2f} ms; "
      f"{sum(1 for g in gaps if g[0] > 20000)} gaps > 20 us = {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms; "
      f"median gap {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.1f} us")
agg = collections.Counter()
for g, a, b in gaps:
    if g > 20000:
        agg[(a, b)] += g / 1e6
print("largest waits (after -> before), summed over the step:")
for (a, b), v in agg.most_common(12):
    print(f"  {v:6.2f} ms  {a}  ->  {b}")
