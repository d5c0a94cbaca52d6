#!/usr/bin/env python3
"""gpurun_out/pmc_fetch + pmc_write (scripts/gpu_pmc.sh) -> profiles/gemm_traffic.json: HBM bytes per launch of the GEMM family.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md."""
import csv, glob, json, sys

def per_kernel(tag):
    tot, n = {}, {}
    for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    return tot, n

ft, fn = per_kernel("fetch")
wt, wn = per_kernel("write")
fam = [k for k in ft if "gemm256_kernel" in k or "gemm_bf16_kernel" in k or "gemm256_grouped" in k]
launches = sum(fn[k] for k in fam)
fetch = 2.0 * 1024.0 * sum(ft[k] for k in fam) / launches
write = 1024.0 * sum(wt.get(k, 0.0) for k in fam) / max(1, sum(wn.get(k, 0) for k in fam))
out = {"kernel_family": "gemm256_kernel / gemm256_grouped_kernel / gemm_bf16_kernel (all variants)", "launches_sampled": launches,
       "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/gpu_pmc.sh, streams serialized: "
                 "YAT_SERIAL=1), KiB units, FETCH_SIZE doubled per the gfx950 "
                 "correction of MI355X_MICROARCH.md",
       "source": sys.argv[1] if len(sys.argv) > 1 else "profiles/pmc_per_kernel.txt",
       # bench.py quotes this file in `roofline.traffic`: say where and when it was measured (round-5 review item 3)
       "collected": {"when": __import__("datetime").datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"),
                     "where": "a builder's one-GPU MI355X box (gpurun), NOT the run that prints the bench line: rocprofv3 --pmc "
                              "passes cannot share a run with the timed region",
                     "tree": (sys.argv[2] if len(sys.argv) > 2 else "")}}
json.dump(out, open("profiles/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
