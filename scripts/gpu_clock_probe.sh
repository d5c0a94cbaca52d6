#!/bin/bash
# Shader clock and socket power while the step runs: is the GEMM family clock-limited?  Samples rocm-smi every 0.5 s beside
# a 200-step bench; prints min / median / max of sclk and power over the timed region.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( for i in $(seq 1 80); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > gpurun_out/clock_samples.jsonl &
SP=$!
python bench.py --steps 250 --warmup 10 --no-cpu-baseline --no-gemm-timer > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err
kill $SP 2>/dev/null
python - <<'PY'
import json, re, statistics
s, p = [], []
for line in open("gpurun_out/clock_samples.jsonl"):
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "sclk" in k.lower():
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            if m: s.append(int(m.group(1)))
        if "power" in k.lower() and "W" in k:
            try: p.append(float(v))
            except Exception: pass
print("samples", len(s), len(p))
if s: print("sclk MHz: min %d median %d max %d" % (min(s), statistics.median(s), max(s)), s[:60])
if p: print("power W: min %.0f median %.0f max %.0f" % (min(p), statistics.median(p), max(p)), [round(x) for x in p[:60]])
PY
grep "timed region" gpurun_out/clock_bench.err
head -c 600 gpurun_out/clock_samples.jsonl
