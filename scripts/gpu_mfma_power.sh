#!/bin/bash
# MFMA-only power probe: TFLOP/s, shader clock and socket power of back-to-back bf16 MFMAs of each shape (no memory traffic).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for t in 16 17 32 17; do
  ( for i in $(seq 1 14); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > gpurun_out/mfma_samples_$t.jsonl &
  SP=$!
  scripts/probes/mfma_power $t 3.0
  wait $SP
  python3 - $t <<'PY'
import json, re, sys
s, p = [], []
for line in open(f"gpurun_out/mfma_samples_{sys.argv[1]}.jsonl"):
    try: d = json.loads(line)["card0"]
    except Exception: continue
    s.append(int(re.search(r"(\d+)", d["sclk clock speed:"]).group(1)))
    p.append(float(next(v for k, v in d.items() if "Power" in k)))
s, p = sorted(s[3:-2]), sorted(p[3:-2])
if s: print(f"    sclk {s[len(s)//2]} MHz  power {p[len(p)//2]:.0f} W  ({len(s)} samples)")
PY
done
