#!/usr/bin/env python3
"""Build a variant of libyat_hip.so with extra compiler flags for ONE source (same-box A/B of kernel variants through
YAT_HIP_LIB):  python scripts/build_variant.py NAME SOURCE.hip -DFOO=1 ...  ->  yat_amd/build/variants/libyat_NAME.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yat_amd import build as B

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()                                            # objects of every other source
out_dir = os.path.join(B.HERE, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(B.CSRC, src), "-o", obj]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
import re
cur = None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
    m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur and ("dwglu" in cur or len(sys.argv) > 20):
        print(cur[:60], m.group(1), m.group(2))
objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != src] + [obj]
lib = os.path.join(out_dir, f"libyat_{name}.so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, *objs, "-ldl"], check=True)
print(lib)
