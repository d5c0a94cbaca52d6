"""Generate tests/golden/nccl_vars.json: the environment /root/reference/utils/set_nccl_vars.py exports when imported.

Run in the build container only:  python tests/golden/make_nccl_vars_golden.py
The module is importable here.  The JSON (variable -> value) is data; tests/test_host_logic.py holds
yat_amd/ddp.py's decision table (which of these the build deliberately does not inherit, and why) to it."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r"""
import json, os, sys
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
before = dict(os.environ)
import utils.set_nccl_vars  # the reference's own module
print(json.dumps({k: v for k, v in os.environ.items() if before.get(k) != v}, sort_keys=True))
"""


def main():
    env = {k: v for k, v in os.environ.items() if not k.startswith("NCCL_")}
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, check=True).stdout
    data = json.loads(out.strip().splitlines()[-1])
    path = os.path.join(HERE, "nccl_vars.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", path, data)


if __name__ == "__main__":
    main()
