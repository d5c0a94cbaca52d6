"""Generate tests/golden/params_*.json by importing the REFERENCE config reader.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_params_golden.py

For every ``tests/golden/config_*.yaml`` (written for this repo: one per BASELINE config plus a
quirks file) it runs /root/reference/common/training_parameters_reader.py:TrainingParameters.read_yaml
and dumps ``vars(params)`` as JSON.  The JSON files are data (inputs + expected outputs); the
reference source itself is never copied.
"""
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from common.training_parameters_reader import TrainingParameters  # the reference's own reader

    for path in sorted(glob.glob(os.path.join(HERE, "config_*.yaml"))):
        p = TrainingParameters()
        p.read_yaml(path)
        out = os.path.join(HERE, "params_" + os.path.basename(path)[len("config_"):-len(".yaml")] + ".json")
        with open(out, "w") as f:
            json.dump(vars(p), f, indent=1, sort_keys=True)
        print("wrote", out)


if __name__ == "__main__":
    main()
