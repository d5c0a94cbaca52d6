"""`python bench.py --gpus N` must start by itself (round-3 review, item 1): with N > 1 and no launcher around it the parent
starts `python -m torch.distributed.run` as a fresh child process before anything touches the GPU, relays rank 0's one JSON
line and the exit code.  The reference's counterpart is `accelerate launch` (README.md:62, common/trainer.py:31-37)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_starts_ranks_and_relays_exit_code():
    """No GPU here: both ranks the parent started must say so and the parent must hand their failure back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (the GPU box runs test_self_launch_world2_over_gloo_on_one_gpu)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "without a launcher: starting" in r.stderr                 # the parent took the self-launch path ...
    assert "--nproc-per-node=2" in r.stderr
    assert "bench.py needs a GPU" in r.stderr                          # ... and a rank ran bench.py's own main() (the launcher
    #                                                                    ends the other one as soon as the first has failed)
    assert r.stdout.strip() == ""                                      # no JSON line from a failed run


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagree" in r.stderr


@pytest.mark.gpu
def test_self_launch_world2_over_gloo_on_one_gpu():
    """The driver's N > 1 command shape, by itself: two ranks on cuda:0 (RCCL refuses two ranks on one device, so the
    transport is gloo; everything else -- rank discovery, broadcast, bucket hooks, barrier + max over ranks -- is the N > 1
    path), a 2-block model, the JSON line on stdout and nothing else."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-gemm-timer", "--comm-steps", "2"],
                       env=_env(YAT_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert d["comm"]["world"] == 2 and "error" not in d["comm"]
    # what the communicator itself saw (round-4 review 5b), not the launcher's environment; and the channel cap in force
    assert d["comm"]["communicator"]["world"] == 2 and d["comm"]["communicator"]["rank"] == 0
    assert "gloo" in d["comm"]["communicator"]["owner"]
    assert d["comm"]["rccl_channels"]["NCCL_MAX_NCHANNELS"] is None          # RCCL's default unless asked (yat_amd/ddp.py)
