"""`python bench.py --gpus N` without an outer launcher must start its N ranks itself (the driver runs exactly that) and
fail CLEANLY, with the failing rank's message and a non-zero exit code, on a node with fewer GPUs -- not with a usage text."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(180)
def test_self_launch_fails_cleanly_without_enough_gpus():
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode != 0
    assert "needs 2 GPUs on this node" in r.stderr and "rank 1" in r.stderr       # the child's own message
    assert "stopping the other ranks" in r.stderr and "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""                                                  # no half-written JSON line


def test_launched_rank_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=170, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
