"""Data-parallel step on the GPU: two ranks (one process each) against the single-process global batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_global_batch():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577",
                        os.path.join(ROOT, "tests", "dp_gpu_worker.py"), ROOT],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("dp gpu ok") == 2
