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


def test_bench_multi_rank_path_and_replica_selfcheck():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), rehearsed on the
    one-GPU box: gloo instead of RCCL (RCCL refuses two ranks on one device), both ranks on cuda:0.  Covers everything
    around the collective that the 8-GPU run will execute: rank-sharded inputs, the exchange autotune, max-over-ranks
    timing, the exchange-alone timing, and the self-check that every rank holds bit-identical parameters afterwards."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SISS_DIST_BACKEND="gloo", SISS_BENCH_SINGLE_DEVICE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "small", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    chk = out["config"]["dp_selfcheck"]
    assert chk == {"rccl_ranks_seen": 2, "replicas_identical": True, "finite": True, "backend": "gloo"}, chk
    assert out["config"]["global_batch"] == 4 and out["value"] > 0
    assert out["config"]["dp_allreduce"]["allreduce"]["ms"] > 0


def test_bench_launches_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with NO torchrun on the command line and no WORLD_SIZE in the environment: the parent starts
    the ranks itself (a child `torch.distributed.run`, before any GPU call) and its stdout carries rank 0's JSON line -- how a
    driver that runs `--gpus 8` the way it runs `--gpus 1` reaches the data-parallel path (gloo rehearsal on the one GPU)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SISS_DIST_BACKEND="gloo", SISS_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", "small", "--batch", "2", "--no-cpu-baseline", "--no-kernel-timing"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["dp_selfcheck"]["rccl_ranks_seen"] == 2
    assert out["config"]["dp_selfcheck"]["replicas_identical"] is True
