"""RCCL first contact on the one GPU of a build box: a world-size-1 `nccl` process group, created in a
fresh child process exactly as bench.py creates it for N > 1, with the test-only switch dp.FORCE_COLLECTIVES making the step
issue its collectives although one rank has nothing to exchange (a sum over one rank is the identity, so the no-group step is
the oracle).  tests/rccl_ws1_worker.py does the work; this file starts it (never re-execs: a child process per leg) and reads
its record.  Replaces /root/reference/delete_celeb.py:99-101,304 (accelerate.prepare -> DDP -> NCCL all-reduce)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_ws1_worker.py"), ROOT, mode],
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WS1 ")]
    rec = json.loads(lines[-1][len("RCCL_WS1 "):]) if lines else None
    return r, rec


def test_world_size_1_rccl_group_runs_every_exchange_mode_of_the_step():
    """(i) dp.allreduce_flat_grads on the REAL flat pair (2 x 113.7 M f32 of the CelebA-HQ UNet) returns it bit for bit; (ii) the
    overlap hook issues its async grouped collective from inside UNetEngine.backward; (iii) one SISSStepper.step per exchange mode
    (overlap = two grouped collectives, serial all-reduce, sharded update = two all-to-alls + shard AdamW + all-gather) equals the
    no-group step (scalars to 1e-4 -- the weight gradients' f32 atomics are order-dependent, DESIGN section 6 -- and > 98 % of the
    updated parameters bit-identical); the exchange autotune runs on the real backend."""
    r, rec = _run("steps", 900)
    assert r.returncode == 0 and rec is not None and rec["ok"], (r.stdout[-3000:], r.stderr[-4000:])
    print("\n" + json.dumps(rec, indent=1))
    assert rec["backend"] == "nccl" and rec["world_size"] == 1
    assert rec["flat_pair"]["params"] >= 113_673_219
    assert set(rec["modes"]) == {"overlap", "serial", "serial_sharded"}
    assert rec["modes"]["overlap"]["calls"]["all_reduce"] == 4


def test_hipgraph_capture_of_a_step_with_its_collective_is_attempted_and_recorded():
    """(iv) What SISS_GRAPH_DP=1 asks of bench.py: capture ONE step including the (serial) RCCL all-reduce into a hipGraph and
    replay it.  Whether RCCL under torch 2.10 / ROCm 7.2 allows that is the QUESTION: the test passes when the attempt produced a
    record -- captured + replayed equal to the no-group step, or the error it was refused with (DESIGN section 5 quotes it)."""
    r, rec = _run("graph", 600)
    assert rec is not None, (r.returncode, r.stdout[-3000:], r.stderr[-4000:])
    print("\n" + json.dumps(rec, indent=1))
    gc = rec["graph_capture"]
    assert gc["attempted"]
    assert gc.get("replay_matches_no_group_step") or gc.get("error"), gc
    # (v) bench.py's N > 1 autotune candidate: captured -> the stepper stays on the serial exchange and its state is put back;
    # refused -> the previous mode is restored (in process)
    if gc.get("replayed"):
        cand = rec["captured_serial_candidate"]
        assert (cand["captured"] and cand["state_restored"] and cand["mode_after"] == "serial/allreduce") or \
               (not cand["captured"] and cand["error"] and cand["mode_after"] == "overlap/allreduce"), cand
