import os
import sys

import pytest

# The ORACLE side of the GPU parity tests is torch on the same GPU (MIOpen convolutions).  MIOpen's default find mode benchmarks every
# new convolution shape exhaustively on first use -- 250 s of the suite (the full-size fp32 / autocast oracles: 112 + 94 + 43 s) spent
# tuning kernels that run a handful of times.  FAST picks by heuristic; the oracle's numbers are the same, only its speed differs
# (tests/test_hip_vs_torch_rocm.py, which TIMES that stack, says so in its output).  Only a default: an exported value wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))          # tests/parity_util.py
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
