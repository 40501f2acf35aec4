"""The C ABI used WITHOUT Python ("plain extern "C" launchers ... so kernels can be unit-benchmarked
without Python"): tools/cabi_example.cpp is compiled against include/siss_hip.h, linked to libsiss_hip.so and run --
a 3x3 convolution through siss_gemm_nt checked against a scalar CPU loop, and siss_mixture_fwd's importance-weight
invariant, on plain hipMalloc buffers."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_example_builds_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is part of the image"
    lib_dir = os.path.join(ROOT, "siss_amd")
    assert os.path.exists(os.path.join(lib_dir, "libsiss_hip.so")), "build the library first (python -m siss_amd.build)"
    exe = str(tmp_path / "cabi_example")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tools", "cabi_example.cpp"), "-L" + lib_dir, "-lsiss_hip",
                        "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cabi example ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
