"""The kernels' LOGIC on a machine without a GPU: a few GPU tests re-run in a subprocess on the CPU execution harness of
tests/hipemu (the product's solver.hip and kernel headers compiled unchanged as C++; a development tool, not a
backend - see tests/hipemu/README.md). The subprocess keeps the harness build out of this test session, whose loader
stays on the real library. Skipped when the harness cannot be built here (it needs the ROCm clang++)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_slice_of_the_gpu_suite_on_the_cpu_harness():
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    if not os.path.exists(build_emu.CXX):
        pytest.skip(f"{build_emu.CXX} is not available")
    try:
        build_emu.build()
    except subprocess.CalledProcessError as e:
        pytest.skip(f"the harness does not build here: {e}")
    env = dict(os.environ, RBA_EMU="1")
    sel = ("test_compute_error or (test_solve_and_apply and sqrt-schur_jacobi) or (test_stage2_variants and small and float32) "
           "or (test_invalid_projections and ERROR_VALID-float32)")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-k", sel,
                          os.path.join(ROOT, "tests", "test_reference_gpu.py"),
                          os.path.join(ROOT, "tests", "test_zz_candidates_gpu.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
