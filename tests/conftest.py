import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


if os.environ.get("RBA_EMU") == "1":
    # Development tool, NOT a backend: run the product's kernels on the CPU in the execution harness of tests/hipemu
    # (see tests/hipemu/hip/hip_runtime.h) - e.g. `RBA_EMU=1 python -m pytest tests -m gpu -k small`. The product package
    # is untouched; only this test session points its loader at the harness build.
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu  # noqa: E402
    import rootba_amd._lib as _rba_lib  # noqa: E402
    _rba_lib.LIB_PATH = build_emu.build()
    # (the harness build of the library opens this stand-in where the product opens librccl.so.1)
    os.environ.setdefault("HIPEMU_RCCL", os.path.join(os.path.dirname(_rba_lib.LIB_PATH), "fake_rccl", "librccl.so.1"))


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RBA_EMU") != "1":
        return
    # the CPU harness is for logic at test sizes: full-size workloads would take hours, the CLI binary and the
    # multi-process tests use the real library
    too_big = ("test_full_size_venice", "test_gpu_baseline_configs", "test_bal_qr_hip", "test_gpu_rccl_multi",
               "test_hip_reproduces_the_tutorial_run")
    if os.environ.get("RBA_EMU_CLI") == "1":
        # the caller put a copy of the harness build named librootba_hip.so on LD_LIBRARY_PATH: the CLI binary
        # (RUNPATH $ORIGIN, searched after LD_LIBRARY_PATH) then runs on the harness too
        too_big = tuple(t for t in too_big if t != "test_bal_qr_hip")
    skip = pytest.mark.skip(reason="not run on the CPU execution harness (size / real library needed)")
    for item in items:
        if any(t in item.nodeid for t in too_big):
            item.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_problem():
    """40 cameras / 400 landmarks, preprocessed like the reference pipeline."""
    from rootba_amd import problem as P
    raw = P.synthetic_problem(40, 400, 1700, seed=7)
    # start far from the optimum (like real BAL data) so that the gradient is
    # not a cancellation residue and f32 comparisons are meaningful
    return P.preprocess(raw, seed=7, translation_sigma=0.5, point_sigma=0.5)


@pytest.fixture(scope="session")
def ladybug_problem():
    """Synthetic stand-in for BAL ladybug problem-49-7776."""
    from rootba_amd import problem as P
    return P.preprocess(P.named_synthetic("ladybug-49"))


def rel_err(a, b):
    """Reference test metric |a-b| / (|a|+|b|) (src/rootba/testing/eigen_utils.hpp:105-108)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    s = np.linalg.norm(a) + np.linalg.norm(b)
    return 0.0 if s == 0 else d / s
