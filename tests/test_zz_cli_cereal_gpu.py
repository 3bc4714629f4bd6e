"""`bal_qr_hip --save-output` on the GPU: the `.cereal` problem cache written after a solve (kept in its own file, after the
other GPU tests: it was written when the round's GPU budget was spent and first runs in the round-end suite)."""
import json
import os
import subprocess

import numpy as np
import pytest

from rootba_amd import build
from rootba_amd import problem as P
from test_host_cpp import _cereal_parse


@pytest.fixture(scope="module")
def app():
    build.build()
    return build.APP


@pytest.fixture(scope="module")
def bal_file(tmp_path_factory):
    raw = P.synthetic_problem(16, 150, 600, seed=9)
    path = str(tmp_path_factory.mktemp("bal") / "problem-16-150-pre.txt")
    P.write_bal(raw, path)
    return path, raw


@pytest.mark.gpu
def test_bal_qr_hip_saves_the_optimised_problem(app, bal_file, tmp_path):
    """`--save-output` (BalProblem::postprocress, bal_problem.cpp:556-568): the `.cereal` file written after the solve
    holds the optimised state - its cost, evaluated through the Python binding, is the run's final cost."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    path, _ = bal_file
    log_path, cache = str(tmp_path / "ba_log.json"), str(tmp_path / "optimized.cereal")
    out = subprocess.run([app, "--input", path, "--max-num-iterations", "6", "--robust-norm", "HUBER", "--log-path", log_path,
                          "--save-output", "--output-optimized-path", cache], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    log = json.load(open(log_path))
    opt = _cereal_parse(open(cache, "rb").read())
    start = P.normalize(P.read_bal(path), 100.0)
    assert (opt.n_cams, opt.n_lms, opt.n_obs) == (start.n_cams, start.n_lms, start.n_obs)
    assert np.array_equal(opt.obs_cam_idx, start.obs_cam_idx) and np.array_equal(opt.obs_xy, start.obs_xy)
    assert not np.allclose(opt.lms, start.lms)
    g = LinearizorHIP(opt, np.float64, L.default_options(robust_norm=1))
    assert abs(g.compute_error().all_error - log["cost"][-1]) <= 1e-9 * log["cost"][-1]
    # and the cache is a valid input of the tool itself
    again = subprocess.run([app, "--input", cache, "--no-normalize", "--max-num-iterations", "1", "--robust-norm", "HUBER",
                            "--log-path", log_path], capture_output=True, text=True)
    assert again.returncode == 0, again.stderr
    assert abs(json.load(open(log_path))["cost"][0] - log["cost"][-1]) <= 1e-9 * log["cost"][-1]
