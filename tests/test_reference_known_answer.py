"""The reference's only numeric known-answer for this path: the tutorial console output for the
REAL BAL file ladybug/problem-49-7776-pre.txt (reference docs/PoBATutorial.md:151-169):

    ./bin/bal --input data/rootba/bal/ladybug/problem-49-7776-pre.txt
    ...
    Final Cost: error: 1.3650e+04 (mean res: 0.59, num: 31843), error valid: 1.3621e+04 (..., num: 31812)
    NO_CONVERGENCE: Solver did not converge after maximum number of 20 iterations

i.e. defaults (solver_options.hpp / bal_dataset_options.hpp: SQUARE_ROOT, SCHUR_JACOBI, double,
squared norm, normalize to scale 100, no perturbation, no depth filter, 20 iterations).
The BAL data is not part of this image (no network): both tests SKIP unless the file is found at
$RBA_BAL_DATA/ladybug/..., ../rootba_data/bal/ladybug/... (scripts/download-bal-problems.sh:29) or
data/rootba/bal/ladybug/... — on any box that has it they pin the oracle and the HIP path to the
reference's own printed numbers. The tutorial says "a similar result" and its log shows 10 PCG
iterations per step (not reproducible from the shipped default of 500), hence 4 printed digits are
compared with 1 % slack on the cost and exactly on the observation counts.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = os.path.join("ladybug", "problem-49-7776-pre.txt")
CANDIDATES = [os.path.join(os.environ.get("RBA_BAL_DATA", "/nonexistent"), NAME),
              os.path.join(ROOT, "..", "rootba_data", "bal", NAME),
              os.path.join(ROOT, "data", "rootba", "bal", NAME)]
PATH = next((p for p in CANDIDATES if os.path.exists(p)), None)
needs_data = pytest.mark.skipif(PATH is None, reason="real BAL ladybug-49 file not present (no network in this image)")

FINAL_COST, FINAL_COST_VALID, N_OBS, N_VALID = 1.3650e4, 1.3621e4, 31843, 31812


def _problem():
    from rootba_amd import problem as P
    raw = P.read_bal(PATH)
    assert (raw.n_cams, raw.n_lms, raw.n_obs) == (49, 7776, N_OBS)
    return P.preprocess(raw, normalization_scale=100.0, rotation_sigma=0.0, translation_sigma=0.0,
                        point_sigma=0.0, init_depth_threshold=0.0)


def _check(log, term):
    assert term == 0 and log[-1].iteration == 20  # NO_CONVERGENCE after 20 iterations
    last = [r for r in log if r.step_is_successful][-1]
    assert last.num_obs == N_OBS and last.num_obs_valid == N_VALID
    assert abs(last.cost - FINAL_COST) < 1e-2 * FINAL_COST
    assert abs(last.cost_valid - FINAL_COST_VALID) < 1e-2 * FINAL_COST_VALID


@needs_data
def test_oracle_reproduces_the_tutorial_run():
    from oracle import oracle as O
    o = O.Oracle(_problem(), np.float64, O.default_options())
    _check(*o.optimize_lm())


@needs_data
@pytest.mark.gpu
def test_hip_reproduces_the_tutorial_run():
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    g = LinearizorHIP(_problem(), np.float64, L.default_options())
    _check(*g.optimize_lm())
