"""Generates tests/golden/*.npz.

There are NO golden vectors in the reference and the reference
cannot be built here, so these fixtures are produced by the CPU oracle
(oracle/, float64) on a fixed seeded problem. They pin (a) the oracle against
silent regressions and (b) the HIP path against committed numbers on the GPU
box, where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from rootba_amd import problem as P  # noqa: E402

LAMBDA = 0.1


def golden_problem():
    raw = P.synthetic_problem(24, 300, 1300, seed=11)
    return P.preprocess(raw, seed=11, translation_sigma=0.5, point_sigma=0.5)


def main():
    prob = golden_problem()
    opts = dict(robust_norm=1, huber_parameter=1.0)
    o = O.Oracle(prob, np.float64, O.default_options(**opts))
    ri = o.compute_error()
    assert o.linearize() == 0
    jp_diag2, scaling = None, o.pose_scaling()
    o.set_pose_damping(LAMBDA)
    b, blocks = o.stage2(LAMBDA, scaling, blocks=True)
    x = np.random.default_rng(5).uniform(-1, 1, 9 * prob.n_cams)
    hx = o.right_multiply(x)
    inc_rand = np.random.default_rng(6).uniform(-1, 1, 9 * prob.n_cams) * 0.01
    l_diff = o.back_substitute(inc_rand)
    lms_after = o.get_state()[1]

    o2 = O.Oracle(prob, np.float64, O.default_options(**opts))
    assert o2.linearize() == 0
    inc, cg = o2.solve(1e-4)
    l_diff2 = o2.apply(inc)
    cams_after = o2.get_state()[0]

    o3 = O.Oracle(prob, np.float64, O.default_options(max_num_iterations=10, **opts))
    log, term = o3.optimize_lm()
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_f64.npz"),
        cams=prob.cams, lms=prob.lms, lm_obs_offsets=prob.lm_obs_offsets, obs_cam_idx=prob.obs_cam_idx,
        obs_xy=prob.obs_xy, error=ri.all_error, error_valid=ri.valid_error, residual_sum=ri.all_residual_sum,
        pose_scaling=scaling, jl_col_scale=o.jl_col_scale(), b=b, blocks=blocks, x=x, hx=hx,
        inc_rand=inc_rand, l_diff=l_diff, lms_after=lms_after, inc=inc, cg_iterations=cg.num_iterations,
        l_diff2=l_diff2, cams_after=cams_after,
        lm_cost=np.array([r.cost for r in log]), lm_cg=np.array([r.cg_iterations for r in log]),
        lm_ok=np.array([r.step_is_successful for r in log]), lm_inc_norm=np.array([r.inc_norm for r in log]),
        lm_lambda=np.array([r.lambda_ for r in log]), lm_term=term, lam=LAMBDA)
    print("written", len(log), "LM rows, final cost", log[-1].cost)


if __name__ == "__main__":
    main()
