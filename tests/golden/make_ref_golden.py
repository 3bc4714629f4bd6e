"""Generates tests/golden/ref_small.npz FROM THE REFERENCE'S OWN CODE.

    python tests/golden/make_ref_golden.py        (needs /root/reference; builds oracle/_ref first)

The numbers come from oracle/_ref/librootba_ref.so, i.e. the reference's hot-path sources compiled
unmodified against the third-party stand-ins of oracle/ref_shims/ (see oracle/ref_driver.cpp for what that
pins). The inputs are the fixed problem of tests/golden/small_f64.npz. The fixture travels to the GPU box,
where /root/reference does not exist: tests/test_reference_golden.py holds the oracle (CPU) and the HIP
library (GPU) to it.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref as R  # noqa: E402
from rootba_amd.problem import BalProblem  # noqa: E402

LAMBDA = 0.1
LAMBDA_SOLVE = 1e-4
OPTS = dict(robust_norm=1, huber_parameter=1.0)


def golden_problem():
    G = np.load(os.path.join(HERE, "small_f64.npz"))
    return BalProblem(G["cams"], G["lms"], G["lm_obs_offsets"], G["obs_cam_idx"], G["obs_xy"], "golden")


def eps_of(dt):
    # Sophus::Constants<Scalar>::epsilonSqrt(), the reference's effective jacobi_scaling_epsilon
    return 1e-5 if np.dtype(dt) == np.float64 else float(np.sqrt(np.float32(1e-5)))


def one(prob, dt, out, s):
    r = R.Reference(prob, dt, R.default_options(**OPTS))
    ri = r.compute_error()
    out["error" + s], out["error_valid" + s], out["residual_sum" + s] = ri.all_error, ri.valid_error, ri.all_residual_sum
    out["num_obs_valid" + s] = ri.valid_num_obs
    rc, d, jb = r.stage1(jacobi_blocks=True)
    assert rc == 0
    out["jp_diag2" + s], out["jacobi_blocks" + s] = d, jb.reshape(-1, 9, 9)
    out["jl_col_scale" + s] = r.jl_col_scale()
    scaling = (1.0 / (eps_of(dt) + np.sqrt(d.astype(np.float64)))).astype(dt)
    out["pose_scaling" + s] = scaling
    r.set_pose_damping(LAMBDA)
    b, blocks = r.stage2(LAMBDA, scaling, blocks=True)
    out["b" + s], out["blocks" + s] = b, blocks
    x = np.random.default_rng(5).uniform(-1, 1, 9 * prob.n_cams).astype(dt)
    out["x" + s], out["hx" + s] = x, r.right_multiply(x)
    inc_rand = (np.random.default_rng(6).uniform(-1, 1, 9 * prob.n_cams) * 0.01).astype(dt)
    out["inc_rand" + s], out["l_diff" + s] = inc_rand, r.back_substitute(inc_rand)
    out["lms_after" + s] = r.get_state()[1]

    for name, kw in (("", {}), ("_jacobi", dict(preconditioner_type=0)), ("_sc", dict(solver_type=1)),
                     ("_sc_power", dict(solver_type=1, preconditioner_type=2))):
        r2 = R.Reference(prob, dt, R.default_options(**OPTS, **kw))
        assert r2.linearize() == 0
        inc, cg = r2.solve(LAMBDA_SOLVE)
        out["inc" + name + s], out["cg_iterations" + name + s] = inc, cg.num_iterations
        if name == "":
            out["l_diff2" + s] = r2.apply(inc)
            c, l = r2.get_state()
            out["cams_after" + s], out["lms_after2" + s] = c, l
            out["error_after" + s] = r2.compute_error().all_error

    for name, kw in (("", {}), ("_sc", dict(solver_type=1))):
        r3 = R.Reference(prob, dt, R.default_options(max_num_iterations=10, **OPTS, **kw))
        log, term = r3.optimize_lm()
        out["lm_cost" + name + s] = np.array([q.cost for q in log])
        out["lm_cg" + name + s] = np.array([q.cg_iterations for q in log])
        out["lm_ok" + name + s] = np.array([q.step_is_successful for q in log])
        out["lm_term" + name + s] = term
        c, l = r3.get_state()
        out["lm_cams" + name + s], out["lm_lms" + name + s] = c, l
    return len(log), log[-1].cost


def main():
    if not os.path.isdir(os.path.join(R.REFERENCE_ROOT, "src", "rootba")):
        raise SystemExit("the reference tree is needed to regenerate this fixture")
    R.build(force=True)
    prob = golden_problem()
    out = dict(lam=LAMBDA, lam_solve=LAMBDA_SOLVE)
    for dt, s in ((np.float64, "_f64"), (np.float32, "_f32")):
        print(np.dtype(dt).name, one(prob, dt, out, s))
    np.savez_compressed(os.path.join(HERE, "ref_small.npz"), **out)


if __name__ == "__main__":
    main()
