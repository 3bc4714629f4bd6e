"""The HIP library (through the C ABI) against THE REFERENCE'S OWN CODE, directly: the checker here is
oracle/_ref/librootba_ref.so (the reference's hot-path sources compiled unmodified against third-party stand-ins,
see oracle/ref_driver.cpp), which travels to the GPU box with the snapshot. Same structure and tolerances as the
HIP-vs-oracle tests of tests/test_gpu_parity.py; in float the LM-row cost tolerance is the float resolution of
the cost of a COMPUTED state (1e-4; two float CPU implementations already differ by 1e-5 there).
Skipped when the prebuilt library is absent.

`RBA_TEST_PRODUCT=oracle` runs the same bodies with the CPU oracle in the product's place (a dry run of this
file's logic on a machine without a GPU; not part of any suite).
"""
import os

import numpy as np
import pytest

from conftest import rel_err

DRY = os.environ.get("RBA_TEST_PRODUCT") == "oracle"
pytestmark = [] if DRY else [pytest.mark.gpu]

TOL = {np.float32: 1e-4, np.float64: 1e-10}
DT = [np.float32, np.float64]


@pytest.fixture(scope="module")
def R():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/librootba_ref.so is not present")
    return ref


@pytest.fixture(scope="module")
def ladybug_far():
    from rootba_amd import problem as P
    return P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)


def _product(prob, dtype, **kw):
    if DRY:
        from oracle import oracle as O
        okw = {k: v for k, v in kw.items() if k not in ("implicit_q", "explicit_after")}
        return O.Oracle(prob, dtype, O.default_options(**okw))
    import torch  # noqa: F401  (HIP runtime first, as in bench.py)
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    return LinearizorHIP(prob, dtype, L.default_options(**kw))


def _pair(R, prob, dtype, **kw):
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    # LinearizorQR's mapping (linearizor_qr.cpp:58-68): the validity flag of the blocks follows optimized_cost
    base["use_valid_projections_only"] = int(base.get("optimized_cost", 0) != 0)
    rkw = {k: v for k, v in base.items() if k not in ("implicit_q", "explicit_after")}
    return _product(prob, dtype, **base), R.Reference(prob, dtype, R.default_options(**rkw))


def _reference_iterate(R, prob, dtype, state, lam, n_it, **kw):
    """The reference's PCG iterate after EXACTLY n_it iterations from `state` (eta = 0 switches the Q-model
    stopping test off, conjugate_gradient.hpp:263-276)."""
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    base["use_valid_projections_only"] = int(base.get("optimized_cost", 0) != 0)
    rkw = {k: v for k, v in base.items() if k not in ("implicit_q", "explicit_after")}
    r = R.Reference(prob, dtype, R.default_options(max_cg_it=n_it, eta=0.0, **rkw))
    r.set_state(*state)
    assert r.linearize() == 0
    inc, cg = r.solve(lam)
    assert cg.num_iterations == n_it
    return inc


def _assert_increment(R, prob, dtype, r, lam, ig, cg, ir, cr, tol, **kw):
    """Unconditional: when the two truncated solves stop one iteration apart (the Q-model test is marginal, float
    only), the product's increment is compared with the reference's iterate of the SAME iteration count."""
    assert abs(cg.num_iterations - cr.num_iterations) <= (1 if dtype == np.float32 else 0)
    ref = ir if cg.num_iterations == cr.num_iterations else \
        _reference_iterate(R, prob, dtype, r.get_state(), lam, cg.num_iterations, **kw)
    assert rel_err(ig, ref) < tol, (rel_err(ig, ref), cg.num_iterations, cr.num_iterations)


@pytest.mark.parametrize("dtype", DT)
def test_compute_error(R, small_problem, dtype):
    g, r = _pair(R, small_problem, dtype)
    a, b = g.compute_error(), r.compute_error()
    assert (a.all_num_obs, a.valid_num_obs) == (b.all_num_obs, b.valid_num_obs)
    t = 1e-6 if dtype == np.float32 else 1e-13
    assert abs(a.all_error - b.all_error) / b.all_error < t
    assert abs(a.valid_error - b.valid_error) / b.valid_error < t


VARIANTS = [("sqrt-schur_jacobi", dict()), ("sqrt-jacobi", dict(preconditioner_type=0)),
            ("sqrt-squared-norm", dict(robust_norm=0)),
            ("sc-schur_jacobi", dict(solver_type=1)), ("sc-power", dict(solver_type=1, preconditioner_type=2, power_order=5))]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("name,kw", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_solve_and_apply(R, small_problem, dtype, name, kw):
    """One LM iteration's linear algebra from the same state: the reference's Linearizor{QR,SC} vs the HIP path."""
    tol = TOL[dtype]
    g, r = _pair(R, small_problem, dtype, **kw)
    assert g.linearize() == 0 and r.linearize() == 0
    ig, cg = g.solve(1e-4)
    ir, cr = r.solve(1e-4)
    assert cg.termination_type == cr.termination_type == 1
    _assert_increment(R, small_problem, dtype, r, 1e-4, ig, cg, ir, cr, 10 * tol, **kw)
    lg, lr = g.apply(ir), r.apply(ir)  # the reference's increment on both sides
    assert abs(lg - lr) / (abs(lg) + abs(lr)) < tol
    (cg_, lg_), (cr_, lr_) = g.get_state(), r.get_state()
    assert rel_err(cg_, cr_) < tol and rel_err(lg_, lr_) < tol
    eg, er = g.compute_error(), r.compute_error()
    assert abs(eg.all_error - er.all_error) <= (1e-4 if dtype == np.float32 else 1e-10) * er.all_error


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("optimized_cost", [1, 2], ids=["ERROR_VALID", "ERROR_VALID_AVG"])
def test_invalid_projections(R, small_problem, dtype, optimized_cost):
    """A camera turned away from its points: with the validity-aware costs its observations are dropped
    (zero Jacobian block, Jp_diag2 = 0, preconditioner block lambda I)."""
    from rootba_amd.problem import BalProblem
    p = small_problem
    cams = p.cams.copy()
    x, y, z, w = cams[3, :4]
    cams[3, :4] = [w, -z, y, -x]  # (1, 0, 0, 0) (x) q: a turn by 180 degrees about the camera's x axis
    cams[3, 4:7] *= np.array([1.0, -1.0, -1.0])
    prob = BalProblem(cams, p.lms.copy(), p.lm_obs_offsets, p.obs_cam_idx, p.obs_xy, "camera-3-looks-away")
    tol = TOL[dtype]
    g, r = _pair(R, prob, dtype, optimized_cost=optimized_cost)
    a, b = g.compute_error(), r.compute_error()
    assert (a.all_num_obs, a.valid_num_obs) == (b.all_num_obs, b.valid_num_obs) and b.valid_num_obs < b.all_num_obs
    assert abs(a.valid_error - b.valid_error) <= (1e-5 if dtype == np.float32 else 1e-12) * b.valid_error
    assert g.linearize() == 0 and r.linearize() == 0
    ig, cg = g.solve(1e-2)
    ir, cr = r.solve(1e-2)
    _assert_increment(R, prob, dtype, r, 1e-2, ig, cg, ir, cr, 10 * tol, optimized_cost=optimized_cost)
    lg, lr = g.apply(ir), r.apply(ir)
    assert abs(lg - lr) / (abs(lg) + abs(lr)) < tol
    assert rel_err(g.get_state()[1], r.get_state()[1]) < tol


@pytest.mark.parametrize("dtype", DT)
def test_lm_run(R, ladybug_far, dtype):
    """Whole LM runs at the ladybug-49 size (BASELINE configs 0-1), 12 iterations with the stopping rule off: the
    reference's optimize_lm_ours vs the library's LM driver - same accept / reject decisions and CG counts while
    the steps are large, the same final cost."""
    kw = dict(max_num_iterations=12, function_tolerance=0.0)
    g, r = _pair(R, ladybug_far, dtype, **kw)
    lg, tg = g.optimize_lm()
    lr, tr = r.optimize_lm()
    for a, b in zip(lg[:5], lr[:5]):
        assert bool(a.step_is_successful) == bool(b.step_is_successful)
        assert abs(a.cg_iterations - b.cg_iterations) <= (1 if dtype == np.float32 else 0)
        assert abs(a.cost - b.cost) <= (1e-4 if dtype == np.float32 else 1e-10) * b.cost
    fg = min(x.cost for x in lg if x.step_is_successful)
    fr = min(x.cost for x in lr if x.step_is_successful)
    if dtype == np.float64:
        assert len(lg) == len(lr) and tg == tr
        assert abs(fg - fr) / fr < 1e-9
        assert rel_err(g.get_state()[0], r.get_state()[0]) < 1e-6
    else:
        # float resolution of the cost on this problem (tests/test_gpu_parity.py::test_lm_trajectory_matches_oracle)
        r64 = R.Reference(ladybug_far, np.float64, R.default_options(robust_norm=1, huber_parameter=1.0, **kw))
        f64 = min(x.cost for x in r64.optimize_lm()[0] if x.step_is_successful)
        assert abs(fg - f64) / f64 < 2e-6 and abs(fr - f64) / f64 < 2e-6


@pytest.mark.parametrize("dtype", DT)
def test_per_landmark_qr_at_ladybug_size(R, ladybug_far, dtype):
    """BASELINE configs[1]: "ladybug problem-49-7776 float32, per-landmark QR kernel only vs CPU" - the CPU here is the
    reference's own LandmarkBlock code: Jl column scales, R^T R and |Q1^T r| of every landmark (invariant to the
    reflector conventions), and the stage-1 output Jp_diag2 through the pose scaling."""
    if DRY:
        pytest.skip("the oracle has no landmark_R accessor")
    tol = TOL[dtype]
    g, r = _pair(R, ladybug_far, dtype)
    st, d2 = g.linearize(want_jp_diag2=True)
    assert st == 0
    rc, d_ref, _ = r.stage1()
    assert rc == 0 and rel_err(d2, d_ref) < tol
    assert rel_err(g.jl_col_scale(), r.jl_col_scale()) < tol
    Rg, qg = g.landmark_R(damped=False)
    worst = 0.0
    for l in range(0, ladybug_far.n_lms, 37):
        blk, li = r.block(l)
        Rr = np.triu(blk[:3, li:li + 3].astype(np.float64))
        Rl = np.zeros((3, 3))
        Rl[np.triu_indices(3)] = Rg[l]
        worst = max(worst, rel_err(Rl.T @ Rl, Rr.T @ Rr), rel_err(np.abs(qg[l]), np.abs(blk[:3, li + 3])))
    assert worst < 10 * tol
