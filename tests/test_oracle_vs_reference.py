"""The restated CPU oracle (oracle/rootba_oracle.hpp) against THE REFERENCE'S OWN SOURCES.

oracle/build_ref.sh compiles the reference's hot-path translation units, unmodified and where they lie
under /root/reference/src/rootba, against stand-ins for the third-party libraries that are missing on
this machine (oracle/ref_shims/). oracle/ref.py drives that build with the interface of oracle.Oracle.
These tests therefore pin the restatement against the reference's code for: geometry + Jacobians call
path, Huber weights, cost accumulation, landmark-block layout, Jl / Jp column scaling, Householder and
Givens marginalisation, landmark damping (incl. re-damping), stage 1 / stage 2 reductions, both H*x
reductions, SCHUR_JACOBI / JACOBI / power-series preconditioners, the Ceres-style PCG, back-substitution
and l_diff, camera retraction, the explicit-SC solver, the LM loop, and the BAL loader + normalisation +
filtering. What is NOT pinned is the arithmetic inside the stand-ins (Eigen's Householder / Givens / LLT
kernels, Sophus' SO3::exp, basalt's camera model), which are restated from their published definitions -
see the header of oracle/ref_driver.cpp.

Tolerances are the reference's own (src/rootba/qr/linearization_qr.test.cpp:125, 174-211: 1e-5 float,
1e-12 double on |a-b| / (|a|+|b|)), tightened where the measured agreement allows.
Skipped when neither /root/reference nor a prebuilt oracle/_ref/librootba_ref.so is present.
"""
import numpy as np
import pytest

from conftest import rel_err

DT = [np.float64, np.float32]


def tol(dt, f64, f32):
    return f64 if np.dtype(dt) == np.float64 else f32


@pytest.fixture(scope="module")
def R():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not present")
    return ref


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def both(O, R, prob, dt, **kw):
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    # LinearizorQR's option mapping (linearizor_qr.cpp:58-68): the blocks' validity flag follows optimized_cost
    base["use_valid_projections_only"] = int(base.get("optimized_cost", 0) != 0)
    return O.Oracle(prob, dt, O.default_options(**base)), R.Reference(prob, dt, R.default_options(**base))


def test_default_options_are_the_references(O, R):
    """Every default of the option struct of this repository (include/rootba_hip.h, oracle) equals the
    default in the reference's own SolverOptions declaration (src/rootba/bal/solver_options.hpp)."""
    a, b = O.default_options(), R.default_options()
    for name, _ in O.Options._fields_:
        if name in ("implicit_q", "explicit_after", "num_threads"):  # product-only switches / thread count
            continue
        assert getattr(a, name) == getattr(b, name), name


@pytest.mark.parametrize("dt", DT)
def test_linearize_point(O, R, dt):
    """BalBundleAdjustmentHelper::linearize_point (bal_bundle_adjustment_helper.cpp:111-149) incl. points
    behind the camera and the validity flag with / without the check."""
    rng = np.random.default_rng(5)
    n_invalid = 0
    for i in range(200):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        cam = np.concatenate([q, rng.standard_normal(3), [500.0 + 100 * rng.random(), 1e-2 * rng.standard_normal(),
                                                            1e-3 * rng.standard_normal()]])
        p_w = rng.standard_normal(3) * 3
        obs = rng.standard_normal(2) * 50
        for ignore in (True, False):
            va, ra, jpa, jia, jla = O.linearize_point(obs, p_w, cam, dt, ignore)
            vb, rb, jpb, jib, jlb = R.linearize_point(obs, p_w, cam, dt, ignore)
            assert va == vb
            n_invalid += not vb
            if ignore or vb:
                t = tol(dt, 1e-13, 2e-5)
                assert rel_err(ra, rb) < t and rel_err(jpa, jpb) < t and rel_err(jia, jib) < t and rel_err(jla, jlb) < t
    assert n_invalid > 20


@pytest.mark.parametrize("dt", DT)
def test_camera_retraction(O, R, dt):
    """Camera::apply_inc_pose / apply_inc_intrinsics (bal_problem.hpp:97-109): decoupled SE(3) step."""
    rng = np.random.default_rng(6)
    for scale in (1e-8, 1e-3, 0.3, 2.0):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        cam = np.concatenate([q, rng.standard_normal(3), [400.0, 0.01, -0.001]])
        inc = rng.standard_normal(9) * scale
        assert rel_err(O.apply_inc_camera(cam, inc, dt), R.apply_inc_camera(cam, inc, dt)) < tol(dt, 1e-15, 2e-7)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kw", [dict(), dict(robust_norm=0), dict(optimized_cost=1), dict(huber_parameter=0.3)],
                         ids=["huber", "squared", "valid-only", "huber0.3"])
def test_compute_error(O, R, small_problem, dt, kw):
    o, r = both(O, R, small_problem, dt, **kw)
    a, b = o.compute_error(), r.compute_error()
    assert (a.all_num_obs, a.valid_num_obs, a.is_numerically_valid) == (b.all_num_obs, b.valid_num_obs, b.is_numerically_valid)
    t = tol(dt, 1e-13, 1e-6)
    assert abs(a.all_error - b.all_error) <= t * b.all_error
    assert abs(a.valid_error - b.valid_error) <= t * b.valid_error
    assert abs(a.all_residual_sum - b.all_residual_sum) <= t * b.all_residual_sum


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kw", [dict(), dict(use_householder=0), dict(robust_norm=0), dict(optimized_cost=1),
                                dict(reduction_alg=0), dict(jacobi_scaling_eps=1.0)],
                         ids=["default", "givens", "squared", "valid-only", "reduce-v0", "eps1"])
def test_stage1_stage2_product_backsubstitution(O, R, small_problem, dt, kw):
    """LinearizationQR stage by stage (linearization_qr.hpp:634-815, 406-429, 823-825, 165-179)."""
    prob = small_problem
    o, r = both(O, R, prob, dt, **kw)
    t = tol(dt, 1e-12, 1e-5)
    rc_a, d_a, bl_a = o.stage1(jacobi_blocks=True)
    rc_b, d_b, bl_b = r.stage1(jacobi_blocks=True)
    assert rc_a == rc_b == 0
    assert rel_err(d_a, d_b) < t and rel_err(bl_a, bl_b) < t
    assert rel_err(o.jl_col_scale(), r.jl_col_scale()) < t
    # the marginalised blocks: R^T R, |Q1^T r| and the Gram matrix of Q2^T [Jp r] do not depend on the
    # reflector conventions
    for l in (0, 7, prob.n_lms - 1):
        (A, li), (B, lj) = o.block(l), r.block(l)
        assert A.shape == B.shape and li == lj
        Ra, Rb = np.triu(A[:3, li:li + 3]).astype(np.float64), np.triu(B[:3, li:li + 3]).astype(np.float64)
        assert rel_err(Ra.T @ Ra, Rb.T @ Rb) < t
        Qa = np.delete(A[3:-3], np.s_[li:li + 3], axis=1).astype(np.float64)
        Qb = np.delete(B[3:-3], np.s_[li:li + 3], axis=1).astype(np.float64)
        assert rel_err(Qa.T @ Qa, Qb.T @ Qb) < 10 * t
    eps = float(kw.get("jacobi_scaling_eps", 0.0)) or (1e-5 if np.dtype(dt) == np.float64 else float(np.sqrt(np.float32(1e-5))))
    scaling = (1.0 / (eps + np.sqrt(d_b.astype(np.float64)))).astype(dt)
    x = np.random.default_rng(1).standard_normal(9 * prob.n_cams)
    first = True
    for lam in (1e-3, 0.0, 10.0, 1e-8):  # re-damping without a new linearisation (undo + redo of the Givens sequence)
        o.set_pose_damping(lam)
        r.set_pose_damping(lam)
        b_a, s_a = o.stage2(lam, scaling if first else None)
        b_b, s_b = r.stage2(lam, scaling if first else None)
        first = False
        assert rel_err(b_a, b_b) < t and rel_err(s_a, s_b) < t
        assert rel_err(o.right_multiply(x), r.right_multiply(x)) < t
    inc = 1e-3 * x
    la, lb = o.back_substitute(inc), r.back_substitute(inc)
    assert abs(la - lb) <= tol(dt, 1e-12, 1e-5) * abs(lb)
    assert rel_err(o.get_state()[1], r.get_state()[1]) < tol(dt, 1e-14, 1e-6)


VARIANTS = [("sqrt-schur_jacobi", dict()), ("sqrt-jacobi", dict(preconditioner_type=0)),
            ("sqrt-unstaged", dict(staged_execution=0)), ("sqrt-givens", dict(use_householder=0)),
            ("sqrt-squared", dict(robust_norm=0)), ("sqrt-valid", dict(optimized_cost=1)),
            ("sc-schur_jacobi", dict(solver_type=1)), ("sc-power", dict(solver_type=1, preconditioner_type=2)),
            ("sc-power3", dict(solver_type=1, preconditioner_type=2, power_order=3))]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("name,kw", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_linearize_solve_apply(O, R, small_problem, dt, name, kw):
    """Linearizor{QR,SC}::{linearize, solve, apply} through the reference's factory
    (linearizor.cpp:48-70; linearizor_qr.cpp:78-291; linearizor_sc.cpp; conjugate_gradient.hpp:113-298)."""
    o, r = both(O, R, small_problem, dt, **kw)
    assert o.linearize() == 0 and r.linearize() == 0
    for lam in (1e-4, 1e-2):  # second solve = backtracking on the same linearisation point
        ia, ca = o.solve(lam)
        ib, cb = r.solve(lam)
        assert (ca.num_iterations, ca.termination_type) == (cb.num_iterations, cb.termination_type)
        assert rel_err(ia, ib) < tol(dt, 1e-11, 1e-4)
    o.backup()
    r.backup()
    la, lb = o.apply(ib), r.apply(ib)
    assert abs(la - lb) <= tol(dt, 1e-12, 1e-5) * abs(lb)
    (ca_, la_), (cb_, lb_) = o.get_state(), r.get_state()
    assert rel_err(ca_, cb_) < tol(dt, 1e-15, 1e-6) and rel_err(la_, lb_) < tol(dt, 1e-14, 1e-6)
    ea, eb = o.compute_error(), r.compute_error()
    assert abs(ea.all_error - eb.all_error) <= tol(dt, 1e-12, 1e-4) * eb.all_error
    o.restore()
    r.restore()
    assert rel_err(o.get_state()[0], r.get_state()[0]) == 0.0 and np.array_equal(r.get_state()[1], small_problem.lms.astype(dt))


LM_VARIANTS = [("sqrt", dict()), ("sqrt-jacobi", dict(preconditioner_type=0)), ("sc", dict(solver_type=1)),
               ("sc-power", dict(solver_type=1, preconditioner_type=2)), ("sqrt-valid-avg", dict(optimized_cost=2))]


@pytest.mark.parametrize("name,kw", LM_VARIANTS, ids=[v[0] for v in LM_VARIANTS])
def test_lm_loop_float64(O, R, small_problem, name, kw):
    """optimize_lm_ours (bal_bundle_adjustment.cpp:249-544), double: same number of iterations, same
    accept / reject decisions, same termination; CG counts and costs identical while the solves are
    well determined (the first four iterations: 1e-12), close afterwards (the Q-model stopping test amplifies
    rounding differences of the two H*x summation orders)."""
    o, r = both(O, R, small_problem, np.float64, max_num_iterations=12, **kw)
    la, ta = o.optimize_lm()
    lb, tb = r.optimize_lm()
    assert ta == tb and len(la) == len(lb)
    for i, (a, b) in enumerate(zip(la, lb)):
        assert (a.iteration, a.step_is_successful, a.step_is_valid) == (b.iteration, b.step_is_successful, b.step_is_valid)
        assert (a.num_obs, a.num_obs_valid) == (b.num_obs, b.num_obs_valid)
        if i <= 3:
            assert a.cg_iterations == b.cg_iterations
            assert abs(a.cost - b.cost) <= 1e-12 * b.cost
        else:
            assert abs(a.cg_iterations - b.cg_iterations) <= max(3, b.cg_iterations // 5)
            assert abs(a.cost - b.cost) <= 1e-6 * b.cost
        if i > 0 and i + 1 < len(la):
            # rows of this repository report the damping the iteration USED, the reference's
            # trust_region_radius is the one for the NEXT iteration
            assert abs(la[i + 1].lambda_ - b.lambda_) <= 1e-6 * b.lambda_ or not a.step_is_successful or i > 3
    assert abs(la[-1].cost - lb[-1].cost) <= 1e-8 * lb[-1].cost
    assert rel_err(o.get_state()[0], r.get_state()[0]) < 1e-5


def test_lm_loop_float32(O, R, small_problem):
    """float: the first iterations agree to float accuracy; both runs end at the same optimum (the
    trajectories separate once cost changes reach the float resolution of the cost itself)."""
    o, r = both(O, R, small_problem, np.float32, max_num_iterations=12)
    la, _ = o.optimize_lm()
    lb, _ = r.optimize_lm()
    for a, b in zip(la[:4], lb[:4]):
        assert (a.iteration, a.step_is_successful, a.cg_iterations) == (b.iteration, b.step_is_successful, b.cg_iterations)
        assert abs(a.cost - b.cost) <= 5e-5 * b.cost
    fa = min(x.cost for x in la if x.step_is_successful)
    fb = min(x.cost for x in lb if x.step_is_successful)
    assert abs(fa - fb) <= 1e-5 * fb


def test_lm_loop_rejected_steps(O, R, small_problem):
    """A run that backtracks: a huge initial trust region on a far start makes the first steps fail; the
    lambda / vee bookkeeping and restore() path follow the reference iteration by iteration."""
    from rootba_amd import problem as P
    far = P.preprocess(P.synthetic_problem(20, 150, 600, seed=11), seed=11, translation_sigma=3.0, point_sigma=3.0,
                       rotation_sigma=0.3)
    o, r = both(O, R, far, np.float64, max_num_iterations=10, initial_trust_region_radius=1e12)
    la, ta = o.optimize_lm()
    lb, tb = r.optimize_lm()
    assert ta == tb and len(la) == len(lb)
    assert any(not b.step_is_successful for b in lb[1:]), "the scenario is meant to contain rejected steps"
    for a, b in zip(la, lb):
        assert (a.iteration, a.step_is_successful, a.step_is_valid) == (b.iteration, b.step_is_successful, b.step_is_valid)
        assert abs(a.cost - b.cost) <= 1e-6 * b.cost


def test_ladybug_size_float64(O, R, ladybug_problem):
    """BASELINE config 1 size (49 cameras, 7776 landmarks, 31843 observations), double, one LM iteration
    stage by stage."""
    o, r = both(O, R, ladybug_problem, np.float64)
    assert o.linearize() == 0 and r.linearize() == 0
    ia, ca = o.solve(1e-4)
    ib, cb = r.solve(1e-4)
    assert ca.num_iterations == cb.num_iterations and rel_err(ia, ib) < 1e-10
    la, lb = o.apply(ib), r.apply(ib)
    assert abs(la - lb) <= 1e-12 * abs(lb)
    ea, eb = o.compute_error(), r.compute_error()
    assert abs(ea.all_error - eb.all_error) <= 1e-12 * eb.all_error


def test_bal_loader_normalisation_and_filter(R, tmp_path):
    """BalProblem::load_bal + normalize + filter_obs of the reference (bal_problem.cpp:189-282, 428-506)
    against this repository's host-side mirror (rootba_amd/problem.py, which tests/test_host_cpp.py holds the
    C++ loader to): axis convention, quaternion, y-flip of the observations, median / MAD normalisation,
    depth filter, landmark removal."""
    from rootba_amd import problem as P
    raw = P.synthetic_problem(16, 150, 600, seed=9)
    path = str(tmp_path / "problem-16-150-pre.txt")
    P.write_bal(raw, path)
    for normalize, thr in ((False, 0.0), (True, 0.0), (True, 60.0)):
        ref = R.load_bal(path, normalize=normalize, init_depth_threshold=thr)
        mine = P.read_bal(path)
        if normalize:
            mine = P.normalize(mine, 100.0)
        mine = P.filter_obs(mine, thr)
        assert np.array_equal(ref["lm_obs_offsets"], mine.lm_obs_offsets)
        assert np.array_equal(ref["obs_cam_idx"], mine.obs_cam_idx)
        assert np.array_equal(ref["obs_xy"], mine.obs_xy)
        assert rel_err(ref["lms"], mine.lms) < 1e-14
        assert rel_err(ref["cams"][:, 4:], mine.cams[:, 4:]) < 1e-13
        for qa, qb in zip(ref["cams"][:, :4], mine.cams[:, :4]):  # q and -q are the same rotation
            assert min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) < 1e-13
    assert mine.n_obs < raw.n_obs


# ---- the reference's own checks of the third-party boundary, repeated on the stand-ins --------------------
def test_standin_camera_matches_the_in_tree_projection_formula(R):
    """src/rootba/bal/snavely_projection.test.cpp:155-188 pins basalt::BalCamera::project against the
    reference's IN-TREE formula (snavely_projection.hpp:182-190: m = p.xy / p.z, r2 = |m|^2,
    proj = f (1 + r2 (k1 + r2 k2)) m) on the grid x, y in [-10, 10], z in [0, 5], wherever the projection
    reports success. The same check on the stand-in of oracle/ref_shims/basalt/camera/bal_camera.hpp,
    through the reference's linearize_point (identity pose, observation 0: the residual is the projection)."""
    cams = [(500.0, 0.0, 0.0), (718.856, -0.3, 0.1), (300.0, 1e-2, -1e-3), (1.0, 0.5, 0.25)]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
    n_ok = 0
    for dt, eps_sqrt in ((np.float64, 1e-5), (np.float32, float(np.sqrt(np.float32(1e-5))))):
        for f, k1, k2 in cams:
            cam = np.concatenate([ident, [f, k1, k2]])
            for x in range(-10, 11):
                for y in range(-10, 11):
                    for z in range(0, 6):
                        if z == 0:
                            continue  # division by zero: success is false, nothing is compared (reference: `if (success)`)
                        valid, res, *_ = R.linearize_point([0.0, 0.0], [x, y, z], cam, dt, ignore_validity_check=True)
                        assert valid == (z >= eps_sqrt)
                        m = np.array([x / z, y / z])
                        r2 = m @ m
                        want = f * (1.0 + r2 * (k1 + r2 * k2)) * m
                        assert np.linalg.norm(res - want) <= eps_sqrt * max(np.linalg.norm(res), np.linalg.norm(want)) + 1e-300
                        n_ok += 1
    assert n_ok == 2 * len(cams) * 21 * 21 * 5


def test_standin_jacobians_by_numeric_differentiation(R):
    """src/rootba/bal/bal_bundle_adjustment_helper.test.cpp:54-148: the analytic Jacobians of
    linearize_point against numeric differentiation through Camera::inc_pose (the decoupled SE(3) step),
    the intrinsics increment and the landmark - on the reference build with the stand-in camera / Sophus.
    Central differences, step 1e-6 (double)."""
    rng = np.random.default_rng(12)
    for _ in range(20):
        w = rng.uniform(-1, 1, 3) / 100
        th = np.linalg.norm(w)
        q = np.concatenate([np.sin(th / 2) / th * w, [np.cos(th / 2)]])
        cam = np.concatenate([q, rng.uniform(-1, 1, 3), [500.0 + 200 * rng.random(), 0.1 * rng.uniform(-1, 1), 0.01 * rng.uniform(-1, 1)]])
        p_w = rng.uniform(-1, 1, 3) + [0, 0, 10]
        obs = rng.uniform(-1, 1, 2) * 5
        valid, res, Jp, Ji, Jl = R.linearize_point(obs, p_w, cam, np.float64, ignore_validity_check=False)
        assert valid
        h = 1e-6
        num = np.zeros((2, 9))
        for k in range(9):
            d = np.zeros(9)
            d[k] = h
            rp = R.linearize_point(obs, p_w, R.apply_inc_camera(cam, d), np.float64, False)[1]
            rm = R.linearize_point(obs, p_w, R.apply_inc_camera(cam, -d), np.float64, False)[1]
            num[:, k] = (rp - rm) / (2 * h)
        assert np.allclose(num[:, :6], Jp, rtol=1e-6, atol=1e-6 * np.abs(Jp).max())
        assert np.allclose(num[:, 6:], Ji, rtol=1e-6, atol=1e-6 * np.abs(Ji).max())
        numl = np.zeros((2, 3))
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            numl[:, k] = (R.linearize_point(obs, p_w + d, cam, np.float64, False)[1] -
                          R.linearize_point(obs, p_w - d, cam, np.float64, False)[1]) / (2 * h)
        assert np.allclose(numl, Jl, rtol=1e-6, atol=1e-6 * np.abs(Jl).max())


# ---- edge cases -----------------------------------------------------------------------------------------------
def _quat_mul(a, b):  # (x, y, z, w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


@pytest.fixture(scope="module")
def camera_looking_away(small_problem):
    """Camera 3 turned by 180 degrees about its x axis: every point it observes is BEHIND it (z < 0), so with
    the projection-validity check all its observations are dropped (zero rows in the landmark blocks, a zero
    Jacobian column block for that camera: Jp_diag2 = 0, scaling 1 / eps, preconditioner block lambda I)."""
    from rootba_amd.problem import BalProblem
    p = small_problem
    cams = p.cams.copy()
    cams[3, :4] = _quat_mul(np.array([1.0, 0, 0, 0]), cams[3, :4])
    cams[3, 4:7] *= np.array([1.0, -1.0, -1.0])
    return BalProblem(cams, p.lms.copy(), p.lm_obs_offsets, p.obs_cam_idx, p.obs_xy, "camera-3-looks-away")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("optimized_cost", [0, 1, 2], ids=["ERROR", "ERROR_VALID", "ERROR_VALID_AVG"])
def test_invalid_projections(O, R, camera_looking_away, dt, optimized_cost):
    prob = camera_looking_away
    o, r = both(O, R, prob, dt, optimized_cost=optimized_cost)
    a, b = o.compute_error(), r.compute_error()
    assert (a.all_num_obs, a.valid_num_obs) == (b.all_num_obs, b.valid_num_obs) and b.valid_num_obs < b.all_num_obs
    t = tol(dt, 1e-12, 1e-5)
    assert abs(a.valid_error - b.valid_error) <= t * b.valid_error and abs(a.all_error - b.all_error) <= t * b.all_error
    rc_a, d_a, _ = o.stage1()
    rc_b, d_b, _ = r.stage1()
    assert rc_a == rc_b == 0 and rel_err(d_a, d_b) < t
    if optimized_cost:
        assert np.all(d_b[27:36] == 0) and np.all(d_a[27:36] == 0)
    o2, r2 = both(O, R, prob, dt, optimized_cost=optimized_cost)
    assert o2.linearize() == 0 and r2.linearize() == 0
    ia, ca = o2.solve(1e-2)
    ib, cb = r2.solve(1e-2)
    assert ca.num_iterations == cb.num_iterations and ca.termination_type == cb.termination_type
    assert rel_err(ia, ib) < tol(dt, 1e-10, 2e-4)
    la, lb = o2.apply(ib), r2.apply(ib)
    assert abs(la - lb) <= tol(dt, 1e-11, 2e-5) * abs(lb)
    assert rel_err(o2.get_state()[1], r2.get_state()[1]) < tol(dt, 1e-13, 1e-6)
    # a short LM run with the validity-aware costs
    o3, r3 = both(O, R, prob, np.float64, optimized_cost=optimized_cost, max_num_iterations=4)
    la_, ta = o3.optimize_lm()
    lb_, tb = r3.optimize_lm()
    assert ta == tb and len(la_) == len(lb_)
    for x, y in zip(la_, lb_):
        assert (x.step_is_successful, x.num_obs_valid) == (y.step_is_successful, y.num_obs_valid)
        assert abs(x.cost_valid - y.cost_valid) <= 1e-9 * y.cost_valid


@pytest.mark.parametrize("dt", DT)
def test_two_observation_landmarks_and_long_tracks(O, R, dt):
    """The smallest block the reference accepts (k = 2: 7 rows) next to long tracks (k up to 60: dynamic blocks
    with every padding width 0..3 of landmark_block_dynamic.hpp:62-66)."""
    from rootba_amd import problem as P
    k = np.concatenate([np.full(40, 2), np.arange(2, 61), [3, 5, 6, 7]])
    raw = P.synthetic_problem(70, k.size, int(k.sum()), seed=31, k=k)
    prob = P.preprocess(raw, seed=31, translation_sigma=0.3, point_sigma=0.3)
    kk = prob.obs_per_lm()
    assert kk.min() == 2 and kk.max() >= 50 and {(9 * int(v)) % 4 for v in kk} == {0, 1, 2, 3}
    o, r = both(O, R, prob, dt)
    t = tol(dt, 1e-12, 1e-5)
    rc_a, d_a, _ = o.stage1()
    rc_b, d_b, _ = r.stage1()
    assert rc_a == rc_b == 0 and rel_err(d_a, d_b) < t
    for l in (0, 39, 40, prob.n_lms - 5, prob.n_lms - 1):
        (A, li), (B, lj) = o.block(l), r.block(l)
        assert A.shape == B.shape and li == lj
    o2, r2 = both(O, R, prob, dt)
    assert o2.linearize() == 0 and r2.linearize() == 0
    ia, ca = o2.solve(1e-4)
    ib, cb = r2.solve(1e-4)
    assert abs(ca.num_iterations - cb.num_iterations) <= (0 if np.dtype(dt) == np.float64 else 1)
    if ca.num_iterations == cb.num_iterations:
        assert rel_err(ia, ib) < tol(dt, 1e-10, 5e-4)
    la, lb = o2.apply(ib), r2.apply(ib)
    assert abs(la - lb) <= tol(dt, 1e-11, 5e-5) * abs(lb)


def test_extreme_damping(O, R, small_problem):
    """lambda at both ends of the LM range (min_lambda = 1e-16 ... 1e8): the Givens damping sequence and its
    undo (landmark_block_base.ipp:165-210) with sqrt(lambda) far below / above the entries of R."""
    o, r = both(O, R, small_problem, np.float64)
    rc, d, _ = r.stage1()
    o.stage1()
    scaling = 1.0 / (1e-5 + np.sqrt(d))
    x = np.random.default_rng(2).standard_normal(9 * small_problem.n_cams)
    first = True
    for lam in (1e-16, 1e8, 1e-16, 1.0):
        o.set_pose_damping(lam)
        r.set_pose_damping(lam)
        b_a, s_a = o.stage2(lam, scaling if first else None)
        b_b, s_b = r.stage2(lam, scaling if first else None)
        first = False
        assert rel_err(b_a, b_b) < 1e-11 and rel_err(s_a, s_b) < 1e-11
        assert rel_err(o.right_multiply(x), r.right_multiply(x)) < 1e-11


def test_host_cpp_loader_pipeline_against_the_references(R, tmp_path):
    """The PRODUCT's C++ host loader (rootba_amd/csrc/host/bal_problem.hpp, through `bal_qr_hip --dry-run`) against
    load_normalized_bal_problem<double> of the reference (bal_problem.cpp:773-852) on the same file: load, normalise,
    PERTURB with the same seed (std::default_random_engine + a fresh normal_distribution per 3-vector: the same
    engine consumption as the reference), depth filter."""
    import json
    import subprocess
    from rootba_amd import build
    from rootba_amd import problem as P
    build.build()
    raw = P.synthetic_problem(16, 150, 600, seed=9)
    path = str(tmp_path / "problem-16-150-pre.txt")
    P.write_bal(raw, path)
    cases = [([], {}),
             (["--translation-sigma", "0.05", "--point-sigma", "0.02", "--rotation-sigma", "0.01", "--random-seed", "38401"],
              dict(translation_sigma=0.05, point_sigma=0.02, rotation_sigma=0.01, seed=38401)),
             (["--point-sigma", "0.5", "--random-seed", "7", "--init-depth-threshold", "60"],
              dict(point_sigma=0.5, seed=7, init_depth_threshold=60.0)),
             (["--no-normalize", "--rotation-sigma", "0.02", "--random-seed", "1"],
              dict(normalize=False, rotation_sigma=0.02, seed=1))]
    for args, kw in cases:
        out = subprocess.run([build.APP, "--input", path, "--dry-run", *args], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        info = json.loads(out.stdout.strip().splitlines()[-1])
        ref = R.load_bal(path, **kw)
        assert (info["num_cameras"], info["num_landmarks"], info["num_observations"]) == \
            (ref["cams"].shape[0], ref["lms"].shape[0], ref["obs_cam_idx"].size)
        assert np.allclose(info["landmark_sum"], ref["lms"].sum(0), rtol=1e-10, atol=1e-8)
        off = ref["lm_obs_offsets"]
        pos = np.arange(ref["obs_cam_idx"].size) - np.repeat(off[:-1], np.diff(off)) + 1
        chk = float(np.sum(pos * ((ref["obs_cam_idx"] + 1.0) * ref["obs_xy"][:, 0] + ref["obs_xy"][:, 1])))
        assert np.isclose(info["obs_checksum"], chk, rtol=1e-11)
        assert abs(info["rcs_sparsity"] - ref["rcs_sparsity"]) < 1e-12  # BalProblem::compute_rcs_sparsity
        q, qr = np.array(info["cam0"][:4]), ref["cams"][0, :4]
        assert min(np.linalg.norm(q - qr), np.linalg.norm(q + qr)) < 1e-10
        assert np.allclose(info["cam0"][4:], ref["cams"][0, 4:7], rtol=1e-10, atol=1e-9)


# ---- BASELINE.json configurations -------------------------------------------------------------------------------------
def test_baseline_config0_ladybug_float64_whole_run(O, R, ladybug_problem):
    """BASELINE configs[0]: "ladybug problem-49-7776 float64, solver=QR, CPU reference (plumbing, no GPU)" - the
    reference's own CPU-runnable case, run here by the reference build with the reference's default options
    (squared norm, 20 iterations, function_tolerance 1e-6) on the synthetic stand-in of that problem (the real file is
    not in the image): the restated oracle follows it iteration by iteration."""
    o = O.Oracle(ladybug_problem, np.float64, O.default_options())
    r = R.Reference(ladybug_problem, np.float64, R.default_options())
    la, ta = o.optimize_lm()
    lb, tb = r.optimize_lm()
    assert ta == tb == 1 and len(la) == len(lb) >= 5  # CONVERGED by the function tolerance
    for a, b in zip(la, lb):
        assert (a.iteration, a.step_is_successful, a.cg_iterations) == (b.iteration, b.step_is_successful, b.cg_iterations)
        assert abs(a.cost - b.cost) <= 1e-12 * b.cost
    (ca, la_), (cb, lb_) = o.get_state(), r.get_state()
    assert rel_err(ca, cb) < 1e-12 and rel_err(la_, lb_) < 1e-12


@pytest.mark.parametrize("dt", DT)
def test_baseline_config2_trafalgar_size_one_iteration(O, R, dt):
    """BASELINE configs[2] size (257 cameras, 65132 landmarks, ~226k observations): one LM iteration, stage by stage."""
    from rootba_amd import problem as P
    prob = P.preprocess(P.named_synthetic("trafalgar-257"))
    o, r = both(O, R, prob, dt)
    assert o.linearize() == 0 and r.linearize() == 0
    ia, ca = o.solve(1e-4)
    ib, cb = r.solve(1e-4)
    assert abs(ca.num_iterations - cb.num_iterations) <= (0 if np.dtype(dt) == np.float64 else 1)
    if ca.num_iterations == cb.num_iterations:
        assert rel_err(ia, ib) < tol(dt, 1e-10, 1e-3)
    la, lb = o.apply(ib), r.apply(ib)
    assert abs(la - lb) <= tol(dt, 1e-11, 1e-4) * abs(lb)
    ea, eb = o.compute_error(), r.compute_error()
    assert abs(ea.all_error - eb.all_error) <= tol(dt, 1e-12, 1e-4) * eb.all_error


@pytest.mark.parametrize("kw", [dict(max_cg_it=2), dict(min_cg_it=12, eta=0.5), dict(eta=1e-3), dict(eta=0.0, max_cg_it=30)],
                         ids=["max2", "min12", "eta1e-3", "eta0"])
def test_pcg_iteration_limits_and_forcing_sequence(O, R, small_problem, kw):
    """ConjugateGradientsSolver options as LinearizorBase::pcg fills them (linearizor_base.cpp:81-103): the iteration
    cap (termination NO_CONVERGENCE), the minimum iteration count, the forcing-sequence parameter eta."""
    o, r = both(O, R, small_problem, np.float64, **kw)
    assert o.linearize() == 0 and r.linearize() == 0
    for lam in (1e-4, 1e-6):
        ia, ca = o.solve(lam)
        ib, cb = r.solve(lam)
        assert (ca.num_iterations, ca.termination_type) == (cb.num_iterations, cb.termination_type)
        assert rel_err(ia, ib) < 1e-9
    if "max_cg_it" in kw:
        assert cb.num_iterations == kw["max_cg_it"] and cb.termination_type == 0
    if "min_cg_it" in kw:
        assert cb.num_iterations >= kw["min_cg_it"]


def test_lm_loop_non_default_trust_region_parameters(O, R, small_problem):
    """min_relative_decrease, initial_vee, vee_factor, trust-region bounds and a loose function tolerance
    (bal_bundle_adjustment.cpp:264-272, 434-519)."""
    from rootba_amd import problem as P
    far = P.preprocess(P.synthetic_problem(20, 150, 600, seed=11), seed=11, translation_sigma=3.0, point_sigma=3.0,
                       rotation_sigma=0.3)
    kw = dict(max_num_iterations=15, min_relative_decrease=0.3, initial_vee=3.0, vee_factor=4.0,
              initial_trust_region_radius=1e9, max_trust_region_radius=1e10, function_tolerance=1e-4)
    o, r = both(O, R, far, np.float64, **kw)
    la, ta = o.optimize_lm()
    lb, tb = r.optimize_lm()
    assert ta == tb and len(la) == len(lb)
    assert any(not b.step_is_successful for b in lb[1:])
    for i, (a, b) in enumerate(zip(la, lb)):
        assert (a.iteration, a.step_is_successful, a.step_is_valid) == (b.iteration, b.step_is_successful, b.step_is_valid)
        assert abs(a.cost - b.cost) <= 1e-7 * b.cost
        if i + 1 < len(la) and i > 0:
            assert abs(la[i + 1].lambda_ - b.lambda_) <= 1e-6 * b.lambda_  # damping of the next solve


def _write_bundler(prob, path, dead_cameras=(2, 9), seed=0):
    """Bundler v0.3 text from a problem in this repository's convention (z forward, image y down): cameras
    as f k1 k2 / R row-major / t in the Bundler convention (y, z axes inverted), uninitialised cameras (f = 0)
    inserted at `dead_cameras` with views that reference them, views of a point in random order."""
    from rootba_amd import problem as P
    rng = np.random.default_rng(seed)
    flip = np.diag([1.0, -1.0, -1.0])
    n_file = prob.n_cams + len(dead_cameras)
    file_idx = [i for i in range(n_file) if i not in dead_cameras]  # file index of our camera c
    lines = ["# Bundle file v0.3", f"{n_file} {prob.n_lms}"]
    it = iter(range(prob.n_cams))
    for i in range(n_file):
        if i in dead_cameras:
            lines += ["0 0 0", "0 0 0", "0 0 0", "0 0 0", "0 0 0"]
            continue
        c = next(it)
        R = flip @ P.quat_to_rot(prob.cams[c, :4])
        t = flip @ prob.cams[c, 4:7]
        lines.append("%.17g %.17g %.17g" % tuple(prob.cams[c, 7:10]))
        lines += ["%.17g %.17g %.17g" % tuple(row) for row in R]
        lines.append("%.17g %.17g %.17g" % tuple(t))
    off = prob.lm_obs_offsets
    for l in range(prob.n_lms):
        lines.append("%.17g %.17g %.17g" % tuple(prob.lms[l]))
        lines.append("%d %d %d" % tuple(rng.integers(0, 256, 3)))
        views = [(file_idx[prob.obs_cam_idx[o]], prob.obs_xy[o, 0], -prob.obs_xy[o, 1]) for o in range(off[l], off[l + 1])]
        if l % 7 == 0:
            views.append((dead_cameras[l % len(dead_cameras)], 1.5, -2.5))  # a view of an uninitialised camera
        order = rng.permutation(len(views))
        lines.append(" ".join([str(len(views))] + ["%d %d %.17g %.17g" % (views[j][0], int(rng.integers(0, 9999)), views[j][1], views[j][2])
                                                    for j in order]))
    open(path, "w").write("\n".join(lines) + "\n")


def test_bundler_format_host_loader_against_the_references(R, tmp_path):
    """BalProblem::load_bundler (bal_problem.cpp:284-404): uninitialised cameras dropped and the rest renumbered, their
    views skipped, views bucketed in camera order, axis and image-y inversion - the reference's own loader, the
    product's C++ loader (`--input-type BUNDLER` and the file-name autodetection) and the problem the file was
    written from."""
    import json
    import subprocess
    from rootba_amd import build
    from rootba_amd import problem as P
    build.build()
    raw = P.synthetic_problem(16, 150, 600, seed=9)
    path = str(tmp_path / "bundle.out")
    _write_bundler(raw, path)
    ref = R.load_bal(path, normalize=False, input_type="AUTO")  # autodetected from the name
    assert ref["cams"].shape[0] == raw.n_cams and ref["lms"].shape[0] == raw.n_lms
    assert np.array_equal(ref["lm_obs_offsets"], raw.lm_obs_offsets) and np.array_equal(ref["obs_cam_idx"], raw.obs_cam_idx)
    assert np.allclose(ref["obs_xy"], raw.obs_xy, rtol=1e-15, atol=0) and np.allclose(ref["lms"], raw.lms, rtol=1e-15, atol=0)
    assert rel_err(ref["cams"][:, 4:], raw.cams[:, 4:]) < 1e-15
    for qa, qb in zip(ref["cams"][:, :4], raw.cams[:, :4]):
        assert min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) < 1e-13
    for args, kw in (([], dict(input_type="BUNDLER")),
                     (["--input-type", "BUNDLER", "--no-normalize"], dict(input_type="BUNDLER", normalize=False)),
                     (["--init-depth-threshold", "60", "--point-sigma", "0.1", "--random-seed", "3"],
                      dict(input_type="AUTO", init_depth_threshold=60.0, point_sigma=0.1, seed=3))):
        out = subprocess.run([build.APP, "--input", path, "--dry-run", *args], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        info = json.loads(out.stdout.strip().splitlines()[-1])
        want = R.load_bal(path, **kw)
        assert (info["num_cameras"], info["num_landmarks"], info["num_observations"]) == \
            (want["cams"].shape[0], want["lms"].shape[0], want["obs_cam_idx"].size)
        assert np.allclose(info["landmark_sum"], want["lms"].sum(0), rtol=1e-10, atol=1e-8)
        off = want["lm_obs_offsets"]
        pos = np.arange(want["obs_cam_idx"].size) - np.repeat(off[:-1], np.diff(off)) + 1
        chk = float(np.sum(pos * ((want["obs_cam_idx"] + 1.0) * want["obs_xy"][:, 0] + want["obs_xy"][:, 1])))
        assert np.isclose(info["obs_checksum"], chk, rtol=1e-11)
        q, qr = np.array(info["cam0"][:4]), want["cams"][0, :4]
        assert min(np.linalg.norm(q - qr), np.linalg.norm(q + qr)) < 1e-10
        assert np.allclose(info["cam0"][4:], want["cams"][0, 4:7], rtol=1e-10, atol=1e-9)
    # malformed: no comment line; a duplicate view of one camera
    txt = open(path).read()
    bad1 = str(tmp_path / "bundle_nocomment.out")
    open(bad1, "w").write(txt.split("\n", 1)[1])
    assert subprocess.run([build.APP, "--input", bad1, "--dry-run"], capture_output=True, text=True).returncode != 0


def test_host_cpp_solver_options_defaults_and_mapping(R):
    """The C++ host layer's SolverOptions (rootba_amd/csrc/host/linearizor_hip.hpp; what `bal_qr_hip` hands to
    rba_create, printed by --dump-options): defaults equal to the reference's own SolverOptions declaration, the
    validity flag derived from optimized_cost as LinearizorQR does (linearizor_qr.cpp:58-68), use_double = true."""
    import json
    import subprocess
    from rootba_amd import build
    build.build()

    def dump(*args):
        out = subprocess.run([build.APP, "--dump-options", *args], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        return json.loads(out.stdout.strip().splitlines()[-1])
    d, ref = dump(), R.default_options()
    for k, v in d.items():
        if k == "use_double":
            assert v == 1  # solver_options.hpp:257-259
        else:
            assert v == getattr(ref, k), k
    for cost, idx in (("ERROR", 0), ("ERROR_VALID", 1), ("ERROR_VALID_AVG", 2)):
        d = dump("--optimized-cost", cost)
        want = R.default_options(optimized_cost=idx)
        assert (d["optimized_cost"], d["use_valid_projections_only"]) == (idx, int(idx != 0))
        assert want.optimized_cost == idx
    d = dump("--preconditioner-type", "JACOBI", "--robust-norm", "HUBER", "--huber-parameter", "0.5", "--no-staged-execution",
             "--max-num-iterations", "7", "--eta", "0.01", "--max-linear-solver-iterations", "33", "--function-tolerance", "1e-9",
             "--jacobi-scaling-epsilon", "1.0", "--solver-type", "SCHUR_COMPLEMENT")
    assert (d["preconditioner_type"], d["robust_norm"], d["huber_parameter"], d["staged_execution"], d["max_num_iterations"],
            d["eta"], d["max_cg_it"], d["function_tolerance"], d["jacobi_scaling_eps"], d["solver_type"]) == \
        (0, 1, 0.5, 0, 7, 0.01, 33, 1e-9, 1.0, 1)
