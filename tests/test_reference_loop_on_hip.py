"""The reference-side binding as real code: integration/rootba/solver/linearizor_hip.hpp is a `Linearizor<Scalar>`
written against the REFERENCE'S headers that forwards to the C ABI of include/rootba_hip.h; oracle/build_ref.sh
compiles it with the reference's own sources (third-party stand-ins, see oracle/README.md) and wraps the reference's
factory so that `Linearizor<Scalar>::create` returns it (integration/linearizor_factory_hip.cpp - the one `case` a
maintainer adds). The reference's LM loop `optimize_lm_ours` (bal_bundle_adjustment.cpp:249-544), UNMODIFIED, then
drives the library through the drop-in boundary:

  * `-m gpu`:      reference LM loop -> LinearizorHIP binding -> rootba_amd/librootba_hip.so (the product), compared
                   with the same loop on the reference's own LinearizorQR;
  * `-m "not gpu"`: the same object code against oracle/_ref/librootba_hip_mock.so, a test double of the C ABI backed
                   by the CPU oracle - it checks the binding itself (topology / option / state conversion, the
                   backup-restore protocol of rejected steps) where there is no GPU.
One process can bind the C ABI to one provider only, so the two groups never run in the same pytest process.
"""
import numpy as np
import pytest

from conftest import rel_err

DT = [np.float64, np.float32]


def _mods(provider):
    from oracle import ref as R
    if not R.binding_available():
        pytest.skip("oracle/_ref/librootba_ref_binding.so is not present")
    try:
        R.binding_lib(provider)
    except RuntimeError as e:  # the other provider was bound earlier in this process
        pytest.skip(str(e))
    return R


def _pair(R, provider, prob, dt, **kw):
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    base["use_valid_projections_only"] = int(base.get("optimized_cost", 0) != 0)  # linearizor_qr.cpp:58-68
    return (R.ReferenceOnHip(prob, dt, R.default_options(**base), provider=provider),
            R.Reference(prob, dt, R.default_options(**base)))


def _far_problem():
    from rootba_amd import problem as P
    return P.preprocess(P.synthetic_problem(20, 150, 600, seed=11), seed=11, translation_sigma=3.0, point_sigma=3.0,
                        rotation_sigma=0.3)


def check_one_iteration(R, provider, prob, dt, **kw):
    """Linearizor::{compute_error, linearize, solve, apply} of the binding vs the reference's LinearizorQR."""
    f64 = np.dtype(dt) == np.float64
    h, r = _pair(R, provider, prob, dt, **kw)
    a, b = h.compute_error(), r.compute_error()
    assert (a.all_num_obs, a.valid_num_obs) == (b.all_num_obs, b.valid_num_obs)
    assert abs(a.all_error - b.all_error) <= (1e-12 if f64 else 1e-6) * b.all_error
    assert h.linearize() == 0 and r.linearize() == 0
    for lam in (1e-4, 1e-2):  # the second solve = a backtracking step on the same linearisation
        ih, ch = h.solve(lam)
        ir, cr = r.solve(lam)
        assert abs(ch.num_iterations - cr.num_iterations) <= (0 if f64 else 1)
        if ch.num_iterations == cr.num_iterations:
            assert rel_err(ih, ir) < (1e-9 if f64 else 2e-3)
    lh, lr = h.apply(ir), r.apply(ir)
    assert abs(lh - lr) <= (1e-10 if f64 else 1e-4) * abs(lr)
    (ca, la), (cb, lb) = h.get_state(), r.get_state()  # the host BalProblem of each side
    assert rel_err(ca, cb) < (1e-10 if f64 else 1e-4) and rel_err(la, lb) < (1e-10 if f64 else 1e-4)
    a, b = h.compute_error(), r.compute_error()
    assert abs(a.all_error - b.all_error) <= (1e-10 if f64 else 1e-4) * b.all_error


def check_lm_run(R, provider, prob, dt, rows_exact, loose=False, **kw):
    """bundle_adjust_manual -> optimize_lm_ours with the factory returning the binding, vs the same loop on
    LinearizorQR: iteration by iteration while the solves are well determined, then the same optimum."""
    f64 = np.dtype(dt) == np.float64
    h, r = _pair(R, provider, prob, dt, **kw)
    lh, th = h.optimize_lm()
    lr, tr = r.optimize_lm()
    for i, (a, b) in enumerate(zip(lh[:rows_exact], lr[:rows_exact])):
        assert (a.iteration, bool(a.step_is_successful), bool(a.step_is_valid)) == \
            (b.iteration, bool(b.step_is_successful), bool(b.step_is_valid)), i
        # loose: long, badly conditioned solves (tiny lambda) - two correct float64 implementations may stop a few
        # PCG iterations apart and the accepted costs then differ at the level of the truncation, not of rounding
        cg_slack = max(2, b.cg_iterations // 20) if loose else (0 if f64 else 1)
        assert abs(a.cg_iterations - b.cg_iterations) <= cg_slack, i
        assert abs(a.cost - b.cost) <= (1e-6 if loose else 1e-10 if f64 else 1e-4) * b.cost, i
        # the reference's own trust-region bookkeeping (continuous in the step quality below rho = 0.937)
        assert abs(a.lambda_ - b.lambda_) <= (1e-3 if (loose or not f64) else 1e-6) * b.lambda_, i
    fh = min(x.cost for x in lh if x.step_is_successful)
    fr = min(x.cost for x in lr if x.step_is_successful)
    assert abs(fh - fr) <= (1e-7 if f64 else 2e-5) * fr
    if f64:
        assert th == tr and len(lh) == len(lr)
        assert rel_err(h.get_state()[0], r.get_state()[0]) < 1e-5  # the optimised BalProblem on the host
    return lh, lr


# ---- no GPU: the binding against the oracle-backed test double of the C ABI --------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kw", [dict(), dict(preconditioner_type=0), dict(robust_norm=0), dict(optimized_cost=1)],
                         ids=["default", "jacobi", "squared", "valid-only"])
def test_binding_one_iteration_mock(small_problem, dt, kw):
    check_one_iteration(_mods("mock"), "mock", small_problem, dt, **kw)


@pytest.mark.parametrize("dt", DT)
def test_binding_lm_run_mock(small_problem, dt):
    check_lm_run(_mods("mock"), "mock", small_problem, dt, rows_exact=4, max_num_iterations=12)


def test_binding_lm_run_with_rejected_steps_mock():
    """Rejected steps: the driver calls bal_problem.restore() behind the linearizor's back; the binding keeps the host
    problem as the source of truth (state uploaded before every update)."""
    lh, lr = check_lm_run(_mods("mock"), "mock", _far_problem(), np.float64, rows_exact=6, max_num_iterations=10,
                          initial_trust_region_radius=1e12)
    assert any(not x.step_is_successful for x in lr[1:6])


def check_rejected_step_without_camera_change(R, provider, prob, dt):
    """The driver's protocol around a REJECTED step whose camera increment is exactly zero (what the library returns
    when its PCG ends with x = 0: only landmarks move): bal_problem.backup(), apply(0), compute_error,
    bal_problem.restore() - bal_bundle_adjustment.cpp:401-509. BalProblem's cameras equal the backup whether or not
    the driver restored, so the binding cannot infer the restore from them (VERDICT / ADVICE round 3: it kept the
    rejected landmarks on the device and wrote them over the restored host state at the end). Afterwards both sides
    must hold the state from before the step: the device (error through the binding) and the host (BalProblem after
    sync_host), bit for bit, and the next real step must be the one the reference's LinearizorQR takes from there."""
    f64 = np.dtype(dt) == np.float64
    h, r = _pair(R, provider, prob, dt)
    e0 = h.compute_error()
    assert h.linearize() == 0 and r.linearize() == 0
    c0, l0 = (a.copy() for a in h.get_state())
    inc, _ = h.solve(1e-4)
    zero = np.zeros_like(inc)
    h.backup()                      # bal_problem.backup()
    l_diff = h.apply(zero)          # landmarks move, no camera does
    e1 = h.compute_error()
    assert np.isfinite(l_diff) and e1.all_error != e0.all_error  # (the landmark-only step did change the cost)
    h.restore()                     # the driver rejects it: bal_problem.restore()
    e2 = h.compute_error()          # the binding's next call: the device must be back at the backup
    # (same state bit for bit - checked below -; the parallel sum of the evaluation itself has no fixed order)
    tol = 1e-12 if f64 else 1e-6
    assert abs(e2.all_error - e0.all_error) <= tol * e0.all_error and e2.all_num_obs == e0.all_num_obs
    assert abs(e1.all_error - e0.all_error) > 1e3 * tol * e0.all_error
    c2, l2 = h.get_state()          # sync_host + BalProblem
    assert np.array_equal(c2, c0) and np.array_equal(l2, l0)
    # ... and the accepted twin: without the restore the moved landmarks stay, on both sides
    h.solve(1e-4)  # (the back-substitution left the landmark blocks undamped, ipp:247-248: damp them again)
    h.backup()
    h.apply(zero)
    e3 = h.compute_error()
    assert abs(e3.all_error - e1.all_error) <= tol * e1.all_error
    c3, l3 = h.get_state()
    # (float32: the retraction re-normalises the quaternion, so a zero increment may move its last bit - either way
    #  the binding must have followed)
    assert (np.array_equal(c3, c0) if f64 else rel_err(c3, c0) < 1e-6) and not np.array_equal(l3, l0)
    h.restore()
    # the next real step from the restored state is the reference's
    ir, _ = r.solve(1e-4)
    h.solve(1e-4)  # (as the driver does before every apply: the landmark damping of the step)
    h.backup()
    r.backup()
    lh, lr = h.apply(ir), r.apply(ir)
    assert abs(lh - lr) <= (1e-10 if f64 else 1e-4) * abs(lr)
    (ca, la), (cb, lb) = h.get_state(), r.get_state()
    assert rel_err(ca, cb) < (1e-10 if f64 else 1e-5) and rel_err(la, lb) < (1e-10 if f64 else 1e-5)


@pytest.mark.parametrize("dt", DT)
def test_binding_rejected_step_without_camera_change_mock(small_problem, dt):
    check_rejected_step_without_camera_change(_mods("mock"), "mock", small_problem, dt)


def test_reference_lm_loop_rejects_a_landmark_only_step_mock(small_problem, monkeypatch):
    """The same through the reference's UNMODIFIED optimize_lm_ours: the test double returns a zero camera increment
    from its third solve and a negative model cost change from the apply that follows (mock hook
    RBA_MOCK_ZERO_INC_SOLVE), the loop rejects the step and restores BalProblem, raises lambda and solves again from
    the same linearisation. The cost of that NEXT step must be the cost of taking it from the state before the
    rejected one - recomputed here by replaying the two accepted iterations and taking that step by hand."""
    R = _mods("mock")
    kw = dict(max_num_iterations=4, function_tolerance=0.0)
    monkeypatch.setenv("RBA_MOCK_ZERO_INC_SOLVE", "3")
    h, _ = _pair(R, "mock", small_problem, np.float64, **kw)
    rows, _ = h.optimize_lm()
    monkeypatch.delenv("RBA_MOCK_ZERO_INC_SOLVE")
    assert [bool(x.step_is_successful) for x in rows[:5]] == [True, True, True, False, True]
    final_c, final_l = h.get_state()
    # replay: two accepted iterations, then the step of iteration 4 (lambda of row 3 = the raised damping) by hand
    g, _ = _pair(R, "mock", small_problem, np.float64, **dict(kw, max_num_iterations=2))
    rows2, _ = g.optimize_lm()
    assert all(abs(a.cost - b.cost) <= 1e-12 * b.cost for a, b in zip(rows2, rows[:3]))
    g.compute_error()
    assert g.linearize() == 0
    inc, _ = g.solve(rows[3].lambda_)
    g.backup()
    g.apply(inc)
    assert abs(g.compute_error().all_error - rows[4].cost) <= 1e-11 * rows[4].cost
    c, l = g.get_state()
    assert rel_err(c, final_c) < 1e-10 and rel_err(l, final_l) < 1e-10  # (parallel sums inside the solves: no fixed order)


def test_factory_returns_the_reference_linearizors_otherwise(small_problem):
    """Without ROOTBA_LINEARIZOR=hip, and for the other solver types, the wrapped factory is the reference's."""
    R = _mods("mock")
    lib = R.binding_lib("mock")
    for kw in (dict(), dict(solver_type=1), dict(solver_type=1, preconditioner_type=2)):
        a = R.Reference(small_problem, np.float64, R.default_options(robust_norm=1, **kw), library=lib)
        b = R.Reference(small_problem, np.float64, R.default_options(robust_norm=1, **kw))
        assert a.linearize() == 0 and b.linearize() == 0
        ia, ca = a.solve(1e-4)
        ib, cb = b.solve(1e-4)
        assert ca.num_iterations == cb.num_iterations and np.array_equal(ia, ib)


# ---- GPU: the reference's LM loop on the HIP library ---------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("kw", [dict(), dict(preconditioner_type=0)], ids=["default", "jacobi"])
def test_binding_one_iteration_hip(small_problem, dt, kw):
    import torch  # noqa: F401  (HIP runtime first, as in bench.py)
    check_one_iteration(_mods("hip"), "hip", small_problem, dt, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DT)
def test_reference_lm_loop_drives_the_hip_library(small_problem, dt):
    import torch  # noqa: F401
    check_lm_run(_mods("hip"), "hip", small_problem, dt, rows_exact=4, max_num_iterations=12)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DT)
def test_binding_rejected_step_without_camera_change_hip(small_problem, dt):
    import torch  # noqa: F401
    check_rejected_step_without_camera_change(_mods("hip"), "hip", small_problem, dt)


@pytest.mark.gpu
def test_reference_lm_loop_with_rejected_steps_on_the_hip_library():
    import torch  # noqa: F401
    lh, lr = check_lm_run(_mods("hip"), "hip", _far_problem(), np.float64, rows_exact=6, loose=True, max_num_iterations=10,
                          initial_trust_region_radius=1e12)
    assert any(not x.step_is_successful for x in lr[1:6])


def test_binding_on_the_real_library_fails_loudly_without_a_gpu(small_problem, tmp_path):
    """The binding resolves every entry point it calls in the REAL rootba_amd/librootba_hip.so, and without a GPU
    the run dies in rba_create with the library's error (the reference CHECK-aborts; no CPU fallback anywhere)."""
    import os
    import subprocess
    import sys
    from rootba_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("librootba_hip.so is not built")
    if L.device_count() > 0:
        pytest.skip("only meaningful on a box without a GPU")
    _mods("mock")  # (skips when oracle/_ref is absent)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from oracle import ref as R\nfrom rootba_amd import problem as P\n"
            "prob = P.preprocess(P.synthetic_problem(8, 40, 160, seed=1), seed=1)\n"
            "h = R.ReferenceOnHip(prob, np.float64, R.default_options(), provider='hip')\n"
            "h.linearize()\nprint('SURVIVED')\n") % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode != 0 and "SURVIVED" not in out.stdout
    assert "rba_create" in out.stderr
