"""RBA_MIXED (BASELINE config 5's "mixed f32/f64"): double state / observations / costs, float
linear algebra. The reference has no mixed mode (it is templated on one Scalar), so parity is
anchored on both oracle precisions: the costs are the float64 oracle's costs of the same state
(1e-12), one iteration's linear algebra is the float32 oracle's (float tolerances of
tests/test_gpu_parity.py), and a whole LM run lands on the float64 oracle's optimum more closely
than the float32 run can (whose cost has ~1e-6 relative resolution)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _opts(mod, **kw):
    base = dict(robust_norm=1, huber_parameter=1.0)
    base.update(kw)
    return mod.default_options(**base)


@pytest.fixture(scope="module")
def ladybug_far():
    from rootba_amd import problem as P
    return P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)


def _mixed(prob, **kw):
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    return LinearizorHIP(prob, "mixed", _opts(L, **kw))


def test_mixed_cost_is_the_float64_cost(small_problem):
    from oracle import oracle as O
    g = _mixed(small_problem)
    o64 = O.Oracle(small_problem, np.float64, _opts(O))
    a, b = g.compute_error(), o64.compute_error()
    assert (a.all_num_obs, a.valid_num_obs) == (b.all_num_obs, b.valid_num_obs)
    assert abs(a.all_error - b.all_error) <= 1e-12 * b.all_error
    assert abs(a.valid_error - b.valid_error) <= 1e-12 * b.valid_error
    # the state round-trips in double, bit for bit
    c, l = g.get_state()
    assert c.dtype == np.float64 and np.array_equal(c, small_problem.cams) and np.array_equal(l, small_problem.lms)


def test_mixed_iteration_is_float_algebra_on_the_double_state(small_problem):
    """One iteration from the same point: the increment is the float32 oracle's (float tolerance); the
    double state after `apply` is old state + increment to double accuracy, i.e. NOT re-rounded to float;
    the new cost is the float64 cost of that state."""
    from oracle import oracle as O
    prob = small_problem
    g = _mixed(prob)
    o32 = O.Oracle(prob, np.float32, _opts(O))
    assert g.linearize() == 0 and o32.linearize() == 0
    ig, cg = g.solve(1e-4)
    io, co = o32.solve(1e-4)
    assert ig.dtype == np.float32 and cg.termination_type == 1
    assert abs(cg.num_iterations - co.num_iterations) <= 1
    ref = io
    if cg.num_iterations != co.num_iterations:  # the float iterate after exactly the product's count (eta = 0: no stopping test)
        o_n = O.Oracle(prob, np.float32, _opts(O, max_cg_it=cg.num_iterations, eta=0.0))
        assert o_n.linearize() == 0
        ref, cn = o_n.solve(1e-4)
        assert cn.num_iterations == cg.num_iterations
    assert rel_err(ig, ref) < 2e-3
    c0, l0 = g.get_state()
    lg, lo = g.apply(io), o32.apply(io)
    assert abs(lg - lo) <= 1e-4 * abs(lo)
    c1, l1 = g.get_state()
    co1, lo1 = o32.get_state()
    assert rel_err(c1, co1) < 1e-6 and rel_err(l1, lo1) < 1e-6
    # the landmark update is exact in double: new - old is a float number (the scaled increment) ...
    d = l1 - l0
    assert np.array_equal(d.astype(np.float32).astype(np.float64), d) or \
        np.max(np.abs(d.astype(np.float32).astype(np.float64) - d)) <= 1e-12 * np.max(np.abs(l1))
    # ... and the masters are not float roundings of themselves (they carry more than 24 bits)
    assert not np.array_equal(l1.astype(np.float32).astype(np.float64), l1)
    o64 = O.Oracle(prob, np.float64, _opts(O))
    o64.set_state(c1, l1)
    a, b = g.compute_error(), o64.compute_error()
    assert abs(a.all_error - b.all_error) <= 1e-12 * b.all_error
    # backup / restore cover the double state
    g.backup()
    g.apply(io)
    g.restore()
    c2, l2 = g.get_state()
    assert np.array_equal(c2, c1) and np.array_equal(l2, l1)


def test_mixed_lm_run_reaches_the_float64_optimum(ladybug_far):
    """Fixed 14 iterations, stopping rule off: the mixed run follows the float32 decisions while the steps
    are large and ends at the float64 optimum within 2e-7 relative - the float32 run cannot resolve its own
    cost better than ~1e-6 (tests/test_gpu_parity.py::test_lm_trajectory_matches_oracle)."""
    import torch  # noqa: F401
    from oracle import oracle as O
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    kw = dict(max_num_iterations=14, function_tolerance=0.0)
    gm = _mixed(ladybug_far, **kw)
    lm, _ = gm.optimize_lm()
    g32 = LinearizorHIP(ladybug_far, np.float32, _opts(L, **kw))
    l32, _ = g32.optimize_lm()
    o64 = O.Oracle(ladybug_far, np.float64, _opts(O, **kw))
    l64, _ = o64.optimize_lm()
    for a, b in zip(lm[:5], l64[:5]):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-5 * b.cost
    f64 = min(r.cost for r in l64 if r.step_is_successful)
    fm = min(r.cost for r in lm if r.step_is_successful)
    f32 = min(r.cost for r in l32 if r.step_is_successful)
    assert abs(fm - f64) / f64 < 2e-7, (fm, f64, f32)
    # the reported cost of the mixed run IS the float64 cost of its final state
    c, l = gm.get_state()
    o64.set_state(c, l)
    assert abs(o64.compute_error().all_error - lm[-1].cost) <= 1e-11 * lm[-1].cost or not lm[-1].step_is_successful
    # every accepted step decreased the double cost
    acc = [r.cost for r in lm if r.step_is_successful]
    assert all(b <= a for a, b in zip(acc, acc[1:]))


def test_mixed_is_square_root_only(small_problem):
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    with pytest.raises(RuntimeError, match="RBA_MIXED"):
        LinearizorHIP(small_problem, "mixed", _opts(L, solver_type=1))
