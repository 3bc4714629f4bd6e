"""The oracle's matrix-free Schur-complement solver (solver_type 2, oracle/rootba_oracle.hpp: sc_mf_prepare) - the float64
referee of lock-steps at sizes where the dense landmark blocks (square-root solver) and the dense H_pp (SC solver) do not
fit the host (final-13682: 55 / 121 GB against 5.6 GB) - against those two where they fit: same system, same PCG
recurrence, so in float64 the iterates of every index agree to rounding ("checked against its
dense path at trafalgar size")."""
import numpy as np
import pytest

from conftest import rel_err


def _solve(prob, solver_type, lam, n_it, **kw):
    from oracle import oracle as O
    o = O.Oracle(prob, np.float64, O.default_options(robust_norm=1, huber_parameter=1.0, solver_type=solver_type,
                                                    max_cg_it=n_it, eta=0.0, **kw))
    assert o.linearize() == 0
    inc, cg = o.solve(lam)
    assert cg.num_iterations == n_it
    return np.asarray(inc), o


@pytest.mark.parametrize("lam", [1e-2, 1e-5])
def test_matrix_free_sc_is_the_dense_sc_and_the_square_root_solver(small_problem, lam):
    ref, _ = _solve(small_problem, 0, lam, 25)
    dense, _ = _solve(small_problem, 1, lam, 25)
    mf, _ = _solve(small_problem, 2, lam, 25)
    assert rel_err(mf, dense) < 1e-10 and rel_err(mf, ref) < 1e-10


def test_matrix_free_sc_referee_at_trafalgar_size():
    """trafalgar-257 (bench.py's workload), the float scaling epsilon the referee is used with, 60 iterations of a
    solve that needs ~270 in float32: the square-root solver's iterate to 1e-9; the back-substitution gives the same
    landmark update and model cost change."""
    import types

    import bench
    from lockstep import EPS_SQRT_FLOAT
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    prob = bench.make_problem("trafalgar-257", args)[0]
    kw = dict(jacobi_scaling_eps=EPS_SQRT_FLOAT)
    ref, o0 = _solve(prob, 0, 1e-6, 60, **kw)
    mf, o2 = _solve(prob, 2, 1e-6, 60, **kw)
    assert rel_err(mf, ref) < 1e-9, rel_err(mf, ref)
    l0, l2 = o0.apply(ref), o2.apply(mf)
    assert abs(l0 - l2) <= 1e-8 * abs(l0)
    assert rel_err(o0.get_state()[1], o2.get_state()[1]) < 1e-10


def test_matrix_free_sc_rejects_other_preconditioners(small_problem):
    from oracle import oracle as O
    o = O.Oracle(small_problem, np.float64, O.default_options(solver_type=2, preconditioner_type=0))
    assert o.linearize() == 0
    with pytest.raises(ValueError):
        o.solve(1e-4)
