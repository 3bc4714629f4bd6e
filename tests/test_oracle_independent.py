"""Independent second check of the oracle that shares NO code with it.

The oracle (oracle/rootba_oracle.hpp) restates the reference's analytic Jacobians,
Householder / Givens landmark blocks, PCG and back-substitution line by line. Here the
same quantities are derived a different way, from the definition of the problem only:

* residuals written in torch (float64) straight from the BAL camera model and the
  reference's retraction  T <- se3_expd(inc) T  (bal_problem.hpp:99-109), with
  ``torch.matrix_exp`` for the rotation update;
* the dense Jacobian by ``torch.autograd`` (no analytic derivative anywhere);
* the reduced camera system, its solution, the landmark update and the model cost change
  from DENSE normal equations with ``numpy.linalg`` (projector  I - Jl (Jl^T Jl + lambda I)^-1 Jl^T),
  i.e. the textbook Schur complement instead of the reference's QR marginalisation.

A misreading of the reference that the oracle and the HIP kernels could share (sign of the
pose Jacobian, retraction order, Huber weight, scaling epsilons, damping placement, what
`l_diff` sums over) would show up here. What this cannot pin: Sophus' epsilonSqrt value
(an input constant) — it is passed in explicitly below.
Tolerances: 1e-9 relative on linear-algebra invariants in float64 (the oracle's own
tests use 1e-10..1e-12 between its two formulations), 1e-7 on the PCG solution (solved
to eta = 1e-14).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import oracle as O

EPS_SQRT_F64 = 1e-5  # Sophus::Constants<double>::epsilonSqrt (assumed)
HUBER = 1.0
LAMBDA = 1e-2


def _hat(w):
    z = torch.zeros((), dtype=w.dtype)
    return torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]),
                        torch.stack([-w[1], w[0], z])])


def _quat_to_rot(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(q.shape[:-1] + (3, 3))


def _residuals(inc, lms, cams, obs_cam, obs_lm, obs_xy):
    """2 n_obs residuals at camera increment `inc` (n_c x 9: upsilon, omega, df, dk1, dk2)."""
    n_c = cams.shape[0]
    R0 = _quat_to_rot(cams[:, :4])
    t0 = cams[:, 4:7]
    E = torch.stack([torch.matrix_exp(_hat(inc[c, 3:6])) for c in range(n_c)])
    R = E @ R0                                             # exp(omega) R
    t = (E @ t0[..., None])[..., 0] + inc[:, 0:3]          # exp(omega) t + upsilon
    intr = cams[:, 7:10] + inc[:, 6:9]
    pc = (R[obs_cam] @ lms[obs_lm][..., None])[..., 0] + t[obs_cam]
    m = pc[:, :2] / pc[:, 2:3]
    r2 = (m * m).sum(-1)
    f, k1, k2 = intr[obs_cam, 0], intr[obs_cam, 1], intr[obs_cam, 2]
    proj = (f * (1 + k1 * r2 + k2 * r2 * r2))[:, None] * m
    return (proj - obs_xy).reshape(-1), pc[:, 2]


class DenseModel:
    """Everything the hot path computes, from dense float64 linear algebra."""

    def __init__(self, prob):
        self.prob = prob
        self.n_c, self.n_l, self.n_o = prob.n_cams, prob.n_lms, prob.n_obs
        k = np.diff(prob.lm_obs_offsets)
        self.obs_lm = np.repeat(np.arange(self.n_l), k)
        self.obs_cam = np.asarray(prob.obs_cam_idx, dtype=np.int64)
        tc = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
        self.cams, self.lms, self.obs_xy = tc(prob.cams), tc(prob.lms), tc(prob.obs_xy)
        self.oc, self.ol = torch.tensor(self.obs_cam), torch.tensor(self.obs_lm)

    def residuals(self, inc=None, lms=None):
        inc = torch.zeros(self.n_c, 9, dtype=torch.float64) if inc is None else inc
        lms = self.lms if lms is None else lms
        return _residuals(inc, lms, self.cams, self.oc, self.ol, self.obs_xy)

    def cost(self, lms=None, cams=None):
        if cams is not None:
            saved, self.cams = self.cams, cams
        r, z = self.residuals(lms=lms)
        if cams is not None:
            self.cams = saved
        r2 = (r.reshape(-1, 2) ** 2).sum(-1).numpy()
        w = np.where(r2 < HUBER ** 2, 1.0, HUBER / np.sqrt(np.maximum(r2, 1e-300)))
        return float((0.5 * (2 - w) * w * r2).sum()), float(np.sqrt(r2).sum()), int((z.numpy() >= EPS_SQRT_F64).sum())

    def linearize(self):
        inc0 = torch.zeros(self.n_c, 9, dtype=torch.float64)
        f = lambda i, l: self.residuals(i, l)[0]  # noqa: E731
        Ji, Jl = torch.autograd.functional.jacobian(f, (inc0, self.lms), vectorize=True)
        r = f(inc0, self.lms).numpy()
        Jp = Ji.reshape(2 * self.n_o, 9 * self.n_c).numpy()
        Jl = Jl.reshape(2 * self.n_o, 3 * self.n_l).numpy()
        r2 = (r.reshape(-1, 2) ** 2).sum(-1)
        w = np.where(r2 < HUBER ** 2, 1.0, HUBER / np.sqrt(np.maximum(r2, 1e-300)))
        sw = np.repeat(np.sqrt(w), 2)
        self.Jp, self.Jl, self.r = Jp * sw[:, None], Jl * sw[:, None], r * sw
        self.jp_diag2 = (self.Jp ** 2).sum(0)
        self.Dp = 1.0 / (EPS_SQRT_F64 + np.sqrt(self.jp_diag2))
        self.Dl = 1.0 / (EPS_SQRT_F64 + np.sqrt((self.Jl ** 2).sum(0)))
        self.Jps, self.Jls = self.Jp * self.Dp, self.Jl * self.Dl

    def reduced_system(self, lam):
        Hll = self.Jls.T @ self.Jls + lam * np.eye(3 * self.n_l)   # block diagonal by construction
        self.Hll_inv = np.linalg.inv(Hll)
        P = np.eye(2 * self.n_o) - self.Jls @ self.Hll_inv @ self.Jls.T
        S = self.Jps.T @ P @ self.Jps + lam * np.eye(9 * self.n_c)
        b = self.Jps.T @ (P @ self.r)
        return S, b

    def back_substitute(self, inc):
        delta = -self.Hll_inv @ (self.Jls.T @ (self.r + self.Jps @ inc))
        g = self.Jps @ inc + self.Jls @ delta
        l_diff = -(0.5 * g @ g + g @ self.r)
        return delta, l_diff


@pytest.fixture(scope="module")
def tiny_problem():
    from rootba_amd import problem as P
    raw = P.synthetic_problem(8, 60, 260, seed=11)
    return P.preprocess(raw, seed=11, translation_sigma=0.3, point_sigma=0.3)


def _opts(**kw):
    return O.default_options(robust_norm=1, huber_parameter=HUBER, **kw)


def test_cost_matches_the_definition(tiny_problem):
    m = DenseModel(tiny_problem)
    ri = O.Oracle(tiny_problem, np.float64, _opts()).compute_error()
    err, rsum, nvalid = m.cost()
    assert ri.all_num_obs == tiny_problem.n_obs and ri.valid_num_obs == nvalid
    assert abs(ri.all_error - err) < 1e-12 * err
    assert abs(ri.all_residual_sum - rsum) < 1e-12 * rsum


def test_linearisation_against_autograd_and_dense_normal_equations(tiny_problem):
    m = DenseModel(tiny_problem)
    m.linearize()
    o = O.Oracle(tiny_problem, np.float64, _opts(eta=1e-14, max_cg_it=5000))
    n = 9 * m.n_c
    # stage 1: Jp_diag2, Jl column scale, pose scaling
    rc, d2, _ = o.stage1()
    assert rc == 0
    assert rel_err(d2, m.jp_diag2) < 1e-10
    assert rel_err(o.jl_col_scale().ravel(), m.Dl) < 1e-10
    assert o.linearize() == 0
    assert rel_err(o.pose_scaling(), m.Dp) < 1e-10

    # stage 2 + PCG to convergence: b, SCHUR_JACOBI blocks, H x, the increment
    S, b = m.reduced_system(LAMBDA)
    inc_o, cg = o.solve(LAMBDA)
    assert cg.termination_type == 1
    assert rel_err(o.last_b(), b) < 1e-9
    blocks = o.precond_blocks()
    for c in range(m.n_c):
        assert rel_err(blocks[c], S[9 * c:9 * c + 9, 9 * c:9 * c + 9]) < 1e-9
    x = np.random.default_rng(3).uniform(-1, 1, n)
    assert rel_err(o.right_multiply(x), S @ x) < 1e-9
    inc_dense = -np.linalg.solve(S, b)   # H (-x) = b, result negated (linearizor_base.cpp:100)
    assert rel_err(inc_o, inc_dense) < 1e-7

    # back-substitution: l_diff, landmark update, camera retraction
    delta, l_diff = m.back_substitute(inc_dense)
    l_o = o.apply(inc_dense)
    assert abs(l_o - l_diff) < 1e-9 * abs(l_diff)
    cams_o, lms_o = o.get_state()
    lms_new = np.asarray(tiny_problem.lms, np.float64) + (delta * m.Dl).reshape(-1, 3)
    assert rel_err(lms_o, lms_new) < 1e-11
    # cameras: the model's own retraction at the un-scaled increment
    inc_un = torch.tensor((inc_dense * m.Dp).reshape(-1, 9))
    R0 = _quat_to_rot(m.cams[:, :4])
    E = torch.stack([torch.matrix_exp(_hat(inc_un[c, 3:6])) for c in range(m.n_c)])
    R_new = (E @ R0).numpy()
    t_new = ((E @ m.cams[:, 4:7][..., None])[..., 0] + inc_un[:, :3]).numpy()
    R_o = _quat_to_rot(torch.tensor(np.asarray(cams_o, np.float64)[:, :4])).numpy()
    assert rel_err(R_o, R_new) < 1e-11
    assert rel_err(cams_o[:, 4:7], t_new) < 1e-11
    assert rel_err(cams_o[:, 7:10], m.cams[:, 7:10].numpy() + inc_un[:, 6:9].numpy()) < 1e-12
    # ... and the cost at the new state, evaluated by the independent residual code
    err_new, _, _ = m.cost(lms=torch.tensor(lms_new),
                           cams=torch.tensor(np.concatenate([np.asarray(cams_o, np.float64)], 0)))
    ri = o.compute_error()
    assert abs(ri.all_error - err_new) < 1e-11 * err_new
    # the model predicted a decrease and the true cost went down
    assert l_diff > 0 and err_new < m.cost()[0]


def test_power_series_and_jacobi_blocks_from_dense_pieces(tiny_problem):
    """JACOBI blocks = diagonal blocks of Jp^T Jp + lambda I; the power-series preconditioner
    = sum_i (Hpp^-1 E0)^i Hpp^-1 with E0 = Jp^T Jl Hll^-1 Jl^T Jp (preconditioner.hpp:180-245)."""
    m = DenseModel(tiny_problem)
    m.linearize()
    n, order = 9 * m.n_c, 5
    m.reduced_system(LAMBDA)
    Hpp = np.zeros((n, n))
    G = m.Jps.T @ m.Jps
    for c in range(m.n_c):
        Hpp[9 * c:9 * c + 9, 9 * c:9 * c + 9] = G[9 * c:9 * c + 9, 9 * c:9 * c + 9]
    Hpp += LAMBDA * np.eye(n)
    E0 = m.Jps.T @ m.Jls @ m.Hll_inv @ m.Jls.T @ m.Jps
    Hinv = np.linalg.inv(Hpp)
    M = sum(np.linalg.matrix_power(Hinv @ E0, i) @ Hinv for i in range(order + 1))
    o = O.Oracle(tiny_problem, np.float64, _opts(preconditioner_type=2, power_order=order))
    assert o.linearize() == 0
    v = np.random.default_rng(5).normal(size=n)
    assert rel_err(o.power_precond(LAMBDA, v), M @ v) < 1e-9
    oj = O.Oracle(tiny_problem, np.float64, _opts(preconditioner_type=0))
    assert oj.linearize() == 0
    oj.solve(LAMBDA)
    # (the stored JACOBI blocks are D Jp^T Jp D; lambda enters as the separate diagonal of
    #  BlockDiagonalPreconditioner, linearizor_qr.cpp:227-232)
    bj = oj.precond_blocks()
    for c in range(m.n_c):
        assert rel_err(bj[c] + LAMBDA * np.eye(9), Hpp[9 * c:9 * c + 9, 9 * c:9 * c + 9]) < 1e-9


def test_lm_iteration_against_dense_gauss_newton(tiny_problem):
    """One full LM iteration of the oracle's driver: with the PCG solved to convergence the
    accepted step must be the damped Gauss-Newton step of the dense model, and the logged
    cost the independently evaluated one."""
    m = DenseModel(tiny_problem)
    m.linearize()
    o = O.Oracle(tiny_problem, np.float64, _opts(eta=1e-14, max_cg_it=5000, max_num_iterations=1))
    log, _ = o.optimize_lm()
    assert log[0].iteration == 0 and abs(log[0].cost - m.cost()[0]) < 1e-12 * log[0].cost
    lam0 = 1.0 / 1e4  # initial_trust_region_radius = 1e4 (solver_options.hpp)
    assert abs(log[1].lambda_ - lam0) < 1e-15
    S, b = m.reduced_system(lam0)
    inc = -np.linalg.solve(S, b)
    delta, l_diff = m.back_substitute(inc)
    assert abs(log[1].l_diff - l_diff) < 1e-7 * abs(l_diff)
    assert log[1].step_is_successful == 1
    cams_o, lms_o = o.get_state()
    lms_new = np.asarray(tiny_problem.lms, np.float64) + (delta * m.Dl).reshape(-1, 3)
    assert rel_err(lms_o, lms_new) < 1e-7
    err_new, _, _ = m.cost(lms=torch.tensor(np.asarray(lms_o, np.float64)),
                           cams=torch.tensor(np.asarray(cams_o, np.float64)))
    assert abs(log[1].cost - err_new) < 1e-11 * err_new
