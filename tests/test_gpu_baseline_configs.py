"""GPU-vs-oracle comparisons at BASELINE.json's own single-GPU configurations (synthetic stand-ins of
the same size; the real BAL files when present):

  config 2  ladybug-49     f32  per-landmark QR invariants vs the CPU oracle
  config 3  trafalgar-257  f32  full LM run (QR + PCG pipeline) in lock-step with the oracle
  config 4  venice-1778    f32  first four LM iterations in lock-step (cost, CG count, |inc|)
  (config 1, ladybug-49 f64 "plumbing", is tests/test_gpu_parity.py::test_lm_trajectory_matches_oracle)

Tolerances: 1e-6 relative on the final cost (north_star). Per-iteration increment VECTORS are compared in lock-step at
identical states (tests/lockstep.py) against the oracle's iterate of the same precision and iteration index, and - for
float32 - both against the float64 oracle's iterate of that index: what float32 resolves here is ~1e-4 ... 1e-3 of the
increment for EITHER implementation (SURVEY.md §8c's 1e-4 is met only by the first, 2-3 iteration solves), so the
assertion that carries weight is the accuracy parity `gpu_vs_f64 <= c * oracle32_vs_f64`. Measured values
(MI355X, round 3) are quoted at each assertion. The oracle runs on the GPU box's host cores (trafalgar ~1.4 LM it/s,
venice ~10 s for three iterations).
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _opts(mod, **kw):
    return mod.default_options(robust_norm=1, huber_parameter=1.0, **kw)


def _bench_problem(name):
    """Exactly bench.py's workload (CVPR'21 common settings)."""
    import types

    import bench
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    return bench.make_problem(name, args)[0]


def _lockstep(name, dts, n_it, precond=1):
    from lockstep import lockstep_rows
    rows = list(lockstep_rows(_bench_problem(name), dts, n_it, precond))
    assert all(rows[0]["ok"]), rows[0]
    return rows[1:]


def _pair(prob, dtype, **kw):
    import torch  # noqa: F401
    from oracle import oracle as O
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    return LinearizorHIP(prob, dtype, _opts(L, **kw)), O.Oracle(prob, dtype, _opts(O, **kw))


def test_config2_ladybug49_f32_landmark_qr_vs_cpu():
    """BASELINE config 2: the per-landmark QR kernel alone, float32, against the CPU oracle."""
    prob = _bench_problem("ladybug-49")
    g, o = _pair(prob, np.float32)
    rc, d2, _ = o.stage1()
    st, d2g = g.linearize(want_jp_diag2=True)
    assert rc == 0 and st == 0
    assert rel_err(d2g, d2) < 1e-5
    assert rel_err(g.jl_col_scale(), o.jl_col_scale()) < 1e-5
    # R^T R and |Q1^T r| per landmark are invariant to the reflector signs
    Rg, qg = g.landmark_R(damped=False)
    k = prob.obs_per_lm()
    from oracle import oracle as O
    o64 = O.Oracle(prob, np.float64, _opts(O))
    assert o64.stage1()[0] == 0
    eg = eo = 0.0
    for l in np.random.default_rng(0).choice(prob.n_lms, 200, replace=False):
        blk, li = o.block(int(l))
        Ro = np.triu(blk[:3, li:li + 3].astype(np.float64))
        R = np.zeros((3, 3))
        R[np.triu_indices(3)] = Rg[l]
        assert rel_err(R.T @ R, Ro.T @ Ro) < 1e-4
        assert blk.shape[0] == 2 * k[l] + 3
        # |Q1^T r|: the residuals are ~0.5 px differences of ~1e3 px projections near the optimum, so
        # float32 itself carries ~1e-4 relative error there - accuracy parity w.r.t. the float64 oracle
        b64, _ = o64.block(int(l))
        n64 = np.linalg.norm(b64[:3, li + 3])
        eg = max(eg, abs(np.linalg.norm(qg[l].astype(np.float64)) - n64) / (1e-3 + n64))
        eo = max(eo, abs(np.linalg.norm(blk[:3, li + 3].astype(np.float64)) - n64) / (1e-3 + n64))
    assert eg < 2e-3 and eg <= 3 * eo + 1e-5, (eg, eo)


def test_config3_trafalgar257_f32_full_lm_run(monkeypatch):
    """BASELINE config 3: full QR + PCG pipeline, float32, 12 LM iterations against the oracle.

    Measured on this workload (scripts/traj_compare.py): with every product matrix-free
    (explicit_after = 0, the reference's algorithm step by step) the GPU reproduces the float32
    oracle's CG iteration counts exactly (3, 11, 57, 186, 274, 271, 3, 3); in float64 the default
    configuration (assembled matrix + fused PCG) reproduces the float64 oracle's counts exactly
    (185, 275, 272, 245, 329, ...). Once a float32 solve needs hundreds of iterations the Q-model
    stopping test (eta = 0.1) is decided by rounding noise in EITHER implementation (the float32
    reference stops iteration 7 after 3 CG iterations where float64 needs 245), so there only the
    costs are compared, and the final cost of both float32 runs against the float64 optimum."""
    from oracle import oracle as O
    prob = _bench_problem("trafalgar-257")
    kw = dict(max_num_iterations=12, function_tolerance=0.0)
    g, o = _pair(prob, np.float32, **kw)
    lg, tg = g.optimize_lm()
    lo, to = o.optimize_lm()
    assert lg[0].cost == pytest.approx(lo[0].cost, rel=2e-6)
    # (float32 costs: each side evaluates its sum with ~1e-6 of noise - repeated GPU runs against the same oracle run
    #  scatter between 1e-7 and 2e-6 already at iteration 1, scripts/traj_spread.py - so 4e-6 per iteration here;
    #  the bar of 1e-6 is applied below to the double-evaluated final costs)
    for a, b in zip(lg[1:7], lo[1:7]):
        assert a.step_is_successful == b.step_is_successful == 1
        assert abs(a.cost - b.cost) <= 4e-6 * b.cost
        assert abs(a.lambda_ - b.lambda_) <= 2e-2 * b.lambda_
        if b.cg_iterations <= 60:
            assert abs(a.cg_iterations - b.cg_iterations) <= 1
            assert abs(a.inc_norm - b.inc_norm) <= 1e-2 * b.inc_norm
        else:
            # noise-limited solves: repeated GPU runs stop iteration 6 after 240 or 261 CG iterations (oracle 259)
            # and the increment norms differ by up to ~10 % accordingly (scripts/traj_spread.py)
            assert 0.5 * b.cg_iterations <= a.cg_iterations <= 2 * b.cg_iterations
            assert abs(a.inc_norm - b.inc_norm) <= 0.15 * b.inc_norm
    o64 = O.Oracle(prob, np.float64, _opts(O, **kw))
    f64 = min(r.cost for r in o64.optimize_lm()[0] if r.step_is_successful)
    # north_star: same final cost within 1e-6 relative (both float32 runs vs the float64 optimum). The cost of the
    # FINAL STATES is evaluated in double: a float32 evaluation of this sum carries ~6e-7 of noise by itself
    # (residuals of ~0.5 px are differences of ~1e3 px projections), which says nothing about where the run ended.
    def cost64(lin):
        ev = O.Oracle(prob, np.float64, _opts(O, **kw))
        ev.set_state(*lin.get_state())
        return ev.compute_error().all_error
    fg, fo = cost64(g), cost64(o)
    # Measured (round 3, three GPU runs each; the float atomics of the matrix-free product make runs differ): after 12
    # iterations the float32 GPU run ends 5.3e-7 / 1.0e-6 / 1.3e-6 above the float64 optimum, the float32 oracle 6.2e-7;
    # after 16 iterations 1.7e-7 ... 8.2e-7 against 8.3e-7, after 20 2.9e-7 ... 8.5e-7 against 6.5e-7: a float32 STATE
    # resolves the optimum to ~1e-6, for either implementation, so the bar here is 2e-6 and "not worse than the float32
    # oracle by more than 1e-6"; the mixed mode (double state, tests/test_gpu_mixed.py) meets 2e-7.
    assert abs(fg - f64) / f64 < 2e-6 and abs(fo - f64) / f64 < 2e-6, (fg, fo, f64)
    assert (fg - fo) / fo < 1.5e-6
    # and the float32 costs the runs report are those costs up to that evaluation noise
    assert abs(min(r.cost for r in lg if r.step_is_successful) - fg) / fg < 2e-6

    # the reference's algorithm step by step: every product matrix-free
    monkeypatch.setenv("RBA_EXPLICIT_AFTER", "0")
    g0, o0 = _pair(prob, np.float32, max_num_iterations=6, function_tolerance=0.0)
    l0, _ = g0.optimize_lm()
    lo0, _ = o0.optimize_lm()
    for a, b in zip(l0[1:], lo0[1:]):
        assert a.step_is_successful == b.step_is_successful == 1
        # (solves of hundreds of iterations: the float atomics of the matrix-free product make two GPU runs differ - the
        #  last solve of this run stops at 270 ... 272 iterations in three runs of four and at 250 - 252 in the fourth
        #  (profiles/r4_trafalgar_default_lockstep_runs.log, gpurun_out of round 6: 252 against the oracle's 271); 2 % up
        #  to 60 iterations, 10 % beyond, as for the lock-step runs below)
        tol = b.cg_iterations // 50 if b.cg_iterations <= 60 else b.cg_iterations // 10
        assert abs(a.cg_iterations - b.cg_iterations) <= max(1, tol)
        assert abs(a.cost - b.cost) <= 4e-6 * b.cost
        assert abs(a.inc_norm - b.inc_norm) <= (1e-2 if b.cg_iterations <= 60 else 5e-2) * b.inc_norm


def test_config4_venice1778_f32_lockstep_four_iterations():
    """BASELINE config 4 (single GPU): the headline workload, iterations 1..4 against the oracle."""
    prob = _bench_problem("venice-1778")
    kw = dict(max_num_iterations=4, function_tolerance=0.0)
    g, o = _pair(prob, np.float32, **kw)
    lg, _ = g.optimize_lm()
    lo, _ = o.optimize_lm()
    assert len(lg) == len(lo) == 5
    # 5e6 float32 residuals: the cost itself resolves to ~1e-6 relative
    assert lg[0].cost == pytest.approx(lo[0].cost, rel=2e-6)
    for a, b in zip(lg[1:], lo[1:]):
        assert a.step_is_successful == b.step_is_successful == 1
        assert abs(a.cg_iterations - b.cg_iterations) <= (1 if b.cg_iterations <= 60 else b.cg_iterations // 4)
        assert abs(a.cost - b.cost) <= 2e-6 * b.cost
        # (free-running trajectories: the states differ by float32 rounding from iteration 2 on, so only the norm is
        #  compared here; the increment VECTORS are compared from identical states in
        #  test_config4_venice1778_f32_increment_vectors)
        assert abs(a.inc_norm - b.inc_norm) <= (1e-2 if b.cg_iterations <= 30 else 6e-2) * b.inc_norm
        assert abs(a.lambda_ - b.lambda_) <= 2e-2 * b.lambda_
    # size-independent properties at full size: states agree after the four accepted steps
    (cg_, lg_), (co_, lo_) = g.get_state(), o.get_state()
    assert rel_err(cg_, co_) < 1e-4 and rel_err(lg_, lo_) < 1e-4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_config5_power_series_preconditioner_at_trafalgar_size(dtype):
    """BASELINE config 5's solver path (square-root operator + PoBA power-series preconditioner,
    PowerSCPreconditioner::solve_assign, src/rootba/cg/preconditioner.hpp:180-245) at trafalgar-257
    size: the GPU applies the series through the assembled reduced matrix
    ((Hpp^-1 E0) t = t - Hpp^-1 (S + lambda I) t), the oracle through the landmark blocks."""
    prob = _bench_problem("trafalgar-257")
    kw = dict(max_num_iterations=5, function_tolerance=0.0, preconditioner_type=2, power_order=10)
    g, o = _pair(prob, dtype, **kw)
    lg, _ = g.optimize_lm()
    lo, _ = o.optimize_lm()
    assert len(lg) == len(lo) == 6
    ctol = 2e-6 if dtype == np.float32 else 1e-8
    for a, b in zip(lg[1:], lo[1:]):
        assert a.step_is_successful == b.step_is_successful == 1
        assert abs(a.cost - b.cost) <= ctol * b.cost
        if dtype == np.float64:
            assert a.cg_iterations == b.cg_iterations
        elif b.cg_iterations <= 30:
            assert abs(a.cg_iterations - b.cg_iterations) <= 1
    # and it does what a better preconditioner should: fewer PCG iterations than SCHUR_JACOBI
    gj, _ = _pair(prob, dtype, max_num_iterations=5, function_tolerance=0.0)
    lj, _ = gj.optimize_lm()
    assert sum(r.cg_iterations for r in lg) < sum(r.cg_iterations for r in lj)


# ---- vector-level lock-step at BASELINE sizes ("compare the increment vector, not its norm") ----
def test_config3_trafalgar257_f32_increment_vectors_reference_algorithm(monkeypatch):
    """Every product matrix-free (explicit_after = 0: the reference's algorithm step by step), float32, six iterations
    with 3 / 11 / 57 / 186 / 274 / 272 PCG iterations: identical PCG counts, increments no further from the float32
    oracle's than two float32 results can be (twice the oracle's own distance from float64; measured 1.0e-4 ... 1.3e-3,
    run to run - the matrix-free product flushes with float atomics), and as close to the float64 iterate as the float32 oracle is
    (measured gpu 1.3e-4 / 3.0e-4 / 1.38e-3 / 9.1e-4 / 1.29e-3 against oracle 1.2e-4 / 3.2e-4 / 1.38e-3 / 8.9e-4 / 1.26e-3)."""
    monkeypatch.setenv("RBA_EXPLICIT_AFTER", "0")
    rows = _lockstep("trafalgar-257", "float32", 6)
    assert len(rows) == 6
    for r in rows:
        assert r["termination"] == 1 and abs(r["cg_gpu"] - r["cg_oracle"]) <= 1, r
        assert r["cost_rel"] < 2e-6 and r["hx_rel"] < 1e-5 and r["l_diff_rel"] < 2e-3, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4, r
        assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r
        assert r["cams_rel"] < 1e-6 and r["lms_rel"] < 1e-5, r


def test_config3_trafalgar257_f32_increment_vectors_default_configuration():
    """The default configuration switches long solves to the ASSEMBLED reduced matrix. Rounds 2-3 held
    that matrix in float32 and accepted 2.5e-2 on the increments of such solves (S + E, |E| ~ eps |S|: 3-10 x further
    from float64 than the float32 reference, different PCG counts - VERDICT round 3, weak 1). Since round 4 the matrix
    is DOUBLE, derived from the float factors with reflectors and rotations that are orthogonal to double precision
    (kernels_a64.hpp): the default configuration is held to the SAME assertions as the all-matrix-free run above -
    PCG counts of the float32 oracle (3 / 11 / 57 / 186 / 279 / 276), increments as close to the float64 iterate as the
    float32 oracle's."""
    rows = _lockstep("trafalgar-257", "float32", 6)
    assert len(rows) == 6
    for r in rows:
        # (PCG counts of the 186 / 274 / 272-iteration solves: the oracle's - 186, 274 and 270 ... 272 in most runs; one
        #  run in four of the LAST solve stops at 250, with either form of stage 1 (profiles/r4_trafalgar_default_lockstep_runs.log, four runs: the first
        #  products of a solve are matrix-free and flush with float atomics, and the Q-model quantity hovers around its
        #  threshold for the last twenty iterations) - its increment is then 1.55e-3 from the float64 iterate against the
        #  float32 oracle's 1.39e-3, inside the accuracy assertions below, which are what this test is about. Counts
        #  within 10 %, the bound VERDICT round 3 set for counts at this length.)
        #  Solves of up to 60 iterations: the oracle's count exactly.
        band = 0 if r["cg_oracle"] <= 60 else max(1, r["cg_oracle"] // 10)
        assert r["termination"] == 1 and abs(r["cg_gpu"] - r["cg_oracle"]) <= band, r
        assert r["cost_rel"] < 2e-6 and r["hx_rel"] < 1e-5 and r["l_diff_rel"] < 2e-3, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4, r
        assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r
        assert r["cams_rel"] < 1e-6 and r["lms_rel"] < 1e-5, r


def test_config3_trafalgar257_f32_deterministic_mode(monkeypatch):
    """RBA_DETERMINISTIC=1 (matrix-free products summed camera-major in a fixed order, kernels.hpp: k_hx_det_gather;
    VERDICT round 4, next 6c) at BASELINE size: two 12-iteration float32 LM runs of the default configuration on two
    handles agree BIT BY BIT (costs, PCG counts, final cameras and landmarks) - the run-to-run spread of the default
    mode's long solves (the count of the last one: 270 ... 272, one run in four 250, see the test above) is gone. The
    lock-step rows of the deterministic mode are held to the assertions of the default mode with the count band of the
    long solves at 3 % instead of 10 %."""
    monkeypatch.setenv("RBA_DETERMINISTIC", "1")
    prob = _bench_problem("trafalgar-257")
    runs = []
    for _ in range(2):
        g, _ = _pair(prob, np.float32, max_num_iterations=12, function_tolerance=0.0)
        rows, _ = g.optimize_lm()
        runs.append((rows, g.get_state()))
        g.close()
    (ra, sa), (rb, sb) = runs
    assert [r.cg_iterations for r in ra] == [r.cg_iterations for r in rb], ([r.cg_iterations for r in ra], [r.cg_iterations for r in rb])
    assert [r.cost for r in ra] == [r.cost for r in rb]
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])
    rows = _lockstep("trafalgar-257", "float32", 6)
    assert len(rows) == 6
    for r in rows:
        band = 0 if r["cg_oracle"] <= 60 else max(1, (3 * r["cg_oracle"] + 99) // 100)
        assert r["termination"] == 1 and abs(r["cg_gpu"] - r["cg_oracle"]) <= band, r
        assert r["cost_rel"] < 2e-6 and r["hx_rel"] < 1e-5 and r["l_diff_rel"] < 2e-3, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4, r
        assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r


def test_config4_venice1778_f32_increment_vectors():
    """BASELINE config 4, the headline workload: iterations 1..3 (2 / 6 / 28 PCG iterations) in lock-step. Identical PCG
    counts; increments within twice the float32 oracle's distance from float64 of the oracle's (measured 2.1e-4 / 8.1e-4 / 6.5e-4) and as close to float64
    as the float32 oracle is (measured 1.6e-4 / 6.1e-4 / 5.1e-4 for both)."""
    rows = _lockstep("venice-1778", "float32", 3)
    assert len(rows) == 3
    for r in rows:
        assert r["termination"] == 1 and r["cg_gpu"] == r["cg_oracle"], r
        assert r["cost_rel"] < 2e-6 and r["hx_rel"] < 1e-5 and r["l_diff_rel"] < 1e-4, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4 and r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r
        assert r["cams_rel"] < 1e-6 and r["lms_rel"] < 1e-5, r


# tests/golden/lockstep_venice-1778_f32_it6.npz: iteration 6 of the venice fixture (the 336-iteration solve) on its own,
# TRACKED (11 MB, float32 state of 994 K landmarks - incompressible), so that a clean clone checks a long solve on the
# assembled matrix against the float32 and float64 oracle iterates
TRACKED_FIXTURES = {"venice-1778-it6": ("lockstep_venice-1778_f32_it6.npz",
                                        "799ec7b158a447787ec9ba052125f806de8a41e17151dbe939e4c331b46c9c8d"),
                    # the CPU float64 referee of the final-13682 lock-step (scripts/make_referee_fixture.py)
                    "final-13682-referee64": ("referee64_final-13682.npz", "2541ae685743f409c452c39f35c4367e76afca1ec513f61840978ed03a4ab870")}


def _fixture(name):
    """The oracle side of a long lock-step. The tracked fixture must be there and must be the recorded file; the big ones
    (tests/golden/_big, git-ignored, 321 MB, ~25 CPU-minutes to regenerate with scripts/make_lockstep_fixture.py) travel
    with the snapshot to the GPU box - a run WITHOUT them fails instead of passing by skipping, unless the developer says
    so (RBA_ALLOW_MISSING_FIXTURES=1)."""
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    if name in TRACKED_FIXTURES:
        fname, sha = TRACKED_FIXTURES[name]
        path = os.path.join(here, "golden", fname)
        assert os.path.exists(path), f"{path}: the tracked fixture is missing"
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == sha, f"{path} is not the recorded fixture"
        return np.load(path)
    path = os.path.join(here, "golden", "_big", f"lockstep_{name}_f32.npz")
    if not os.path.exists(path):
        if os.environ.get("RBA_ALLOW_MISSING_FIXTURES") == "1":
            pytest.skip(f"{path} is absent and RBA_ALLOW_MISSING_FIXTURES=1")
        pytest.fail(f"{path} is absent: scripts/make_lockstep_fixture.py {name} <iterations> writes it (oracle solves of "
                    "~25 CPU-minutes); RBA_ALLOW_MISSING_FIXTURES=1 skips instead")
    return np.load(path)


def _fixture_rows(name, dts="float32", tag="", **extra):
    """tests/lockstep.py with the oracle side read from the fixture: per stored iteration the GPU is set to the oracle's
    state, linearised, and solves with the oracle run's lambda; compared with the float32 oracle's increment and the
    float64 oracle's iterate of the same index (a second handle with max_cg_it = the oracle's count, eta = 0 supplies
    the GPU's iterate of that index when the counts differ)."""
    import torch  # noqa: F401
    from lockstep import rel
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    fx = _fixture(name)
    name = str(fx["workload"])
    prob = _bench_problem(name)
    kw = dict(robust_norm=1, huber_parameter=1.0, function_tolerance=0.0)
    kw.update(extra)
    gdt = "mixed" if dts == "mixed" else np.float32
    g = LinearizorHIP(prob, gdt, L.default_options(**kw))
    rows = []
    for it in fx["iterations"]:
        c_, l_ = fx[f"cams_{it}"], fx[f"lms_{it}"]
        up = (lambda a: a.astype(np.float64)) if gdt == "mixed" else (lambda a: a)
        g.set_state(up(c_), up(l_))
        eg = g.compute_error()
        assert g.linearize() == 0
        lam, n32 = float(fx[f"lambda_{it}"]), int(fx[f"cg32_{it}"])
        ig, cg = g.solve(lam)
        row = {"it": int(it), "lambda": lam, "cost_rel": float(abs(eg.all_error - fx[f"cost_{it}"]) / fx[f"cost_{it}"]),
               "cost_gpu": float(eg.all_error), "cost_oracle32": float(fx[f"cost_{it}"]),
               "cg_gpu": cg.num_iterations, "cg_oracle": n32, "termination": cg.termination_type,
               "termination_oracle": int(fx[f"term32_{it}"])}
        same = ig
        if cg.num_iterations != n32:
            gn = LinearizorHIP(prob, gdt, L.default_options(**dict(kw, max_cg_it=n32, eta=0.0)))
            gn.set_state(up(c_), up(l_))
            assert gn.linearize() == 0
            same, cn = gn.solve(lam)
            assert cn.num_iterations == n32
            del gn
        row["inc_rel"] = rel(same, fx[f"inc32_{it}"])
        if f"inc64_{it}" in fx:
            ref64 = fx[f"inc64_{it}"]
        else:
            # final-13682: the float64 oracle of the fixture run does not fit the host (55 GB of landmark blocks). Referee:
            # the float64 PCG iterate of the same index from the same state by the oracle's MATRIX-FREE Schur-complement
            # solver on the CPU (solver_type 2, scripts/make_referee_fixture.py; tests/test_oracle_referee.py holds it
            # to the other two oracle solvers where they fit) - independent of the HIP library (rounds 3-4 used a float64
            # run of the library itself here).
            referee = _fixture(f"{name}-referee64")
            ref64 = referee[f"inc64_{it}"]
            row["referee"] = "cpu float64, matrix-free Schur complement"
            # the float64 cost of the state: what both float32 costs are held to (round 6)
            c64 = float(referee[f"cost64_{it}"])
            row["cost64"] = c64
            row["cost_gpu_vs_f64"] = abs(row["cost_gpu"] - c64) / c64
            row["cost_oracle32_vs_f64"] = abs(row["cost_oracle32"] - c64) / c64
            # where the error of either float32 increment sits: share of |inc - inc64|^2 in its three worst cameras
            for key, v in (("gpu", same), ("oracle32", fx[f"inc32_{it}"])):
                d = ((np.asarray(v, np.float64) - ref64).reshape(-1, 9) ** 2).sum(axis=1)
                j = np.argsort(-d)[:3]
                row[f"{key}_err_top_cameras"] = [[int(c), round(float(d[c] / d.sum()), 4)] for c in j]
        row["gpu_vs_f64"] = rel(same, ref64)
        row["oracle32_vs_f64"] = rel(fx[f"inc32_{it}"], ref64)
        row["own_vs_f64"] = rel(ig, ref64)  # the increment the solve returned (its own stopping index)
        l_diff = g.apply(fx[f"inc32_{it}"])
        row["l_diff_rel"] = float(abs(l_diff - fx[f"l_diff_{it}"]) / abs(fx[f"l_diff_{it}"]))
        rows.append(row)
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):  # (the numbers DESIGN.md quotes)
        with open(os.path.join(out, f"fixture_lockstep_{name}_{dts}{'_' + tag if tag else ''}.jsonl"), "w") as f:
            f.writelines(json.dumps(r) + "\n" for r in rows)
    return rows


def test_config4_venice1778_f32_increment_vectors_long_solves():
    """The solves that dominate the timed region of the headline bench: venice-1778
    iterations 3..8 of the float32 oracle's run (28 / 124 / 330 / 480 / ... PCG iterations), default configuration
    (assembled double matrix after the measured break-even), from the oracle's states (fixture, see _fixture_rows).
    Same assertions as the short solves: the oracle's PCG count within 2 %, the iterate of the oracle's index as close
    to float64 as the float32 oracle's."""
    rows = _fixture_rows("venice-1778")
    assert len(rows) >= 5
    for r in rows:
        # (iteration 8 runs into max_linear_solver_iterations = 500 in the oracle too: NO_CONVERGENCE for both)
        assert r["termination"] == r["termination_oracle"], r
        assert abs(r["cg_gpu"] - r["cg_oracle"]) <= max(1, r["cg_oracle"] // 50), r
        assert r["cost_rel"] < 2e-6 and r["l_diff_rel"] < 2e-3, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4, r
        assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r


def test_config4_venice1778_f32_long_solve_tracked_fixture():
    """The same assertions on the one long solve whose fixture is tracked (iteration 6 of the run: 336 PCG iterations on
    the assembled double matrix through the persistent kernel) - the check a clean clone can make."""
    rows = _fixture_rows("venice-1778-it6", tag="it6")
    assert len(rows) == 1 and rows[0]["cg_oracle"] == 336
    r = rows[0]
    assert r["termination"] == r["termination_oracle"] == 1, r
    assert abs(r["cg_gpu"] - r["cg_oracle"]) <= max(1, r["cg_oracle"] // 50), r
    assert r["cost_rel"] < 2e-6 and r["l_diff_rel"] < 2e-3, r
    assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4, r
    assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r


def test_config5_final13682_f32_lockstep_iterations_3_to_7(monkeypatch):
    """final-13682 (BASELINE config 5's size) in lock-step with the float32 oracle over LM
    iterations 3..7 - the range where the round-3 GPU run left the oracle's trajectory (13 instead of 49 PCG iterations at
    iteration 4, then a rejected step with cost 3.6e14) - from the oracle's states (fixture; the float64 oracle does not
    fit the host: increments are compared with the float32 oracle's iterate of the same index only).
    Round 6: the fixture is a self-consistent LM run of the oracle driven by scripts/make_lockstep_fixture.py (PCG counts
    22 / 50 / 5 / 3 / 137, every step accepted); the paragraph below is about the fixture of rounds 4-5.

    What the fixture itself shows (profiles/r4_final13682_oracle_f32_lm_and_replay.log): the float32 oracle's PCG counts
    at this size are NOT reproducible between two runs of the oracle - its LM run takes 22 / 49 / 5 / 3 / 2 iterations
    where its replay from the same states (same code, another OpenMP summation order) takes 23 / 36 / 5 / 3 / 137: the
    Q-model stopping test (eta = 0.1) is decided by float32 rounding there. So the counts are held to the factor two
    the oracle differs from itself by, the increment of the ORACLE'S index to float32 accuracy - with every product
    matrix-free (the reference algorithm) and in the default configuration (assembled double matrix)."""
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RBA_EXPLICIT_AFTER", raising=False)
        else:
            monkeypatch.setenv("RBA_EXPLICIT_AFTER", env)
        rows = _fixture_rows("final-13682", tag="matrix_free" if env else "default")
        assert len(rows) == 5
        for r in rows:
            assert r["termination"] == 1, r
            assert 0.5 * r["cg_oracle"] - 1 <= r["cg_gpu"] <= 2 * r["cg_oracle"] + 1, r
            # accuracy parity against the float64 referee - since round 5 the CPU's: the oracle's matrix-free
            # Schur-complement solver in float64 (tests/golden/referee64_final-13682.npz; it reproduces the numbers the
            # float64 run of the HIP library gave as referee in round 4 to three digits on iterations 3 - 6 and gives
            # 6.2e-3 instead of 8.8e-3 for the float32 oracle's 137-iteration solve) -, iterate of the oracle's index
            # EVERY iteration since round 6. What the exempted state was, by measurement
            # (profiles/r6_final13682_iteration6_diagnosis.txt): (a) rounds 4-5 replayed the lambda schedule of a SEPARATE
            # oracle LM run and went on from a step the LM loop would have rejected; scripts/make_lockstep_fixture.py now
            # drives a self-consistent LM run, every stored state is one the loop visits. (b) The state of iteration 6
            # STILL separated the two float32 results (GPU 8.4e-3, oracle 1.1e-3 from float64; costs 3e-5 apart): step 5
            # moves two 2-observation landmarks onto a camera plane (depth +-0.008 at |p_w| = 126, residuals of thousands
            # of pixels). p_c = R p_w + t in float arithmetic carries an absolute error of 1e-7 |p_w| ~ 1e-5 whatever the
            # depth: those two landmarks' rows are wrong by 1e-3, they are 2.4e-5 of the float32 oracle's cost error and
            # 28 % + 4 % of its increment error sits in their two cameras. Not the summation order of the QR (ensemble of
            # 12288 nearly rank-deficient blocks, tests/test_gpu_qr_accuracy.py: the fused order is not worse). Fixed
            # where it arises: the library evaluates p_c in double from the float state (device_utils.hpp,
            # camera_frame_point) - so at such a state it is CLOSER to float64 than the reference's float32 arithmetic,
            # and both float32 costs are held to the float64 cost of the state, not to each other.
            assert r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 2e-4, r
            assert r["cost_gpu_vs_f64"] <= max(3e-6, 1.5 * r["cost_oracle32_vs_f64"]), r
            assert r["l_diff_rel"] < 5e-3, r


def test_config5_mixed_precision_with_power_series_at_trafalgar_size():
    """BASELINE config 5's actual combination - mixed f32/f64 state + PoBA power-series preconditioner - at trafalgar-257
    size against the float32 oracle from identical (float-representable) states: PCG counts within 5 % (2 / 6 / 29 / 92 /
    97); the series and, from the second product on, the operator run through the assembled DOUBLE matrix
    (kernels_a64.hpp), so the increments are held to the accuracy parity of the matrix-free runs (round 3, float32
    matrix: 2.5e-2 stated, 3.9e-3 / 6.1e-3 measured on the long solves)."""
    rows = _lockstep("trafalgar-257", "mixed", 5, precond=2)
    assert len(rows) == 5
    for r in rows:
        assert r["termination"] == 1 and abs(r["cg_gpu"] - r["cg_oracle"]) <= max(1, r["cg_oracle"] // 20), r
        assert r["cost_rel"] < 5e-6 and r["hx_rel"] < 1e-5, r
        assert r["inc_rel"] <= 2 * r["oracle32_vs_f64"] + 2e-4 and r["gpu_vs_f64"] <= 1.5 * r["oracle32_vs_f64"] + 1e-4, r
        assert r["cams_rel"] < 1e-6 and r["lms_rel"] < 1e-5, r


@pytest.mark.timeout(1500)
def test_config5_final13682_mixed_power_series_two_ranks_split_products(monkeypatch):
    """BASELINE config 5 as it would run on a node: final-13682, mixed precision, PoBA power
    series (order 10), landmarks sharded over TWO ranks - here two ranks on the one GPU behind a single handle
    (rba_create_sharded with a repeated device id: host-memory exchange) - with the products on the assembled matrix
    split over the ranks (RBA_PCG_SPLIT=1). The four LM iterations the float32 CPU oracle's run covers before its
    trajectory turns chaotic (profiles/r4_final13682_oracle_f32_power_lm.log, 2 h 47 min of CPU): PCG counts
    2 / 2 / 11 / 3 and the costs to 5e-5."""
    import torch  # noqa: F401
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    oracle_rows = [(2, 5988292.093465802), (2, 5899113.737017494), (11, 5812048.237415455), (3, 5720012.289130197)]
    monkeypatch.setenv("RBA_PCG_SPLIT", "1")
    prob = _bench_problem("final-13682")
    g = LinearizorHIP(prob, "mixed", _opts(L, max_num_iterations=4, function_tolerance=0.0, preconditioner_type=2,
                                          power_order=10), devices=[0, 0])
    cuts = g.shard_ranges()
    assert len(cuts) == 3 and 0 < cuts[1] < prob.n_lms, cuts
    rows, _ = g.optimize_lm()
    pcg = g.pcg_counters()
    g.close()
    rows = [r for r in rows if r.iteration >= 1]
    assert len(rows) == 4
    for r, (n, cost) in zip(rows, oracle_rows):
        assert r.step_is_successful == 1 and r.cg_iterations == n, (r.iteration, r.cg_iterations, n)
        assert abs(r.cost - cost) <= 5e-5 * cost, (r.iteration, r.cost, cost)
    # the series ran through the assembled matrix, its products split over the two ranks
    assert pcg["assemblies"] >= 1 and pcg["products_assembled"] > 0, pcg
