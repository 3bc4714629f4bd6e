"""Vector-level lock-step of the HIP library against the CPU oracle (helper of tests/test_gpu_baseline_configs.py and
scripts/lockstep_vectors.py).

Every iteration both sides start from the IDENTICAL state (the oracle's, copied to the GPU - two float32 landmark
updates differ by rounding, ~1e-6 of the scene size = ~1e-2 px in the next residuals, which would otherwise be compared
as a solver difference), are linearised there, solve with the lambda schedule of an oracle LM run, and the oracle's
increment is applied to both. Compared per iteration: cost, PCG iteration count, the increment VECTOR against

  * the oracle's iterate of the same precision and the SAME iteration index (its own increment when the counts
    agree, else a re-run with max_cg_it = the GPU's count and eta = 0), and
  * (float32 / mixed) the FLOAT64 oracle's iterate of that index from the same state with the float scaling epsilon:
    `gpu_vs_f64` against `oracle32_vs_f64` says whether the GPU's float32 result is as accurate as the reference
    algorithm's float32 result,

one H x for a random x, the model cost change and the states after the landmark / camera update."""
import time

import numpy as np

EPS_SQRT_FLOAT = 3.1622776601683794e-3  # Sophus epsilonSqrt<float>: the Jacobian-scaling epsilon of a float run


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(a) + np.linalg.norm(b)))


def lockstep_rows(prob, dts, n_it, precond=1, with_f64=True, **extra):
    """dts: 'float32' | 'float64' | 'mixed'. Yields one dict per LM iteration (first: the oracle LM run's summary)."""
    import torch  # noqa: F401
    from oracle import oracle as O
    from rootba_amd import _lib as L
    from rootba_amd.linearizor import LinearizorHIP
    odt = np.float64 if dts == "float64" else np.float32
    gdt = "mixed" if dts == "mixed" else odt
    kw = dict(robust_norm=1, huber_parameter=1.0, max_num_iterations=n_it, function_tolerance=0.0,
              preconditioner_type=precond)
    kw.update(extra)
    t0 = time.time()
    lo, _ = O.Oracle(prob, odt, O.default_options(**kw)).optimize_lm()
    yield {"oracle_lm_seconds": time.time() - t0, "lambdas": [float(r.lambda_) for r in lo],
           "cg": [r.cg_iterations for r in lo], "ok": [r.step_is_successful for r in lo]}
    g = LinearizorHIP(prob, gdt, L.default_options(**kw))
    o2 = O.Oracle(prob, odt, O.default_options(**kw))
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(odt)
    for r in lo[1:]:
        t0 = time.time()
        c_, l_ = o2.get_state()
        if gdt == "mixed":
            g.set_state(c_.astype(np.float64), l_.astype(np.float64))
        else:
            g.set_state(c_, l_)
        eg, eo = g.compute_error(), o2.compute_error()
        assert g.linearize() == 0 and o2.linearize() == 0
        lam = float(r.lambda_)
        ig, cg = g.solve(lam)
        io, co = o2.solve(lam)
        row = {"it": r.iteration, "lambda": lam, "cost_rel": float(abs(eg.all_error - eo.all_error) / eo.all_error),
               "cg_gpu": cg.num_iterations, "cg_oracle": co.num_iterations, "termination": cg.termination_type,
               "inc_norm_rel": float(abs(np.linalg.norm(ig) - np.linalg.norm(io)) / np.linalg.norm(io))}
        okw = dict(kw, max_cg_it=cg.num_iterations, eta=0.0)
        ref = io
        if cg.num_iterations != co.num_iterations:
            on = O.Oracle(prob, odt, O.default_options(**okw))
            on.set_state(c_, l_)
            assert on.linearize() == 0
            ref, cn = on.solve(lam)
            assert cn.num_iterations == cg.num_iterations
        row["inc_rel"] = rel(ig, ref)
        if odt == np.float32 and with_f64:
            o64 = O.Oracle(prob, np.float64, O.default_options(**dict(okw, jacobi_scaling_eps=EPS_SQRT_FLOAT)))
            o64.set_state(c_.astype(np.float64), l_.astype(np.float64))
            assert o64.linearize() == 0
            ref64, cn = o64.solve(lam)
            row["gpu_vs_f64"], row["oracle32_vs_f64"] = rel(ig, ref64), rel(ref, ref64)
        row["hx_rel"] = rel(g.right_multiply(x), o2.right_multiply(x))
        ldg, ldo = g.apply(io), o2.apply(io)
        row["l_diff_rel"] = float(abs(ldg - ldo) / abs(ldo))
        (cg_, lg_), (co_, lo_) = g.get_state(), o2.get_state()  # (the update itself: float32 resolution)
        row["cams_rel"], row["lms_rel"] = rel(cg_, co_), rel(lg_, lo_)
        row["seconds"] = time.time() - t0
        yield row
        if not r.step_is_successful:
            break
