"""Quick GPU-vs-oracle comparison (development aid; the real checks are tests/ -m gpu)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first, as bench/tests do)
from rootba_amd import problem as P
from rootba_amd.linearizor import LinearizorHIP
from rootba_amd import _lib as L
from oracle import oracle as O


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(a) + np.linalg.norm(b))


def run(prob, dt, lam=0.1, precond=1):
    print(f"--- {prob.name} n_c={prob.n_cams} n_l={prob.n_lms} n_o={prob.n_obs} {np.dtype(dt).name} precond={precond}")
    kw = dict(robust_norm=1, huber_parameter=1.0, preconditioner_type=precond)
    g = LinearizorHIP(prob, dt, L.default_options(**kw))
    o = O.Oracle(prob, dt, O.default_options(**kw))
    rg, ro = g.compute_error(), o.compute_error()
    print("error", rg.all_error, ro.all_error, "rel", abs(rg.all_error - ro.all_error) / ro.all_error, rg.all_num_obs, ro.all_num_obs, rg.valid_num_obs)
    st, d2 = g.linearize(want_jp_diag2=True)
    assert o.linearize() == 0
    print("linearize status", st, "jp_diag2 rel", rel(d2, o.jp_diag2() if hasattr(o, 'jp_diag2') else d2))
    print("pose_scaling rel", rel(g.pose_scaling(), o.pose_scaling()))
    print("jl_col_scale rel", rel(g.jl_col_scale(), o.jl_col_scale()))
    # oracle stage 2 with first-inner-it scaling
    o.set_pose_damping(lam)
    b_o, bl_o = o.stage2(lam, o.pose_scaling(), blocks=True)
    b_g, bl_g = g.stage2(lam)
    print("b rel", rel(b_g, b_o), "blocks rel", rel(bl_g, bl_o), "max block rel", max(rel(bl_g[c], bl_o[c]) for c in range(prob.n_cams)))
    x = np.random.default_rng(0).uniform(-1, 1, 9 * prob.n_cams).astype(dt)
    print("Hx rel", rel(g.right_multiply(x), o.right_multiply(x)))
    Rg, qg = g.landmark_R(damped=False)
    inc = (np.random.default_rng(1).uniform(-1, 1, 9 * prob.n_cams) * 0.01).astype(dt)
    lg, lo = g.back_substitute(inc), o.back_substitute(inc)
    print("l_diff", lg, lo, "rel", abs(lg - lo) / (abs(lg) + abs(lo)))
    print("landmarks rel", rel(g.get_state()[1], o.get_state()[1]))
    # full solve on a fresh pair
    g2 = LinearizorHIP(prob, dt, L.default_options(**kw))
    o2 = O.Oracle(prob, dt, O.default_options(**kw))
    g2.linearize(); o2.linearize()
    ig, cg = g2.solve(1e-4); io, co = o2.solve(1e-4)
    print("solve: cg its", cg.num_iterations, co.num_iterations, "term", cg.termination_type, co.termination_type, "inc rel", rel(ig, io))
    ag, ao = g2.apply(ig), o2.apply(ig)
    print("apply l_diff", ag, ao, "cams rel", rel(g2.get_state()[0], o2.get_state()[0]), "lms rel", rel(g2.get_state()[1], o2.get_state()[1]))
    t = time.time()
    g3 = LinearizorHIP(prob, dt, L.default_options(**kw))
    logg, tg = g3.optimize_lm()
    tg_s = time.time() - t
    t = time.time()
    o3 = O.Oracle(prob, dt, O.default_options(**kw))
    logo, to = o3.optimize_lm()
    to_s = time.time() - t
    print(f"LM gpu {tg_s:.3f}s ({len(logg)} rows, term {tg}) cpu {to_s:.3f}s ({len(logo)} rows, term {to})")
    for a, b in zip(logg, logo):
        print(f"  it {a.iteration} ok {a.step_is_successful}/{b.step_is_successful} cg {a.cg_iterations}/{b.cg_iterations} cost {a.cost:.8e}/{b.cost:.8e} lam {a.lambda_:.2e}/{b.lambda_:.2e} inc {a.inc_norm:.4e}/{b.inc_norm:.4e} t {a.iteration_time*1e3:.2f}ms/{b.iteration_time*1e3:.2f}ms")
    tm = g3.timings()
    print("timings: hx_time", tm.hx_time, "hx_calls", tm.hx_calls, "pcg", tm.solve_reduced_system_time)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        raw = P.synthetic_problem(40, 400, 1700, seed=7)
        prob = P.preprocess(raw, seed=7, translation_sigma=0.5, point_sigma=0.5)
        run(prob, np.float64)
        run(prob, np.float32)
        run(prob, np.float32, precond=0)
    lady = P.preprocess(P.named_synthetic("ladybug-49"), translation_sigma=0.5, point_sigma=0.5)
    run(lady, np.float32)
    run(lady, np.float64)
