"""Oracle side of the vector-level lock-step at BASELINE sizes, computed ONCE on the host cores of the development
container and stored as a fixture, so that the GPU tests do not spend rationed GPU-box minutes on CPU solves
(tests/lockstep.py does both sides live: venice-1778 iterations 4..8 - 124 / 330 / 480 PCG iterations in float32 AND
float64 - are ~20 minutes of oracle time).

Per LM iteration of a float32 oracle run (function_tolerance = 0): the state both sides are linearised at, lambda, the
float32 oracle's increment and PCG count, and the FLOAT64 oracle's iterate of the same index from the same state
(float Jacobian-scaling epsilon). Written to tests/golden/_big/lockstep_<workload>_f32.npz (git-ignored: 12 MB per
venice state; travels to the GPU box with the snapshot; tests that need it skip when it is absent - regenerate with this
script, ~25 min for venice on 8 cores).

usage: python scripts/make_lockstep_fixture.py <workload> <iterations> [first_stored_iteration]"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from lockstep import EPS_SQRT_FLOAT  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    name, n_it = sys.argv[1], int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    with_f64 = os.environ.get("FIXTURE_F64", "1") == "1"
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    prob = bench.make_problem(name, args)[0]
    kw = dict(robust_norm=1, huber_parameter=1.0, max_num_iterations=n_it, function_tolerance=0.0)
    self_lm = os.environ.get("FIXTURE_SELF_LM", "1") == "1"
    o2 = O.Oracle(prob, np.float32, O.default_options(**kw))
    if self_lm:
        # Round 6: the replay IS an LM run - optimize_lm_ours (bal_bundle_adjustment.cpp:291-521) driven from here through
        # the oracle's Linearizor calls, so that every stored state is one the LM loop visits. (Rounds 4-5 replayed the
        # lambda schedule and the accept decisions of a separate oracle LM run; at final-13682 size two float32 runs of the
        # oracle leave each other's trajectory after iteration 3, and the replay then went on from a step the LM loop
        # would have rejected - the "iteration 6" state that needed an exemption in tests/test_gpu_baseline_configs.py.)
        opt = O.default_options(**kw)
        lam, vee = np.float32(1.0 / opt.initial_trust_region_radius), np.float32(opt.initial_vee)
        min_lam, max_lam = np.float32(1.0 / opt.max_trust_region_radius), np.float32(1.0 / opt.min_trust_region_radius)
        out = {"workload": name, "mode": "self-consistent LM run driven by this script"}
        lambdas, cgs, oks, costs = [], [], [], []
        its = []
        cost = o2.compute_error().all_error
        it = 1
        while it <= n_it:
            t0 = time.time()
            c_, l_ = o2.get_state()
            assert o2.linearize() == 0
            while it <= n_it:  # inner loop: same linearisation point until a step is accepted
                t1 = time.time()
                io, co = o2.solve(float(lam))
                if it >= first:
                    its.append(it)
                    out[f"cams_{it}"], out[f"lms_{it}"] = c_.copy(), l_.copy()
                    out[f"lambda_{it}"], out[f"cost_{it}"] = float(lam), cost
                    out[f"inc32_{it}"], out[f"cg32_{it}"], out[f"term32_{it}"] = io.copy(), co.num_iterations, co.termination_type
                    if with_f64:
                        okw = dict(kw, max_cg_it=co.num_iterations, eta=0.0, jacobi_scaling_eps=EPS_SQRT_FLOAT)
                        o64 = O.Oracle(prob, np.float64, O.default_options(**okw))
                        o64.set_state(c_.astype(np.float64), l_.astype(np.float64))
                        assert o64.linearize() == 0
                        ref64, cn = o64.solve(float(lam))
                        assert cn.num_iterations == co.num_iterations
                        out[f"inc64_{it}"] = ref64.copy()
                        del o64
                o2.backup()
                l_diff = np.float32(o2.apply(io))
                if it >= first:
                    out[f"l_diff_{it}"] = float(l_diff)
                new_cost = o2.compute_error().all_error
                rho = np.float32(cost - new_cost) / l_diff
                ok = bool(np.isfinite(l_diff) and l_diff > 0 and rho > 0)
                lambdas.append(float(lam)), cgs.append(co.num_iterations), oks.append(int(ok)), costs.append(new_cost)
                print(f"it {it}: lambda {float(lam):.3e}, cg {co.num_iterations}, cost {cost:.6e} -> {new_cost:.6e} "
                      f"{'accepted' if ok else 'REJECTED'}, {time.time() - t1:.0f} s", flush=True)
                it += 1
                if ok:
                    lam = max(min_lam, lam * np.float32(max(1.0 / 3, 1 - (2 * float(rho) - 1) ** 3)))
                    vee = np.float32(opt.initial_vee)
                    cost = new_cost
                    break
                o2.restore()
                lam = vee * lam
                vee = vee * np.float32(opt.vee_factor)
                if lam > max_lam:
                    it = n_it + 1
        out.update(lambdas=np.array(lambdas), cg_lm_run=np.array(cgs), ok_lm_run=np.array(oks), costs_lm_run=np.array(costs))
    else:
        t0 = time.time()
        lo, _ = O.Oracle(prob, np.float32, O.default_options(**kw)).optimize_lm()
        print(f"oracle LM run: {time.time() - t0:.0f} s, cg {[r.cg_iterations for r in lo]}, ok {[r.step_is_successful for r in lo]}",
              flush=True)
        out = {"workload": name, "lambdas": np.array([r.lambda_ for r in lo]), "cg_lm_run": np.array([r.cg_iterations for r in lo]),
               "ok_lm_run": np.array([r.step_is_successful for r in lo]), "costs_lm_run": np.array([r.cost for r in lo])}
        its = []
        for r in lo[1:]:
            t0 = time.time()
            c_, l_ = o2.get_state()
            e = o2.compute_error()
            assert o2.linearize() == 0
            lam = float(r.lambda_)
            io, co = o2.solve(lam)
            it = r.iteration
            if it >= first:
                its.append(it)
                out[f"cams_{it}"], out[f"lms_{it}"] = c_.copy(), l_.copy()
                out[f"lambda_{it}"], out[f"cost_{it}"] = lam, e.all_error
                out[f"inc32_{it}"], out[f"cg32_{it}"], out[f"term32_{it}"] = io.copy(), co.num_iterations, co.termination_type
                if with_f64:
                    okw = dict(kw, max_cg_it=co.num_iterations, eta=0.0, jacobi_scaling_eps=EPS_SQRT_FLOAT)
                    o64 = O.Oracle(prob, np.float64, O.default_options(**okw))
                    o64.set_state(c_.astype(np.float64), l_.astype(np.float64))
                    assert o64.linearize() == 0
                    ref64, cn = o64.solve(lam)
                    assert cn.num_iterations == co.num_iterations
                    out[f"inc64_{it}"] = ref64.copy()
                    del o64
                    a, b = np.asarray(io, np.float64), ref64
                    print(f"  it {it}: oracle32 vs f64 {np.linalg.norm(a - b) / (np.linalg.norm(a) + np.linalg.norm(b)):.3e}", flush=True)
            l_diff = o2.apply(io)
            out[f"l_diff_{it}"] = l_diff
            print(f"it {it}: lambda {lam:.3e}, cg {co.num_iterations}, {time.time() - t0:.0f} s", flush=True)
            if not r.step_is_successful:
                break
    out["iterations"] = np.array(its)
    d = os.path.join(ROOT, "tests", "golden", "_big")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"lockstep_{name}_f32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 20, "MB")


if __name__ == "__main__":
    main()
