"""The INDEPENDENT float64 referee of the final-13682 lock-step: for every iteration of
tests/golden/_big/lockstep_final-13682_f32.npz (the float32 oracle's states, lambdas and PCG counts) the float64 PCG
iterate of the SAME index from the SAME state, computed on the CPU by the oracle's matrix-free Schur-complement solver
(oracle/rootba_oracle.hpp, solver_type 2: 5.6 GB of per-observation Jacobians instead of the 55 GB of dense landmark
blocks / 121 GB of dense H_pp the other two oracle solvers need at this size; checked against both at sizes they fit,
tests/test_oracle_referee.py). Float scaling epsilon, as the float32 runs it referees.
Round 6: also the float64 COST of every stored state (`cost64_<it>`; the float32 cost of a state with landmarks next to a
camera plane is itself uncertain at 1e-5, tests/test_gpu_baseline_configs.py holds both float32 costs to this one).
Writes tests/golden/referee64_<workload>.npz (tracked: 13682 x 9 doubles per iteration).
usage: python scripts/make_referee_fixture.py [final-13682] [--costs-only]   (--costs-only: keep the stored increments)"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from lockstep import EPS_SQRT_FLOAT  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    costs_only = "--costs-only" in sys.argv
    name = argv[0] if argv else "final-13682"
    fx = np.load(os.path.join(ROOT, "tests", "golden", "_big", f"lockstep_{name}_f32.npz"))
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    prob = bench.make_problem(name, args)[0]
    out = {"workload": name, "iterations": fx["iterations"]}
    path = os.path.join(ROOT, "tests", "golden", f"referee64_{name}.npz")
    old = dict(np.load(path)) if costs_only else None
    for it in fx["iterations"]:
        t0 = time.time()
        lam, n32 = float(fx[f"lambda_{it}"]), int(fx[f"cg32_{it}"])
        o = O.Oracle(prob, np.float64, O.default_options(robust_norm=1, huber_parameter=1.0, solver_type=2,
                                                        max_cg_it=n32, eta=0.0, jacobi_scaling_eps=EPS_SQRT_FLOAT))
        o.set_state(fx[f"cams_{it}"].astype(np.float64), fx[f"lms_{it}"].astype(np.float64))
        out[f"cost64_{it}"] = float(o.compute_error().all_error)
        if costs_only:
            out[f"inc64_{it}"] = old[f"inc64_{it}"]
            c32 = float(fx[f"cost_{it}"])
            print(f"iteration {int(it)}: float64 cost {out[f'cost64_{it}']:.4f}, float32 oracle's {c32:.4f} "
                  f"({abs(c32 - out[f'cost64_{it}']) / out[f'cost64_{it}']:.2e})", flush=True)
            del o
            continue
        assert o.linearize() == 0
        inc, cg = o.solve(lam)
        assert cg.num_iterations == n32, (cg.num_iterations, n32)
        out[f"inc64_{it}"] = np.asarray(inc, np.float64).copy()
        i32 = np.asarray(fx[f"inc32_{it}"], np.float64)
        d = np.linalg.norm(i32 - inc) / (np.linalg.norm(i32) + np.linalg.norm(inc))
        print(f"iteration {int(it)}: lambda {lam:.3e}, {n32} PCG iterations, float32 oracle vs this referee {d:.3e} "
              f"({time.time() - t0:.0f} s)", flush=True)
        del o
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
