"""Cost of the DROP-IN path: the reference's own LM loop (optimize_lm_ours, compiled from /root/reference by
oracle/build_ref.sh) driving the HIP library through the reference-side binding
(integration/rootba/solver/linearizor_hip.hpp: Linearizor::{compute_error, linearize, solve, apply}), against the
library's resident LM loop rba_optimize_lm on the same problem ("next round" 8).
usage: python scripts/dropin_bench.py [workload] [iterations]   -> one JSON line"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
from oracle import ref as R
from rootba_amd import _lib as L
from rootba_amd.linearizor import LinearizorHIP


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    prob = bench.make_problem(name, args)[0]
    kw = dict(robust_norm=1, huber_parameter=1.0, max_num_iterations=n_it, function_tolerance=0.0)
    out = {"workload": name, "iterations": n_it}
    # the reference's loop through the binding
    R.binding_lib("hip")
    t0 = time.perf_counter()
    h = R.ReferenceOnHip(prob, np.float32, R.default_options(**kw), provider="hip")
    out["reference_problem_setup_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    rows, term = h.optimize_lm()
    wall = time.perf_counter() - t0
    its = [r for r in rows if r.iteration >= 1]
    t_it = sum(r.iteration_time for r in its)
    out["drop_in"] = {"lm_iterations": len(its), "wall_s_incl_rba_create": wall, "sum_iteration_time_s": t_it,
                      "it_per_s": len(its) / t_it, "cg_iterations": sum(r.cg_iterations for r in its),
                      "final_cost": [r.cost for r in rows if r.step_is_successful][-1],
                      # where the reference's loop spends an iteration: the four Linearizor calls as its IterationSummary
                      # timed them (wall time around each call of the binding, synchronisations and state protocol
                      # included), and what is left - the reference's own host code (BalProblem::backup() walks 1 M
                      # landmark structs, bal_problem.cpp:590-598; logging)
                      "ms_per_iteration": {"total": 1e3 * t_it / len(its),
                                           "compute_error": 1e3 * sum(r.residual_time for r in its) / len(its),
                                           "linearize": 1e3 * sum(r.stage1_time for r in its) / len(its),
                                           "solve": 1e3 * sum(r.pcg_time for r in its) / len(its),
                                           "apply": 1e3 * sum(r.backsub_time for r in its) / len(its)}}
    m = out["drop_in"]["ms_per_iteration"]
    m["reference_host_code"] = m["total"] - m["compute_error"] - m["linearize"] - m["solve"] - m["apply"]
    # the library's own loop (state resident, no Linearizor interface in between)
    g = LinearizorHIP(prob, np.float32, L.default_options(**kw))
    g.lm_begin()
    g.lm_step()
    g.synchronize()
    t0 = time.perf_counter()
    n, cg, more = 0, 0, True
    while more:
        row, more = g.lm_step()
        n += 1
        cg += row.cg_iterations
    g.synchronize()
    t = time.perf_counter() - t0
    out["rba_optimize_lm"] = {"lm_iterations": n, "seconds": t, "it_per_s": n / t, "cg_iterations": cg}
    out["ratio"] = out["drop_in"]["it_per_s"] / out["rba_optimize_lm"]["it_per_s"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
