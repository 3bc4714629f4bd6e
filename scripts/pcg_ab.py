"""A/B timing of the PCG on the assembled reduced matrix under environment switches (development aid):
every variant builds its own solver, linearises once and runs 500-iteration solves that switch to
the assembled matrix after the first product; prints the solve time and microseconds per iteration.
usage: python scripts/pcg_ab.py [workload] "ENV=val,ENV=val" ...   ('' = defaults)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from rootba_amd import problem as P
from rootba_amd.linearizor import LinearizorHIP
from rootba_amd import _lib as L


def run(prob, envs, dt=np.float32, lam=1e-6, reps=3, iters=500):
    saved = {}
    for kv in [e for e in envs.split(",") if e]:
        k, v = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        out = dict(env=envs, solve_ms=[], us_per_it=[])
        for n_it in (100, iters):
            g = LinearizorHIP(prob, dt, L.default_options(robust_norm=1, huber_parameter=1.0, max_cg_it=n_it, min_cg_it=n_it,
                                                          eta=1e-30, explicit_after=1))
            g.compute_error()
            assert g.linearize() == 0
            ts = []
            for _ in range(reps):
                inc, cg = g.solve(lam)
                ts.append(1e3 * g.timings().solve_reduced_system_time)
            out["solve_ms"].append([round(t, 3) for t in ts])
            out.setdefault("cg", []).append(cg.num_iterations)
        a, b = min(out["solve_ms"][0]), min(out["solve_ms"][1])
        out["us_per_it"] = round(1e3 * (b - a) / (out["cg"][1] - out["cg"][0]), 2)
        return out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
    variants = sys.argv[2:] or [""]
    prob = P.preprocess(P.named_synthetic(name), translation_sigma=0.5, point_sigma=0.5)
    for v in variants:
        print(json.dumps(run(prob, v)), flush=True)
