"""A/B timing of the matrix-free product and the back-substitution under environment switches
(development aid): every variant builds its own solver on the same workload, runs one
linearisation, one PCG solve with all products matrix-free (HIP events around every product)
and one apply; prints average microseconds per product / per stage.
usage: python scripts/hx_ab.py [workload] "ENV=val,ENV=val" "ENV=val" ...   ('' = defaults)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first, as bench/tests do)
from rootba_amd import problem as P
from rootba_amd.linearizor import LinearizorHIP
from rootba_amd import _lib as L


def run(prob, envs, dt=np.float32, lam=1e-4, reps=3):
    saved = {}
    for kv in [e for e in envs.split(",") if e]:
        k, v = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    os.environ["RBA_EXPLICIT_AFTER"] = "0"
    os.environ["RBA_HX_TIMING_STRIDE"] = "1"
    try:
        g = LinearizorHIP(prob, dt, L.default_options(robust_norm=1, huber_parameter=1.0, max_cg_it=30, min_cg_it=30))
        out = dict(env=envs, hx_us=[], bs_us=[], s1_us=[], s2_us=[])
        inc_ref = None
        for _ in range(reps):
            g.compute_error()
            assert g.linearize() == 0
            inc, cg = g.solve(lam)
            tm = g.timings()
            out["hx_us"].append(1e6 * tm.hx_time / max(1, tm.hx_calls))
            out["s1_us"].append(1e6 * tm.stage1_time)
            out["s2_us"].append(1e6 * tm.stage2_time)
            inc_ref = inc if inc_ref is None else inc_ref
            g.apply(np.zeros_like(inc))  # state unchanged: the next repetition linearises the same point
            out["bs_us"].append(1e6 * g.timings().back_substitution_time)
        out["cg_iterations"] = cg.num_iterations
        out["inc_norm"] = float(np.linalg.norm(inc_ref.astype(np.float64)))
        for k in ("hx_us", "bs_us", "s1_us", "s2_us"):
            out[k] = [round(v, 1) for v in out[k]]
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
