"""Run-to-run spread of the float32 LM trajectory against the float32 / float64 oracles (development aid):
python scripts/traj_spread.py [workload] [runs] -- prints, per GPU run, the relative cost difference to the float32 oracle per
iteration and the double-evaluated final cost against the float64 optimum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from oracle import oracle as O
from rootba_amd import _lib as L
from rootba_amd import problem as P
from rootba_amd.linearizor import LinearizorHIP

name = sys.argv[1] if len(sys.argv) > 1 else "trafalgar-257"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prob = P.preprocess(P.named_synthetic(name), translation_sigma=0.5, point_sigma=0.5)
kw = dict(robust_norm=1, huber_parameter=1.0, max_num_iterations=12, function_tolerance=0.0)
o = O.Oracle(prob, np.float32, O.default_options(**kw))
lo, _ = o.optimize_lm()
o64 = O.Oracle(prob, np.float64, O.default_options(**kw))
l64, _ = o64.optimize_lm()
f64 = min(r.cost for r in l64 if r.step_is_successful)
ev = O.Oracle(prob, np.float64, O.default_options(**kw))
ev.set_state(*o.get_state())
print("oracle f32: cg", [r.cg_iterations for r in lo], "final (double eval) rel to f64 optimum", (ev.compute_error().all_error - f64) / f64)
print("oracle f32 vs f64 per-iteration cost rel", ["%.1e" % ((a.cost - b.cost) / b.cost) for a, b in zip(lo[1:], l64[1:])])
for env in ({"RBA_S2_COMPACT": "1"}, {"RBA_S2_COMPACT": "0"}):
    os.environ.update(env)
    for _ in range(runs):
        g = LinearizorHIP(prob, np.float32, L.default_options(**kw))
        lg, _ = g.optimize_lm()
        ev.set_state(*g.get_state())
        print(env, "cg", [r.cg_iterations for r in lg], "cost rel vs o32", ["%.1e" % ((a.cost - b.cost) / b.cost) for a, b in zip(lg[1:], lo[1:])],
              "final", "%.2e" % ((ev.compute_error().all_error - f64) / f64))
