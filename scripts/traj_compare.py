"""Print the GPU and oracle LM logs side by side (development aid).  python scripts/traj_compare.py trafalgar-257 12"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from oracle import oracle as O
from rootba_amd import _lib as L
from rootba_amd.linearizor import LinearizorHIP

name, n = sys.argv[1], int(sys.argv[2])
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
prob = bench.make_problem(name, args)[0]
kw = dict(robust_norm=1, huber_parameter=1.0, max_num_iterations=n, function_tolerance=0.0)
g = LinearizorHIP(prob, dt, L.default_options(**kw))
o = O.Oracle(prob, dt, O.default_options(**kw))
lg, _ = g.optimize_lm()
lo, _ = o.optimize_lm()
for a, b in zip(lg, lo):
    print(f"it {a.iteration:2d} ok {a.step_is_successful}/{b.step_is_successful} cg {a.cg_iterations:4d}/{b.cg_iterations:4d} "
          f"cost {a.cost:.9e}/{b.cost:.9e} rel {abs(a.cost-b.cost)/b.cost:.1e} inc {a.inc_norm:.4e}/{b.inc_norm:.4e} "
          f"lam {a.lambda_:.3e}/{b.lambda_:.3e}")
