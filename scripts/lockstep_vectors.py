"""Vector-level lock-step of the HIP library against the CPU oracle at BASELINE sizes: prints what
tests/test_gpu_baseline_configs.py asserts (tests/lockstep.py), one JSON line per LM iteration.
usage: python scripts/lockstep_vectors.py <workload> <float32|float64|mixed> [iterations] [preconditioner_type]"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from lockstep import lockstep_rows

if __name__ == "__main__":
    name, dts = sys.argv[1], sys.argv[2]
    n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    precond = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
    prob = bench.make_problem(name, args)[0]
    for row in lockstep_rows(prob, dts, n_it, precond):
        print(json.dumps(row), flush=True)
