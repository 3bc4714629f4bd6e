# trafalgar-257 float32 default configuration in lock-step with the oracle, per stage-1 form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4E
for F in 1 0 1 0; do
RBA_S1_FUSED=$F python - <<PY 2>&1 | grep -v Gloo
import sys, json
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from lockstep import lockstep_rows
from test_gpu_baseline_configs import _bench_problem
rows = list(lockstep_rows(_bench_problem("trafalgar-257"), "float32", 6, 1))
for r in rows[1:]:
    print("FUSED=$F", {k: (round(v, 8) if isinstance(v, float) else v) for k, v in r.items() if k in ("cg_gpu", "cg_oracle", "inc_rel", "gpu_vs_f64", "oracle32_vs_f64", "termination")})
PY
done
