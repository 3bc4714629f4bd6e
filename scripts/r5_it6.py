"""final-13682 lock-step, iteration 6 (the exempted state): the two forms of stage 1 (RBA_S1_FUSED=1 / 0) against the CPU
float64 referee - is the fused kernel the reason the GPU's 3-iteration increment moved from 6.1e-3 (round 4's file) to
1.41e-2 from float64?   usage (GPU box): python scripts/r5_it6.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import test_gpu_baseline_configs as T
    rows = T._fixture_rows("final-13682", tag="s1_" + sys.argv[1])
    for r in rows:
        print(json.dumps({k: r[k] for k in ("it", "cg_gpu", "cost_gpu", "inc_rel", "gpu_vs_f64", "oracle32_vs_f64", "l_diff_rel")}))
else:
    for fused in ("1", "0"):
        print("RBA_S1_FUSED=" + fused, flush=True)
        subprocess.run([sys.executable, __file__, fused], env=dict(os.environ, RBA_S1_FUSED=fused), check=False)
