# matrix-free product: landmarks with 32 < k <= 64 inside the persistent kernel (default) against a kernel of their own
cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "implicit_q" 2>&1 | grep -E "passed|failed"
for W in 1 0 1 0; do
RBA_HX_WIDE_INSIDE=$W python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-reference-semantics 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('WIDE_INSIDE=$W value', round(d['value'],1), d['value_repeats']['values'], 'frac', round(r['frac'],4), 'launch ms', round(r['avg_launch_ms'],5))"
done
