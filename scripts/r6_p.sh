# round 6, call P: what the driver runs at round end, on the final tree - the GPU tier with -x, smoke(), the bench line
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
python - <<PY
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
c=d['config']; r=d['roofline']
print('venice VALUE', round(d['value'],1), [round(v,1) for v in d['value_repeats']['values']], 'ms/step', round(d['ms_per_step'],4), 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'dense', (c.get('value_dense_covisibility') or {}).get('value'))
print('  roofline', r['bound'], round(r['achieved'],1), r['peak'], round(r['frac'],3), r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
print('  keys', sorted(d.keys()))
PY
