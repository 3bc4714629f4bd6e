# round 6, call AF: the round-end rehearsal (GPU tier -x, smoke, default bench line) + config 5 and the power-series
# regime of venice with the last library (series steps on one workgroup per tile)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6af
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
B="python bench.py --cpu-baseline-iters 0 --no-pmc --no-dense-companion"
$B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power.json 2> $O/final_mixed_power.log
$B --steps 20 --warmup 5 --preconditioner POWER_SCHUR_COMPLEMENT --no-reference-semantics > $O/venice_power.json 2> $O/venice_power.log
grep "  it " $O/final_mixed_power.log | cut -c1-130
for f in venice final_mixed_power venice_power; do python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
c=d['config']
print('$f VALUE', round(d['value'],2), [round(v,1) for v in d['value_repeats']['values']], 'ms/step', round(d['ms_per_step'],3), 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'dense', (c.get('value_dense_covisibility') or {}).get('value'), 'frac', round(d['roofline']['frac'] or 0,3))
PY
done
