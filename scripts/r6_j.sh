# round 6, call J: evidence set after the deletion of the round-1 PCG loop - full GPU tier, the bench lines of every regime
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6j
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
B="python bench.py --cpu-baseline-iters 0 --no-pmc --no-dense-companion"
$B --steps 20 --warmup 5 --workload trafalgar-257 > $O/trafalgar.json 2> $O/trafalgar.log
$B --steps 20 --warmup 5 --workload ladybug-49 > $O/ladybug.json 2> $O/ladybug.log
$B --steps 20 --warmup 5 --use-double > $O/venice_f64.json 2> $O/venice_f64.log
$B --steps 20 --warmup 5 --mixed > $O/venice_mixed.json 2> $O/venice_mixed.log
$B --steps 20 --warmup 5 --preconditioner POWER_SCHUR_COMPLEMENT > $O/venice_power.json 2> $O/venice_power.log
RBA_EXPLICIT_AFTER=0 $B --steps 10 --warmup 3 --preconditioner POWER_SCHUR_COMPLEMENT --workload trafalgar-257 --repeats 1 --no-reference-semantics > $O/trafalgar_power_matrix_free.json 2> $O/trafalgar_power_matrix_free.log
$B --steps 20 --warmup 5 --solver-type SCHUR_COMPLEMENT > $O/venice_sc.json 2> $O/venice_sc.log
$B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power.json 2> $O/final_mixed_power.log
$B --steps 10 --warmup 3 --workload final-13682 --repeats 1 --no-reference-semantics > $O/final.json 2> $O/final.log
RBA_DETERMINISTIC=1 $B --steps 20 --warmup 5 --no-reference-semantics --repeats 1 > $O/venice_det.json 2> $O/venice_det.log
for f in venice trafalgar ladybug venice_f64 venice_mixed venice_power trafalgar_power_matrix_free venice_sc final_mixed_power final venice_det; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    c=d['config']
    print('$f VALUE', round(d['value'],2), [round(v,1) for v in d['value_repeats']['values']], 'ms/step', round(d['ms_per_step'],4), 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'dense', (c.get('value_dense_covisibility') or {}).get('value'), 'ok', c['successful_steps'])
    print('  stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'] or 0,3)) for k,v in d['roofline']['stages'].items()}, 'roof', round(d['roofline']['frac'] or 0,3), d['roofline']['whole_iteration']['frac_without_pcg'])
    print('  executed', d['roofline']['stages']['pcg']['executed'])
    if 'cpu_baseline' in d: print('  cpu', d['cpu_baseline'])
except Exception as e:
    print('$f', repr(e))
PY
done
