# round 6, call D: double-precision camera-frame point - the final-13682 lock-step, parity tier, bench lines
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -k "config5_final13682" > $O/pytest_final.log 2>&1; tail -5 $O/pytest_final.log
cat gpurun_out/fixture_lockstep_final-13682_float32_matrix_free.jsonl gpurun_out/fixture_lockstep_final-13682_float32_default.jsonl > $O/final_rows.jsonl
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_baseline_configs.py::test_config5_final13682_f32_lockstep_iterations_3_to_7 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-dense-companion > $O/venice.json 2> $O/venice.log
python - <<PY
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
c=d['config']
print('VALUE', round(d['value'],1), [round(v,1) for v in d['value_repeats']['values']], 'refsem', (c.get('value_reference_semantics') or {}).get('value'))
print('  stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'] or 0,3)) for k,v in d['roofline']['stages'].items()})
PY
