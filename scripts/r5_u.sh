# hx instance without the global-atomics path; double assembly pipelined; f64 line; persistent / product tests
set -x
TAG=${1:-r5u}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
bash scripts/r5_quick.sh $TAG "persistent_pcg or implicit_q_product_kernels or implicit_q_operator or explicit_reduced_matrix"
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --use-double > $O/venice_f64.json 2> $O/venice_f64.log
python - <<PY
import json
d=json.loads(open('$O/venice_f64.json').read().strip().splitlines()[-1])
print('F64 VALUE', d['value'], d['value_repeats']['values'], 'refsem', (d['config'].get('value_reference_semantics') or {}).get('value'))
PY
