set -x
TAG=${1:-r5y}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_timers_inside or substage_timers" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc > $O/venice.json 2> $O/venice.log
python -c "
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['value_repeats']['values'], (d['config'].get('value_reference_semantics') or {}).get('value'), d['roofline']['frac'])"
