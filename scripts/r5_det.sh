# deterministic mode (RBA_DETERMINISTIC=1) and the two-rank config-5 test: first GPU run of both
set -x
TAG=${1:-r5det}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deterministic" > $O/pytest_det_parity.log 2>&1
tail -5 $O/pytest_det_parity.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "deterministic_mode" > $O/pytest_det_traf.log 2>&1
tail -15 $O/pytest_det_traf.log
B="python bench.py --cpu-baseline-iters 0 --no-pmc --no-reference-semantics --repeats 1"
RBA_DETERMINISTIC=1 $B --steps 20 --warmup 5 > $O/venice_det.json 2> $O/venice_det.log
RBA_DETERMINISTIC=1 $B --steps 20 --warmup 5 > $O/venice_det2.json 2> $O/venice_det2.log
RBA_DETERMINISTIC=1 RBA_EXPLICIT_AFTER=0 $B --steps 20 --warmup 5 > $O/venice_det_mf.json 2> $O/venice_det_mf.log
diff <(grep "^  it" $O/venice_det.log | sed 's/ t .*//') <(grep "^  it" $O/venice_det2.log | sed 's/ t .*//') && echo "DETERMINISTIC RUNS IDENTICAL"
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), d['config'].get('successful_steps'), round(d['roofline']['frac'] or 0,3), d['roofline'].get('achieved'))"; done
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "two_ranks_split" > $O/pytest_cfg5.log 2>&1
tail -15 $O/pytest_cfg5.log
