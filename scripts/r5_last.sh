# last numbers of the round: the drop-in path and the deterministic mode with the final library
set -x
TAG=${1:-r5last}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/dropin_bench.py venice-1778 20 > $O/dropin.json 2> $O/dropin.log; tail -c 700 $O/dropin.json
B="python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-reference-semantics"
RBA_DETERMINISTIC=1 $B > $O/venice_det.json 2> $O/venice_det.log
RBA_DETERMINISTIC=1 $B > $O/venice_det2.json 2> $O/venice_det2.log
diff <(grep "^  it" $O/venice_det.log | sed 's/ t .*//') <(grep "^  it" $O/venice_det2.log | sed 's/ t .*//') && echo "DETERMINISTIC RUNS IDENTICAL"
for f in $O/venice_det*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), d['value_repeats']['values'], round(d['roofline']['frac'] or 0,3))"; done
