# mid-round check: PMC of the assembly, double / trafalgar / ladybug lines with the persistent PCG
set -x
TAG=${1:-r5mid}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --cpu-baseline-iters 0 --no-pmc"
$B --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
$B --steps 20 --warmup 5 --use-double > $O/venice_f64.json 2> $O/venice_f64.log
$B --steps 20 --warmup 5 --workload trafalgar-257 > $O/trafalgar.json 2> $O/trafalgar.log
$B --steps 20 --warmup 5 --workload ladybug-49 > $O/ladybug.json 2> $O/ladybug.log
$B --steps 20 --warmup 5 --mixed > $O/venice_mixed.json 2> $O/venice_mixed.log
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), (d.get('value_repeats') or {}).get('values'), d['config'].get('successful_steps'), round(d['roofline']['frac'] or 0,3), (d['config'].get('value_reference_semantics') or {}).get('value'))"; done
bash scripts/run_pmc_a64.sh $TAG 2>&1 | tail -30
