# quick GPU check: default bench line + kernel stats (+ optional pytest selection in $2)
set -x
TAG=${1:-r5q}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$2" > $O/pytest.log 2>&1; tail -15 $O/pytest.log; fi
RBA_VERBOSE=1 python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc > $O/venice.json 2> $O/venice.log
python - <<PY
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['value_repeats']['values'], 'ms/step', d['ms_per_step'], 'refsem', (d['config'].get('value_reference_semantics') or {}).get('value'))
print('stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'],3)) for k,v in d['roofline']['stages'].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:22]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
