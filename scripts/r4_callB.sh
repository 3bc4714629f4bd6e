# stage 1: observation-per-lane fused kernel (1, default) / two kernels (0) / row-per-lane fused kernel (2)
set -x
TAG=${1:-r4B}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "fused_stage1 or lm_trajectory or (explicit_reduced and float32) or venice" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc"
for F in 1 1; do
  RBA_S1_FUSED=$F $B > $O/venice_fused$F.json 2> $O/venice_fused$F.log
  python - <<PY
import json
d=json.loads(open('$O/venice_fused$F.json').read().strip().splitlines()[-1])
print('FUSED=$F VALUE', d['value'], d['value_repeats']['values'], 'ms/step', d['ms_per_step'])
print('stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'],3)) for k,v in d['roofline']['stages'].items()})
PY
done
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
for r in rows[:16]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
