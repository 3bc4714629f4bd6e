# round 6, call AB: the bench lines of call M again with the LAST library (after the dense-regime changes; the GPU tier of
# this library: scripts/r6_p.sh) - the driver's bench line with its in-run
# PMC traffic and CPU baseline, kernel statistics of the same command, per-kernel traffic against the byte model (the table
# tests/test_byte_model.py holds the live model to), trafalgar / ladybug / dense regime / config 5
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
B="python bench.py --cpu-baseline-iters 0 --no-pmc --no-dense-companion"
$B --steps 20 --warmup 5 --workload trafalgar-257 > $O/trafalgar.json 2> $O/trafalgar.log
$B --steps 20 --warmup 5 --workload ladybug-49 > $O/ladybug.json 2> $O/ladybug.log
$B --steps 20 --warmup 5 --workload venice-1778+tail --no-reference-semantics > $O/venice_tail.json 2> $O/venice_tail.log
$B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power.json 2> $O/final_mixed_power.log
for f in venice trafalgar ladybug venice_tail final_mixed_power; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    c=d['config']
    print('$f VALUE', round(d['value'],2), [round(v,1) for v in d['value_repeats']['values']], 'steps', d['steps'], d['warmup'], 'ms/step', round(d['ms_per_step'],4), 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'dense', (c.get('value_dense_covisibility') or {}).get('value'), 'ok', c['successful_steps'])
    r=d['roofline']
    print('  roofline', r['bound'], 'achieved', round(r['achieved'] or 0,1), 'frac', round(r['frac'] or 0,3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'], 'avg_launch_ms', r['avg_launch_ms'])
    print('  stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'] or 0,3)) for k,v in r['stages'].items()}, r['whole_iteration']['frac_without_pcg'])
    if 'cpu_baseline' in d: print('  cpu', {k:d['cpu_baseline'][k] for k in ('value','cores','kind')})
except Exception as e:
    print('$f', repr(e))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:30]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
