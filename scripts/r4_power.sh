set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-baseline-iters 0 --no-reference-semantics --no-pmc"
$B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 > $O/final_mixed_power.json 2> $O/final_mixed_power.log; grep "  it " $O/final_mixed_power.log
$B --steps 10 --warmup 3 --workload final-13682 --mixed --repeats 1 > $O/final_mixed.json 2> $O/final_mixed.log; grep "  it " $O/final_mixed.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --preconditioner POWER_SCHUR_COMPLEMENT > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_power.csv
rm -rf $O/prof
python - <<PY
import csv, json
rows=list(csv.DictReader(open('$O/kernel_stats_power.csv')))
print('total kernel ms', sum(int(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:14]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
for f in ('final_mixed_power','final_mixed'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['config'].get('successful_steps'), d['config'].get('cg_iterations_per_step'))
PY
