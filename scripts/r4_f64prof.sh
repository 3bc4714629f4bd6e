set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r4f64
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --use-double > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_f64.csv
rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_f64.csv')))
print('total kernel ms', sum(int(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:16]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
grep "  it " $O/prof.log | head -12
