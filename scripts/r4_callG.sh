# double assembly: a wavefront per off-diagonal block (default) against a workgroup per block (RBA_A64_WPB=4)
set -x
TAG=${1:-r4G}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -x -k "(explicit_reduced and float32) or (explicit_switch and float32)" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for W in 1 4; do
cd /tmp && export TMPDIR=/tmp
RBA_A64_WPB=$W rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$W -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc > $O/prof$W.json 2> $O/prof$W.log
cd $GRAFT_REPO_ROOT
find $O/prof$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats$W.csv
rm -rf $O/prof$W
python - <<PY
import csv, json
rows=list(csv.DictReader(open('$O/kernel_stats$W.csv')))
d=json.loads(open('$O/prof$W.json').read().strip().splitlines()[-1])
print('WPB=$W value', d['value'])
for r in rows:
    if 'a64' in r['Name']:
        n=r['Name']; n=n[:n.index('(')] if '(' in n else n
        print(f"  {n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
