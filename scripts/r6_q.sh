# round 6, call Q: persistent PCG kernel without scratch - its parity tests, the venice lock-step counts, bench + kernel statistics
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6q
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_baseline_configs.py -m gpu -q -k "persistent or config4 or config3 or sharded or explicit_switch or trajectory" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-dense-companion > $O/venice.json 2> $O/venice.log
python - <<PY
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
c=d['config']
print('venice VALUE', round(d['value'],1), [round(v,1) for v in d['value_repeats']['values']], 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'pcg', d['roofline']['stages']['pcg']['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
for r in rows[:4]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us")
PY
grep " it " $O/venice.log | sed -n 6,12p
