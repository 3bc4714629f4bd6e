# round 6, call N: exact 36-byte operand loads of the streaming SpMV (AddressSanitizer finding), batched loads in k_finish_increment
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_reference_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-dense-companion > $O/venice.json 2> $O/venice.log
python - <<PY
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1])
c=d['config']
print('venice VALUE', round(d['value'],1), [round(v,1) for v in d['value_repeats']['values']], 'ms/step', round(d['ms_per_step'],4), 'refsem', (c.get('value_reference_semantics') or {}).get('value'))
print('  stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'] or 0,3)) for k,v in d['roofline']['stages'].items()}, 'roof', round(d['roofline']['frac'] or 0,3), d['roofline']['whole_iteration']['frac_without_pcg'])
PY
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
for r in rows[:20]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:58]:58s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
