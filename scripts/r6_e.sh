# round 6, call E: streaming SpMV (LDS-DMA) - parity, dense regime (venice-1778+tail), config 5 (final-13682 mixed + power series)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "streaming_spmv or persistent_pcg or explicit" > $O/pytest_spmv.log 2>&1; tail -5 $O/pytest_spmv.log
B="python bench.py --cpu-baseline-iters 0 --no-pmc --no-dense-companion"
for mode in 0 1; do
  RBA_SPMV_STREAM=$mode RBA_VERBOSE=1 $B --steps 20 --warmup 5 --workload venice-1778+tail --no-reference-semantics --repeats 1 > $O/tail_stream$mode.json 2> $O/tail_stream$mode.log
  grep "assembly" $O/tail_stream$mode.log | head -3
done
for w in 2 4; do
  RBA_SPMV_STREAM_WAVES=$w RBA_VERBOSE=1 $B --steps 20 --warmup 5 --workload venice-1778+tail --no-reference-semantics --repeats 1 > $O/tail_stream_w$w.json 2> $O/tail_stream_w$w.log
done
for mode in 0 1; do
  RBA_SPMV_STREAM=$mode $B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power_stream$mode.json 2> $O/final_mixed_power_stream$mode.log
done
python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-dense-companion > $O/venice.json 2> $O/venice.log
for f in tail_stream0 tail_stream1 tail_stream_w2 tail_stream_w4 final_mixed_power_stream0 final_mixed_power_stream1 venice; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    c=d['config']
    print('$f VALUE', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'refsem', (c.get('value_reference_semantics') or {}).get('value'), 'ok', c['successful_steps'])
    print('  stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'] or 0,3)) for k,v in d['roofline']['stages'].items()})
    print('  executed', d['roofline']['stages']['pcg']['executed'])
except Exception as e:
    print('$f', repr(e))
PY
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
RBA_SPMV_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$mode -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload venice-1778+tail > $O/prof$mode.json 2> $O/prof$mode.log
find $O/prof$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tail_kernel_stats_stream$mode.csv
rm -rf $O/prof$mode
done
RBA_SPMV_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proff -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT > $O/proff.json 2> $O/proff.log
find $O/proff -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/final_mixed_power_kernel_stats.csv
rm -rf $O/proff
cd $GRAFT_REPO_ROOT
for f in tail_kernel_stats_stream0 tail_kernel_stats_stream1 final_mixed_power_kernel_stats; do python - <<PY
import csv
rows=list(csv.DictReader(open('$O/$f.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('$f total kernel ms', tot/1e6)
for r in rows[:14]:
    n=r['Name']; n=n[:n.index('(')] if '(' in n else n
    print(f"{n[:70]:70s} {r['Calls']:>6s} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
