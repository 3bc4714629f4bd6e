# round 6, call X: streaming product of the float series terms - two chunks in flight, 4 wavefronts per compute unit
# (RBA_SPMV_STREAM_BUFFERS=2) against one chunk in flight, 7 wavefronts per compute unit (=1); config 5 (final-13682
# mixed + power series) on ONE box, kernel durations from rocprofv3
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6x
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -k config5_final13682 > $O/pytest_final.log 2>&1
tail -3 $O/pytest_final.log
for rep in 1 2; do
for b in 2 1; do
  cd /tmp && export TMPDIR=/tmp
  RBA_SPMV_STREAM_BUFFERS=$b timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$b -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT > $O/prof_${b}_$rep.json 2> $O/prof_${b}_$rep.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_b${b}_$rep.csv
  rm -rf $O/prof_$b
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_b${b}_$rep.csv')))
out=[]
for r in rows:
    if 'k_pcgs_spmv' in r['Name'] or 'k_pcgs_series' in r['Name'] or 'reduce_slots' in r['Name']: out.append(f"{r['Name'][:44]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/prof_${b}_$rep.json').read().strip().splitlines()[-1])
print('buffers $b rep $rep:', ' | '.join(sorted(out)), '| value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1))
PY
done
done
