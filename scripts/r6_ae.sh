# round 6, call AE: series steps on a grid of one workgroup per tile of cameras (all but the last term) - parity + config-5
# lock-steps, then A/B on ONE box against the last commit: config 5, first LM iterations (PCG counts 2 / 2 / 11 / 3)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6ae
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "series or power" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
for v in head grid; do
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT > $O/prof_${v}_$rep.json 2> $O/prof_${v}_$rep.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${v}_$rep.csv
  rm -rf $O/prof_$v
  grep "  it  [234] " $O/prof_${v}_$rep.log | cut -c1-130
  python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_${v}_$rep.csv')))
out=[]
for r in rows:
    if 'k_pcgs_' in r['Name']: out.append(f"{r['Name'][10:34]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
print('$v rep $rep:', ' | '.join(sorted(out)))
PY
done
done
cp variants/lib_grid.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
python bench.py --cpu-baseline-iters 0 --no-pmc --no-dense-companion --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power.json 2> $O/final_mixed_power.log
grep "  it " $O/final_mixed_power.log | cut -c1-130
