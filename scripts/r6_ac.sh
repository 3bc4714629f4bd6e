# round 6, call AC: config 5 (final-13682 mixed + power series) - slots a row may receive before a wavefront of its own sums
# them (RBA_HALF_LOWER_MAX, default 192): the gather inside k_pcgs_series_step is a serial loop of 16-load batches
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6ac
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in 192 96 48 24; do
  cd /tmp && export TMPDIR=/tmp
  RBA_HALF_LOWER_MAX=$m timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT > $O/prof_${m}_$rep.json 2> $O/prof_${m}_$rep.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_m${m}_$rep.csv
  rm -rf $O/prof_$m
  grep "  it  [2345] " $O/prof_${m}_$rep.log | cut -c1-130
  python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_m${m}_$rep.csv')))
out=[]
for r in rows:
    if 'k_pcgs_' in r['Name']: out.append(f"{r['Name'][10:34]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
print('max $m rep $rep:', ' | '.join(sorted(out)))
PY
done
done
