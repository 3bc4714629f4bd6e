# round 6, call W: dense regime, A/B on ONE box: transposed-product slots of 72 bytes (last commit) against whole 128-byte lines
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6w
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head slot128; do
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload venice-1778+tail > $O/prof_$v.json 2> $O/prof_$v.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${v}_$rep.csv
  rm -rf $O/prof_$v
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_${v}_$rep.csv')))
out=[]
for r in rows:
    if 'k_pcgs_spmv' in r['Name'] or 'k_pcgs_update' in r['Name'] or 'k_pcgs_reduce' in r['Name']: out.append(f"{r['Name'][10:44]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/prof_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep:', ' | '.join(sorted(out)), '| value', round(d['value'],1), 'its', d['roofline']['stages']['pcg']['executed']['iterations'])
PY
done
done
