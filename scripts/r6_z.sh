# round 6, call Z: what bounds the streaming product of the dense double matrix (venice-1778+tail)? TIMING-ONLY variants of
# the last commit on ONE box (their results are wrong on purpose): no stores of the transposed-product slots / operand
# gathers from eight cached columns / both (the product is then a plain read of the matrix + LDS + arithmetic)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6z
mkdir -p $O
cd $GRAFT_REPO_ROOT
cp rootba_amd/librootba_hip.so $O/../lib_keep.so
for rep in 1 2; do
for v in head noslots nogather readonly; do
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload venice-1778+tail > $O/prof_$v.json 2> $O/prof_$v.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${v}_$rep.csv
  rm -rf $O/prof_$v
  python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_${v}_$rep.csv')))
out=[]
for r in rows:
    if 'k_pcgs_spmv' in r['Name'] or 'k_pcgs_reduce' in r['Name'] or 'k_narrow' in r['Name']: out.append(f"{r['Name'][10:48]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f} (min {float(r['MinNs'])/1e3:.1f})")
print('$v rep $rep:', ' | '.join(sorted(out)))
PY
done
done
rm -f $O/../lib_keep.so
