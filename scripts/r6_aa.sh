# round 6, call AA: slots of the transposed product stored whole by neighbouring lanes (store_slots) - parity file, then A/B
# on ONE box: last commit (head) / timing-only variant without slot stores / staged stores with two buffers / staged with
# one buffer and seven wavefronts per unit / staged, item kernel; venice-1778+tail, then config 5
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6aa
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
run() {  # variant tag env...
  v=$1; tag=$2; shift 2
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion $WL > $O/prof_$tag.json 2> $O/prof_$tag.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${tag}.csv
  rm -rf $O/prof_$tag
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_${tag}.csv')))
out=[]
for r in rows:
    if 'k_pcgs_spmv' in r['Name'] or 'k_pcgs_reduce' in r['Name']: out.append(f"{r['Name'][10:48]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/prof_$tag.json').read().strip().splitlines()[-1])
print('$tag:', ' | '.join(sorted(out)), '| value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],2))
PY
}
STEPS=8
for rep in 1 2; do
  WL="--workload venice-1778+tail"
  run head tail_head_$rep RBA_X=0
  run noslots tail_noslots_$rep RBA_X=0
  run ss tail_ss_b2_$rep RBA_SPMV_STREAM_BUFFERS=2
  run ss tail_ss_b1_$rep RBA_SPMV_STREAM_BUFFERS=1
  run ss tail_ss_item_$rep RBA_SPMV_STREAM=0
  run head tail_head_item_$rep RBA_SPMV_STREAM=0
done
STEPS=6
WL="--workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT"
run head final_head RBA_X=0
run ss final_ss_b1 RBA_SPMV_STREAM_BUFFERS=1
run ss final_ss_b2 RBA_SPMV_STREAM_BUFFERS=2
cp variants/lib_ss.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
