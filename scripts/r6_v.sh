# round 6, call V: dense regime (venice-1778+tail), A/B on ONE box: the transposed-product slots of the half-storage SpMV
# written with plain / non-temporal stores; and the product's FETCH_SIZE / WRITE_SIZE (does the 72-byte slot scatter cost
# read-for-ownership traffic?)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6v
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head nt; do
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
    if 'k_pcgs_' in r['Name']: out.append(f"{r['Name'][10:44]} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
print('$v rep $rep:', ' | '.join(sorted(out)))
PY
done
done
cp variants/lib_head.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion --workload venice-1778+tail > /dev/null 2> $O/pmc_$c.log
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
for c in ('FETCH_SIZE','WRITE_SIZE'):
    acc={}
    for f in glob.glob('$O/pmc_'+c+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c and 'k_pcgs' in r['Kernel_Name']:
                k=r['Kernel_Name'].split('(')[0][10:50]
                a=acc.setdefault(k,[0.0,0]); a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,n) in sorted(acc.items()): print(c, k, 'dispatches', n, 'mean KiB', round(v/n,1), '-> MB (fetch x2 calibration)', round(v/n*1024*(2 if c=='FETCH_SIZE' else 1)/1e6,1))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
