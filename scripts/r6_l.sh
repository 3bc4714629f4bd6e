# round 6, call L: A/B on ONE box - library of the last commit / compact tile map / + half gather of x in the product
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6l
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head ot ot_xh; do
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion > $O/prof_$v.json 2> $O/prof_$v.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${v}_$rep.csv
  rm -rf $O/prof_$v
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_${v}_$rep.csv')))
want=['k_hx_implicit_lds','k_s1_fused_obs','k_cam_pass_mfma','k_bs_tile','k_s2_obs','k_compute_error']
out=[]
for r in rows:
    n=r['Name']
    for w in want:
        if w in n: out.append(f"{w} {float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/prof_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep:', ' | '.join(out), '| value', round(d['value'],1))
PY
done
done
