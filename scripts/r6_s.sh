# round 6, call S: A/B on ONE box - LDS records of the fused stage 1 at a 16-scalar stride (last commit) / 20 scalars
# (conflict-free, 66 registers) / 20 scalars at 64 registers (8 wavefronts per SIMD forced, 12 bytes of scratch)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6s
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head pad20 pad20_w8; do
  cp variants/lib_$v.so rootba_amd/librootba_hip.so; touch rootba_amd/librootba_hip.so rootba_amd/bal_qr_hip
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion > $O/prof_$v.json 2> $O/prof_$v.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${v}_$rep.csv
  rm -rf $O/prof_$v
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_${v}_$rep.csv')))
want=['k_s1_fused_obs','k_hx_implicit_lds','k_bs_tile','k_s2_obs','k_cam_pass_mfma']
out=[]
for r in rows:
    for w in want:
        if w in r['Name']: out.append(f"{w} {float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/prof_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep:', ' | '.join(sorted(out)), '| value', round(d['value'],1))
PY
done
done
