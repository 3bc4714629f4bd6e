# round 6, call O: the trial point's cost evaluated inside the back-substitution (RBA_COST_IN_BS=1, default) against the
# pass of its own (=0) on ONE box; parity tier first
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_sharded.py tests/test_reference_gpu.py tests/test_reference_golden.py tests/test_reference_loop_on_hip.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
for mode in 0 1; do
  cd /tmp && export TMPDIR=/tmp
  RBA_COST_IN_BS=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --repeats 3 --no-pmc --no-dense-companion > $O/bench_${mode}_$rep.json 2> $O/bench_${mode}_$rep.log
  cd $GRAFT_REPO_ROOT
  find $O/prof_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${mode}_$rep.csv
  rm -rf $O/prof_$mode
  python - <<PY
import csv,json
rows=list(csv.DictReader(open('$O/kernel_stats_${mode}_$rep.csv')))
want=['k_bs_tile','k_compute_error','k_reduce_rows<8>','k_update_cameras','k_hx_implicit_lds','k_s1_fused_obs']
out=[]
for r in rows:
    for w in want:
        if w in r['Name']: out.append(f"{w} {r['Calls']}x{float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open('$O/bench_${mode}_$rep.json').read().strip().splitlines()[-1])
st=d['roofline']['stages']
print('COST_IN_BS=$mode rep $rep:', ' | '.join(out), '| value', round(d['value'],1), [round(v,1) for v in d['value_repeats']['values']], 'refsem', round((d['config'].get('value_reference_semantics') or {}).get('value') or 0,1), 'bs', round(st['back_substitution']['ms'],3), st['back_substitution']['frac'], 'ce', round(st['compute_error']['ms'],3), st['compute_error']['frac'], 'final cost', d['config'].get('final_cost'))
PY
done
done
