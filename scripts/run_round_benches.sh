# Round bench set: every line that DESIGN.md / BASELINE.md / profiles/README.md quote. Run on the GPU box:
#   gpurun -- 'bash scripts/run_round_benches.sh <tag>'   -> gpurun_out/<tag>/
set -x
TAG=${1:-r5}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --cpu-baseline-iters 0 --no-pmc"
python bench.py --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
RBA_EXPLICIT_AFTER=0 $B --steps 20 --warmup 5 --no-reference-semantics > $O/venice_matrix_free.json 2> $O/venice_matrix_free.log
$B --steps 20 --warmup 5 --mixed > $O/venice_mixed.json 2> $O/venice_mixed.log
$B --steps 20 --warmup 5 --use-double > $O/venice_f64.json 2> $O/venice_f64.log
$B --steps 20 --warmup 5 --preconditioner JACOBI --no-reference-semantics > $O/venice_jacobi.json 2> $O/venice_jacobi.log
$B --steps 20 --warmup 5 --preconditioner POWER_SCHUR_COMPLEMENT --no-reference-semantics > $O/venice_power.json 2> $O/venice_power.log
$B --steps 20 --warmup 5 --solver-type SCHUR_COMPLEMENT --no-reference-semantics > $O/venice_sc.json 2> $O/venice_sc.log
$B --steps 20 --warmup 5 --workload venice-1778+tail --no-reference-semantics > $O/venice_tail.json 2> $O/venice_tail.log
$B --steps 20 --warmup 5 --workload trafalgar-257 > $O/trafalgar.json 2> $O/trafalgar.log
$B --steps 20 --warmup 5 --workload ladybug-49 > $O/ladybug.json 2> $O/ladybug.log
$B --steps 10 --warmup 3 --workload final-13682 --repeats 1 --no-reference-semantics > $O/final.json 2> $O/final.log
$B --steps 10 --warmup 3 --workload final-13682 --mixed --repeats 1 --no-reference-semantics > $O/final_mixed.json 2> $O/final_mixed.log
$B --steps 10 --warmup 3 --workload final-13682 --mixed --preconditioner POWER_SCHUR_COMPLEMENT --repeats 1 --no-reference-semantics > $O/final_mixed_power.json 2> $O/final_mixed_power.log
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), (d.get('value_repeats') or {}).get('values'), d['config'].get('successful_steps'), round(d['roofline']['frac'] or 0,3), (d['config'].get('value_reference_semantics') or {}).get('value'))"; done
# kernel statistics of the default line (rocprofv3 --kernel-trace --stats)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.log)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
