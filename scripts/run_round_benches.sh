# Round bench set: every line that DESIGN.md / BASELINE.md / profiles/README.md quote. Run on the GPU box:
#   gpurun -- 'bash scripts/run_round_benches.sh'   -> gpurun_out/r1b/
set -x
mkdir -p gpurun_out/r1b
python bench.py > gpurun_out/r1b/venice.json 2> gpurun_out/r1b/venice.log
RBA_EXPLICIT_AFTER=0 python bench.py --cpu-baseline-iters 0 > gpurun_out/r1b/venice_matrix_free.json 2> gpurun_out/r1b/venice_matrix_free.log
python bench.py --dense-blocks --cpu-baseline-iters 0 > gpurun_out/r1b/venice_dense.json 2> gpurun_out/r1b/venice_dense.log
RBA_EXPLICIT_AFTER=0 python bench.py --dense-blocks --cpu-baseline-iters 0 > gpurun_out/r1b/venice_dense_matrix_free.json 2> gpurun_out/r1b/venice_dense_matrix_free.log
python bench.py --solver-type SCHUR_COMPLEMENT --cpu-baseline-iters 0 > gpurun_out/r1b/venice_sc.json 2> gpurun_out/r1b/venice_sc.log
python bench.py --workload trafalgar-257 > gpurun_out/r1b/trafalgar.json 2> gpurun_out/r1b/trafalgar.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r1b/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/r1b/prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r1b/prof.log
cd $GRAFT_REPO_ROOT
python bench.py --workload final-13682 --cpu-baseline-iters 0 > gpurun_out/r1b/final.json 2> gpurun_out/r1b/final.log
tail -c 400 gpurun_out/r1b/venice.json
python bench.py --use-double --cpu-baseline-iters 0 > gpurun_out/r1b/venice_f64.json 2> gpurun_out/r1b/venice_f64.log
python bench.py --use-double --solver-type SCHUR_COMPLEMENT --cpu-baseline-iters 0 > gpurun_out/r1b/venice_f64_sc.json 2> gpurun_out/r1b/venice_f64_sc.log
