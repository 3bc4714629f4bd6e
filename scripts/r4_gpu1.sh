# Round 4, first GPU call: the double assembled matrix on hardware - tests, lock-step, bench, kernel stats, counter names.
set -x
TAG=${1:-r4a}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_final13682.py > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scripts/lockstep_vectors.py trafalgar-257 float32 6 > $O/lockstep_trafalgar_f32.jsonl 2> $O/lockstep_trafalgar.err
RBA_VERBOSE=1 python bench.py --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log; tail -c 1500 $O/venice.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 > $O/prof.json 2> $O/prof.log
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -40 $O/kernel_stats.csv | cut -c1-200
rm -rf $O/prof
rocprofv3 -L > $O/counters.txt 2>&1; grep -i -c mfma $O/counters.txt
