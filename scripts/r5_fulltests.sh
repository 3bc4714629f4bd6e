# final validation: the whole GPU tier + smoke + the default bench line
set -x
TAG=${1:-r5full}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -c 600 $O/bench_default.json
