# final-13682 lock-step against the CPU float64 referee
set -x
TAG=${1:-r5x}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "config5_final13682_f32_lockstep" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
cat gpurun_out/fixture_lockstep_final-13682_float32_default.jsonl gpurun_out/fixture_lockstep_final-13682_float32_matrix_free.jsonl
