# the GPU tests that run on the assembled double matrix of a float solver (after the last change of k_a64_offdiag)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r5sub
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py -m gpu -x -q -k "explicit or assembled or persistent or switch or power_series or lm_trajectory or deterministic_lm or mixed" > $O/a.log 2>&1; tail -3 $O/a.log
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "config3 or config4 or config5_mixed_precision_with_power or two_ranks_split" > $O/b.log 2>&1; tail -3 $O/b.log
timeout 120 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "split or single_process or fallback" > $O/c.log 2>&1; tail -3 $O/c.log
