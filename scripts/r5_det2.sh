# deterministic mode, second call: the LM repeat test after freezing the measured break-even; config-5 two-rank test
set -x
TAG=${1:-r5det2}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deterministic" > $O/pytest_det_parity.log 2>&1
tail -5 $O/pytest_det_parity.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "deterministic_mode" > $O/pytest_det_traf.log 2>&1
tail -5 $O/pytest_det_traf.log
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "two_ranks_split" > $O/pytest_cfg5.log 2>&1
tail -25 $O/pytest_cfg5.log
