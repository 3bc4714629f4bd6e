#!/bin/bash
# full GPU acceptance run + default bench (round 3 development loop)
set -x
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r3/pytest_gpu.log 2>&1
tail -15 gpurun_out/r3/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.log
tail -3 gpurun_out/r3/bench_default.log
python -c "import json,sys; d=json.loads(open('gpurun_out/r3/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['config'].get('value_reference_semantics'), d['roofline']['stages'])"
if [ -n "$RBA_WITH_PMC" ]; then bash scripts/run_pmc_stage_traffic.sh r3/pmc 8; fi
