#!/bin/bash
set -x
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stage2_operator_backsub" > gpurun_out/r3/pytest_gpu_s2.log 2>&1
tail -15 gpurun_out/r3/pytest_gpu_s2.log
bash scripts/r2_prof.sh r3/prof_kpass --steps 20 --warmup 5 --no-reference-semantics 2>&1 | tail -40
tail -3 gpurun_out/r3/prof_kpass/prof.log
