#!/bin/bash
# round 3, GPU call 1: the three default-off candidates of round 2 measured against the defaults
set -x
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
RBA_TEST_CANDIDATES=1 timeout 300 python -m pytest tests/test_zz_candidates_gpu.py -q -m gpu > gpurun_out/r3/pytest_candidates.log 2>&1
tail -8 gpurun_out/r3/pytest_candidates.log
timeout 300 python scripts/s2_ab.py venice-1778 > gpurun_out/r3/s2_ab_venice.jsonl 2> gpurun_out/r3/s2_ab_venice.err
cat gpurun_out/r3/s2_ab_venice.jsonl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/cam_block_pass.hip -o /tmp/cam_block_pass.bin \
  && timeout 120 /tmp/cam_block_pass.bin > gpurun_out/r3/cam_block_pass.txt 2>&1
cat gpurun_out/r3/cam_block_pass.txt
for env in "" "RBA_CAM_BLOCKS=1" "RBA_S2_FUSED_LM=1"; do
  tag=$(echo "${env:-default}" | tr ' =' '__')
  env $env timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics \
    > gpurun_out/r3/bench_${tag}.json 2> gpurun_out/r3/bench_${tag}.log
  python -c "import json,sys; d=json.loads(open('gpurun_out/r3/bench_${tag}.json').read().strip().splitlines()[-1]); print('${tag}', d['value'], d['roofline']['stages']['stage2'])"
done
for env in "" "RBA_HX_THREADS=512"; do
  tag=f64_$(echo "${env:-default}" | tr ' =' '__')
  env $env timeout 300 python bench.py --use-double --steps 10 --warmup 3 --cpu-baseline-iters 0 --no-reference-semantics \
    > gpurun_out/r3/bench_${tag}.json 2> gpurun_out/r3/bench_${tag}.log
  python -c "import json,sys; d=json.loads(open('gpurun_out/r3/bench_${tag}.json').read().strip().splitlines()[-1]); print('${tag}', d['value'], d['roofline'].get('achieved'), d['roofline'].get('avg_launch_ms'))"
done
