mkdir -p gpurun_out/r3
RBA_VERBOSE=1 python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 > gpurun_out/r3/bench_verify.json 2> gpurun_out/r3/bench_verify.log
grep "assembled solve" gpurun_out/r3/bench_verify.log | head -20
python -c "import json; d=json.loads(open('gpurun_out/r3/bench_verify.json').read().strip().splitlines()[-1]); print('venice', d['value'], d['roofline']['stages']['pcg']['executed'])"
RBA_VERBOSE=1 python bench.py --workload trafalgar-257 --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 2>&1 >/dev/null | grep "assembled solve" | head -12
RBA_VERBOSE=1 python bench.py --workload final-13682 --steps 10 --warmup 3 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 > gpurun_out/r3/bench_verify_final.json 2> gpurun_out/r3/bench_verify_final.log
grep "assembled solve\| it " gpurun_out/r3/bench_verify_final.log | tail -30
python -c "import json; d=json.loads(open('gpurun_out/r3/bench_verify_final.json').read().strip().splitlines()[-1]); print('final', d['value'], d['config']['successful_steps'], d['roofline']['stages']['pcg']['executed'])"
