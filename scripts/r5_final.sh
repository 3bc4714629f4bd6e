# end-of-round evidence: bench set, kernel statistics, per-stage HBM traffic, phase trace of the persistent PCG kernel
set -x
TAG=${1:-r5}
O=gpurun_out/$TAG
mkdir -p $O
bash scripts/run_round_benches.sh $TAG 2>&1 | tail -20
RBA_PCGP_TRACE=$O/pcgp_trace_raw.txt python bench.py --steps 7 --warmup 2 --cpu-baseline-iters 0 --no-pmc --no-reference-semantics --repeats 1 > /dev/null 2> /dev/null
python scripts/pcgp_trace.py $O/pcgp_trace_raw.txt > $O/pcgp_phase_trace.txt; tail -14 $O/pcgp_phase_trace.txt; rm -f $O/pcgp_trace_raw.txt
bash scripts/run_pmc_stage_traffic.sh $TAG 2>&1 | tail -15
