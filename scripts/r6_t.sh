# round 6, call T: phase trace of the persistent PCG kernel (its own stamps) after the spills were removed
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6t
mkdir -p $O
cd $GRAFT_REPO_ROOT
rm -f $O/trace.txt
RBA_PCGP_TRACE=$O/trace.txt python bench.py --steps 12 --warmup 0 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc --no-dense-companion > $O/bench.json 2> $O/bench.log
python scripts/pcgp_trace.py $O/trace.txt > $O/pcgp_phase_trace.txt 2>&1; head -40 $O/pcgp_phase_trace.txt
rm -f $O/trace.txt
