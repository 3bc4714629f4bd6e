# final-13682 (BASELINE config 5's size), 12 LM iterations from the initial state: default configuration, every product
# matrix-free, mixed precision - against profiles/r3_final13682_oracle_f32_lm.log (the float32 oracle's own run)
set -x
TAG=${1:-r4final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --workload final-13682 --steps 12 --warmup 0 --repeats 1 --no-reference-semantics --cpu-baseline-iters 0"
RBA_VERBOSE=1 $B > $O/final_f32.json 2> $O/final_f32.log; grep "  it " $O/final_f32.log
RBA_EXPLICIT_AFTER=0 $B > $O/final_f32_matrix_free.json 2> $O/final_f32_matrix_free.log; grep "  it " $O/final_f32_matrix_free.log
RBA_VERBOSE=1 $B --mixed > $O/final_mixed.json 2> $O/final_mixed.log; grep "  it " $O/final_mixed.log
