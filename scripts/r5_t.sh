# stream microbenchmark + quick bench / kernel stats after the unpredicated-operand changes
set -x
TAG=${1:-r5t}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT/scripts/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_read.bin stream_read.hip && timeout 120 /tmp/stream_read.bin > $O/stream_read.txt 2>&1
cat $O/stream_read.txt
cd $GRAFT_REPO_ROOT
bash scripts/r5_quick.sh $TAG "fallback_is_collective or explicit_reduced_matrix_is_the_same_operator or linearization_stage2_operator_backsub"
