# rocprofv3 --kernel-trace --stats of the default bench -> gpurun_out/$1/
set -x
TAG=${1:-r2prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-iters 0 ${@:2} > $OUT/prof.json 2> $OUT/prof.log
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv | cut -c1-180
rm -rf $OUT/prof/*/*kernel_trace.csv
