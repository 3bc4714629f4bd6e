# L2 hit rate and memory-side traffic of the fused PCG kernels -> gpurun_out/r2pmc_pcgs/
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2pmc_pcgs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 7 --warmup 2 --cpu-baseline-iters 0"
RBA_PCG_GRAPHS=0 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p1 -- $B > $OUT/p1.json 2> $OUT/p1.log
RBA_PCG_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p2 -- $B > $OUT/p2.json 2> $OUT/p2.log
RBA_PCG_GRAPHS=0 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p3 -- $B > $OUT/p3.json 2> $OUT/p3.log
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 --filter pcgs > $OUT/pcgs.csv
cat $OUT/pcgs.csv
tail -3 $OUT/p1.log
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
