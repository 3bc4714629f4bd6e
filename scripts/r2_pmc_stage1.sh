# Round 2: SQ counters for the stage-1 kernels (k_linearize_qr_packed<1|2>, k_cam_*), two PMC passes,
# --kernel-trace only.  -> gpurun_out/r2pmc/
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-baseline-iters 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- $B > $OUT/p1.json 2> $OUT/p1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p2 -- $B > $OUT/p2.json 2> $OUT/p2.log
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/p3 -- $B > $OUT/p3.json 2> $OUT/p3.log
cd $GRAFT_REPO_ROOT
python bench.py --cpu-baseline-iters 0 > $OUT/bench.json 2> $OUT/bench.log
tail -c 600 $OUT/bench.json
ls -la $OUT $OUT/p1 | head -40
find $OUT -name "*.csv" | xargs du -sh | head
