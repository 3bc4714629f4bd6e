# MFMA counters of the matrix-core kernels -> gpurun_out/<tag>/pmc_mfma_{f32,f64}.csv (counters in their own passes, --kernel-trace only)
set -x
TAG=${1:-pmcmfma}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for DT in float32 float64; do
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/m1_$DT -- python $GRAFT_REPO_ROOT/scripts/pmc_mfma.py run $DT 4 > $OUT/meta_$DT.txt 2> $OUT/m1_$DT.log
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/m2_$DT -- python $GRAFT_REPO_ROOT/scripts/pmc_mfma.py run $DT 4 > /dev/null 2> $OUT/m2_$DT.log
  python $GRAFT_REPO_ROOT/scripts/pmc_mfma.py parse $OUT/m1_$DT $OUT/m2_$DT $OUT/pmc_mfma_$DT.csv | cut -c1-250
  rm -rf $OUT/m1_$DT $OUT/m2_$DT
done
tail -3 $OUT/m1_float32.log
