# SQ counters of the five streaming kernels of an LM iteration (and the cost evaluation): which unit is busy -> gpurun_out/<tag>/pmc_big_kernels.csv
# (counters in their own passes, --kernel-trace only)
set -x
TAG=${1:-pmcbig}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters_available.txt
P=1
for SET in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$P -- python $GRAFT_REPO_ROOT/scripts/pmc_mfma.py run float32 3 > $OUT/meta_$P.txt 2> $OUT/p$P.log || tail -5 $OUT/p$P.log
  P=$((P+1))
done
python $GRAFT_REPO_ROOT/scripts/pmc_kernels.py $OUT/pmc_big_kernels.csv k_hx_implicit_lds,k_s1_fused_obs,k_cam_pass_mfma,k_bs_tile,k_s2_obs,k_compute_error $OUT/p1 $OUT/p2 $OUT/p3 | cut -c1-400
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
