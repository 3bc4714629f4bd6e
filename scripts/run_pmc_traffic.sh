# HBM traffic of one H*x (dense blocks and implicit-Q) from PMC counters -> gpurun_out/pmc/hx_traffic.json
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py run > $OUT/meta_raw.txt 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py run > /dev/null 2> $OUT/write.log
grep -E '^\{"calib_bytes"|^\{.*hx_bytes' $OUT/meta_raw.txt | tail -1 > $OUT/meta.json
cd $GRAFT_REPO_ROOT
python scripts/pmc_traffic.py parse $OUT/fetch $OUT/write $OUT/hx_traffic.json $OUT/meta.json | tail -30
# keep only the small per-kernel counter tables
find $OUT -name "*counter_collection.csv" | head
