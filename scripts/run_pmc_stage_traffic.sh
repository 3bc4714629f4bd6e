# HBM traffic of every kernel of an LM run vs the byte model -> gpurun_out/<tag>/pmc_stage_traffic.{csv,json}, hx_traffic.json
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -x
TAG=${1:-pmc}
ITERS=${2:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/scripts/pmc_stage_traffic.py run $ITERS > $OUT/meta_raw.txt 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/scripts/pmc_stage_traffic.py run $ITERS > /dev/null 2> $OUT/write.log
grep '^PMC_META ' $OUT/meta_raw.txt | tail -1 | sed 's/^PMC_META //' > $OUT/meta.json
cd $GRAFT_REPO_ROOT
python scripts/pmc_stage_traffic.py parse $OUT/fetch $OUT/write $OUT/meta.json $OUT/pmc_stage_traffic | tail -12
rm -rf $OUT/fetch $OUT/write
