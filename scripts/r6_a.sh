# round 6, call A: QR accuracy ensemble + baseline bench line with kernel statistics
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/qr_accuracy.py 12288 11 12 > $O/qr_accuracy.log 2>&1; tail -30 $O/qr_accuracy.log
cp gpurun_out/qr_accuracy.json $O/ 2>/dev/null
bash scripts/r5_quick.sh r6a "qr_accuracy"
