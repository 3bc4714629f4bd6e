# what the HIP events of the stage timers / product timing cost per LM iteration
set -x
TAG=${1:-r5v}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc --no-reference-semantics"
$B > $O/a_default.json 2> $O/a_default.log
RBA_STAGE_TIMERS=0 $B > $O/b_no_stage_timers.json 2> $O/b_no_stage_timers.log
RBA_STAGE_TIMERS=0 RBA_HX_TIMING_STRIDE=0 $B > $O/c_no_events.json 2> $O/c_no_events.log
RBA_HX_TIMING_STRIDE=0 $B > $O/d_no_hx_events.json 2> $O/d_no_hx_events.log
$B --workload trafalgar-257 > $O/e_traf_default.json 2> $O/e_traf_default.log
RBA_STAGE_TIMERS=0 RBA_HX_TIMING_STRIDE=0 $B --workload trafalgar-257 > $O/f_traf_no_events.json 2> $O/f_traf_no_events.log
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), (d.get('value_repeats') or {}).get('values'))"; done
