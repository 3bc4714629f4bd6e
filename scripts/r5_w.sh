# device stamps instead of HIP events for the stage timers inside rba_lm_step
set -x
TAG=${1:-r5w}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-pmc"
$B > $O/a_venice.json 2> $O/a_venice.log
RBA_STAGE_TIMERS=2 $B --no-reference-semantics > $O/b_venice_events.json 2> $O/b_venice_events.log
$B --workload trafalgar-257 > $O/c_traf.json 2> $O/c_traf.log
$B --workload ladybug-49 > $O/d_ladybug.json 2> $O/d_ladybug.log
$B --use-double --no-reference-semantics > $O/e_f64.json 2> $O/e_f64.log
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value'],2), (d.get('value_repeats') or {}).get('values'), (d['config'].get('value_reference_semantics') or {}).get('value'))
print('   stages', {k:(round(v.get('ms',v.get('ms_per_step',0)),3), round(v['frac'],3)) for k,v in d['roofline']['stages'].items()})"; done
grep "^  it" $O/a_venice.log | tail -4
timeout 900 python -m pytest tests -m gpu -x -q -k "substage_timers or lm_trajectory or deterministic_lm or backup_restore or test_single_process_sharded_handle" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
