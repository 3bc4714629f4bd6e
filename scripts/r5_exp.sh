# persistent-PCG experiments: per-iteration time from the kernel's own stamps for values of an environment knob
# usage: bash scripts/r5_exp.sh TAG VAR v1 v2 ...
TAG=$1; VAR=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -f $O/trace_$v.txt
  env $VAR=$v RBA_PCGP_TRACE=$O/trace_$v.txt python bench.py --steps 7 --warmup 2 --cpu-baseline-iters 0 --no-pmc --no-reference-semantics --repeats 1 > $O/b_$v.json 2> $O/b_$v.log
  echo "== $VAR=$v"
  python scripts/pcgp_trace.py $O/trace_$v.txt | tail -13
  python -c "
import json
d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print('it/s', d['value'])"
done
