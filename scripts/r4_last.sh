# the default bench line (in-run PMC traffic, CPU baseline, reference-semantics run) and its kernel statistics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4L; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/venice.json 2> $O/venice.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --no-reference-semantics --repeats 1 --no-pmc > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.log)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
python -c "
import json
d=json.loads(open('$O/venice.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], d['value_repeats']['values'], 'frac', r['frac'], 'launch', r['avg_launch_ms'], 'traffic', r['traffic'])"
head -4 $O/kernel_stats.csv | cut -c1-160
