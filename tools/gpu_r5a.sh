#!/bin/bash
# round 5, GPU session A: GPU test suite, the random sweeps around the round-4 findings, default bench line -> gpurun_out/<name>/
name=${1:-r5a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -n 12
timeout 900 python tests/tools/random_sweep.py ${SWEEP_FIRST:-150} ${SWEEP_LAST:-260} > $O/sweep.txt 2>&1; tail -n 4 $O/sweep.txt
SWEEP_FACTORS=2.5,3 timeout 600 python tests/tools/random_sweep.py 0 ${LOOSE_LAST:-60} > $O/sweep_loose.txt 2>&1; tail -n 5 $O/sweep_loose.txt
python bench.py --no-build > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print('c1', '%.0f solves/s' % d['value'], '%.4f ms' % d['ms_per_step'], 'iters %.2f' % d['config']['ip_iterations_mean'], 'conv', d['config']['converged'])
for k,v in d.get('alt',{}).items():
    if isinstance(v,dict) and 'solves_per_s' in v: print(k, '%.0f' % v['solves_per_s'])
"
