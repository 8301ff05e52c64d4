#!/bin/bash
# round 5, GPU session B: complete GPU test suite, sweeps of every family, bench line, horizon timing -> gpurun_out/<name>/
name=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -n 25
timeout 900 python tests/tools/random_sweep.py 0 ${SWEEP_LAST:-200} > $O/sweep.txt 2>&1; tail -n 4 $O/sweep.txt
SWEEP_FACTORS=2.5,3 timeout 600 python tests/tools/random_sweep.py 0 ${LOOSE_LAST:-100} > $O/sweep_loose.txt 2>&1; tail -n 5 $O/sweep_loose.txt
timeout 900 python tests/tools/random_sweep_transcriptions.py 0 ${TR_LAST:-20} > $O/sweep_tr.txt 2>&1; tail -n 3 $O/sweep_tr.txt
timeout 900 python tests/tools/random_sweep_loss_functions.py 0 ${LF_LAST:-30} > $O/sweep_lf.txt 2>&1; tail -n 3 $O/sweep_lf.txt
python bench.py --no-build > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print('c1', '%.0f solves/s' % d['value'], '%.4f ms' % d['ms_per_step'], 'iters %.2f' % d['config']['ip_iterations_mean'], 'conv', d['config']['converged'])
for k,v in d.get('alt',{}).items():
    if isinstance(v,dict) and 'solves_per_s' in v: print(k, '%.0f' % v['solves_per_s'], v.get('running_times_not_converging',''))
    if k=='c4': print(k, {a:(round(b['resolves_per_s']) if isinstance(b,dict) else '') for a,b in v.items()})
    if k=='host_buffers': print(k, {a:(round(b['solves_per_s']) if isinstance(b,dict) else '') for a,b in v.items()})
"
timeout 900 python tools/horizon_timing.py 1024 > $O/horizon.txt 2>&1; tail -n 22 $O/horizon.txt
