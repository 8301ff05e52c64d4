#!/bin/bash
# On a GPU box (through gpurun): the GPU test suite, then the headline numbers of the built library.
#   tools/gpu_check.sh <outdir-name> [pmc]      -> gpurun_out/<outdir-name>/
name=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for w in "c1:" "c1_8192:--batch 8192" "c2:--workload c2" "c3:--workload c3"; do
  tag=${w%%:*}; args=${w#*:}
  python bench.py --no-cpu-baseline --no-alt --no-build $args > $O/bench_$tag.json 2>> $O/bench.err
  python -c "
import json
d=json.load(open('$O/bench_$tag.json')); print('$tag', '%.0f solves/s' % d['value'], '%.3f ms' % d['ms_per_step'], 'iters %.2f' % d['config'].get('ip_iterations_mean'), 'conv', d['config'].get('converged'))"
done
if [ "$2" = pmc ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-build --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-build --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-build --steps 3 --warmup 1 > /dev/null 2>&1
  cd $R
  python3 - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if 'solve_kernel' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items(): print(k, '%.4g' % (sum(v)/len(v)), 'launches', len(v))
PY
fi
