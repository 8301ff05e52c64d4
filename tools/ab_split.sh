#!/bin/bash
# A/B on a GPU box: split launches (first pass + follow-up kernel) against the monolithic kernel of the same library
#   tools/ab_split.sh <outdir-name>
name=${1:-absplit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
for mode in split mono; do
  if [ $mode = mono ]; then export MSD_MONOLITHIC=1; else unset MSD_MONOLITHIC; fi
  for w in "c1:" "c1_8192:--batch 8192" "c1ref:--start reference" "c3:--workload c3"; do
    tag=${w%%:*}; args=${w#*:}
    python bench.py --no-cpu-baseline --no-alt --no-build $args > $O/bench_${mode}_$tag.json 2>> $O/bench.err
    python -c "
import json
d=json.load(open('$O/bench_${mode}_$tag.json')); print('$mode $tag', '%.0f solves/s' % d['value'], '%.4f ms' % d['ms_per_step'], 'iters %.2f' % d['config'].get('ip_iterations_mean'), 'conv', d['config'].get('converged'))"
  done
done
