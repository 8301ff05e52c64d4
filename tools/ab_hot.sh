#!/bin/bash
# A/B of tools/build_hot.py variants on a GPU box: tools/ab_hot.sh <outdir-name> <tag> [<tag> ...]  ("product" = the product library)
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
for tag in "$@"; do
  if [ "$tag" = product ]; then unset MSD_LIB; else export MSD_LIB=$R/ms-eetc_amd/lib/variants/libmseetc_hip_$tag.so; fi
  for w in ${WORKLOADS:-"c1:" "c1_8192:--batch_8192_--steps_60"}; do
    t=${w%%:*}; args=${w#*:}; args=${args//_/ }
    python3 bench.py --no-cpu-baseline --no-alt --no-build $args > $O/bench_${tag}_$t.json 2>> $O/bench.err
    python3 -c "
import json
d=json.load(open('$O/bench_${tag}_$t.json')); print('$tag $t', '%.0f solves/s' % d['value'], '%.4f ms' % d['ms_per_step'], 'iters %.2f' % d['config'].get('ip_iterations_mean'), 'conv', d['config'].get('converged'))"
  done
done
