#!/bin/bash
# A/B of tuning builds on a GPU box: tools/ab.sh <outdir-name> <tag> [<tag> ...]   (tags of tools/build_variant.py; env BENCH_ARGS, GEOMS)
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
for tag in "$@"; do
  for g in ${GEOMS:-default}; do
    export MSD_LIB=$R/ms-eetc_amd/lib/variants/libmseetc_hip_$tag.so
    if [ "$g" = default ]; then unset MSD_GEOMETRY; else export MSD_GEOMETRY=$g; fi
    python3 $R/tools/phase_cycles.py > $O/phases_${tag}_$g.txt 2>&1
    python3 $R/bench.py --no-cpu-baseline --no-build ${BENCH_ARGS:-} > $O/bench_${tag}_$g.json 2> $O/bench_${tag}_$g.err
    echo "== $tag $g: $(python3 -c "import json,sys; d=json.load(open('$O/bench_${tag}_$g.json')); print('%.0f solves/s  launch %.3f ms  iters %.2f conv %d' % (d['value'], d['roofline']['launch_ms'], d['config']['ip_iterations_mean'], d['config']['converged']))" 2>&1 | tail -n 1)"
    grep -E "kernel_ms|RICCATI|KKT |ASSEMBLE|UPDATE|MERIT|EVAL|GPHID|OTHER|READBACK" $O/phases_${tag}_$g.txt | awk '{printf "%s ", $0} END {print ""}' | sed 's/  */ /g' | cut -c1-600
  done
done
