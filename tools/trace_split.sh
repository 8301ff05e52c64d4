#!/bin/bash
# per-kernel durations of split and monolithic launches (rocprofv3 --kernel-trace --stats): tools/trace_split.sh <outdir-name>
name=${1:-trsplit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in split mono; do
  if [ $mode = mono ]; then export MSD_MONOLITHIC=1; else unset MSD_MONOLITHIC; fi
  for w in "c1:" "c1_8192:--batch 8192" "c3:--workload c3"; do
    tag=${w%%:*}; args=${w#*:}
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_${mode}_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-build --steps 10 --warmup 2 $args > $O/tr_${mode}_$tag.json 2> $O/tr_${mode}_$tag.err
    echo "== $mode $tag"; cat $(find $O/tr_${mode}_$tag -name '*kernel_stats.csv' | head -1) | cut -c1-200
  done
done
