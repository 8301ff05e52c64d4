#!/bin/bash
# rocprofv3 kernel statistics of the config-4 loop (three cold loops) -> gpurun_out/<name>/c4_kernel_stats.csv
name=${1:-c4prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 $R/bench.py --no-build --no-cpu-baseline --no-alt --workload c4 --steps 3 --warmup 1 > $O/c4.json 2> $O/c4.err
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/c4_kernel_stats.csv; head -12 $O/c4_kernel_stats.csv | cut -c1-200
python3 -c "
import json; d=json.load(open('$O/c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
