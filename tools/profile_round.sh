#!/bin/bash
# Collect the judged measurements of a round on a GPU box: bench line, rocprofv3 kernel trace, HBM counters (separate passes).
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>/
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
[ -z "$GRAFT_REPO_ROOT" ] && out=$(pwd)/gpurun_out/$tag
mkdir -p $out
repo=$(pwd)
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $repo/bench.py --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- python3 $repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc_fetch_bench.json 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- python3 $repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc_write_bench.json 2> $out/pmc_write.err
cd $repo
find $out -name "*.csv" | head -20
