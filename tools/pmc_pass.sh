#!/bin/bash
# One rocprofv3 counter pass over bench.py (no tracing domains besides --kernel-trace): tools/pmc_pass.sh <outdir-name> <tag> <counter> [counter...]
# The library must already be built (run `python3 __graft_entry__.py` un-profiled first; gpurun ships the prebuilt .so).
set -u
name=$1; tag=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o $tag -- python3 $R/bench.py --no-cpu-baseline --no-build --steps 3 --warmup 1 ${BENCH_ARGS:-} > $O/pmc_$tag.json 2> $O/pmc_$tag.err
