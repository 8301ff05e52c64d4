#!/bin/bash
# Collect the judged measurements of a round on a GPU box: bench line, rocprofv3 kernel trace (headline and the alt workloads),
# HBM counters in their own passes (no tracing domain besides --kernel-trace), phase telemetry.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/<round>/
tag=${1:-run}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
python3 __graft_entry__.py > $out/build.log 2>&1            # un-profiled: the compiler never runs under the profiler's preload
python3 bench.py --no-build > $out/bench.json 2> $out/bench.err
python3 tools/phase_cycles.py > $out/phase_cycles.txt 2>&1
cd /tmp && export TMPDIR=/tmp
P="--no-build --no-cpu-baseline --no-alt"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c1 -o c1 -- python3 $repo/bench.py $P > $out/trace_c1.json 2> $out/trace_c1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c1ref -o c1ref -- python3 $repo/bench.py $P --start reference --steps 5 > $out/trace_c1ref.json 2> $out/trace_c1ref.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -o c2 -- python3 $repo/bench.py $P --workload c2 --steps 3 --warmup 1 > $out/trace_c2.json 2> $out/trace_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -o c3 -- python3 $repo/bench.py $P --workload c3 --steps 3 --warmup 1 > $out/trace_c3.json 2> $out/trace_c3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- python3 $repo/bench.py $P --steps 3 --warmup 1 > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- python3 $repo/bench.py $P --steps 3 --warmup 1 > $out/pmc_write.json 2> $out/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq -o sq -- python3 $repo/bench.py $P --steps 3 --warmup 1 > $out/pmc_sq.json 2> $out/pmc_sq.err
cd $repo
python3 tools/make_traffic_json.py $out > $out/hbm_traffic.json 2> $out/hbm_traffic.err
find $out -name "*.csv" | head -40
