#!/bin/bash
# Collect the judged measurements of a round on a GPU box: bench line, rocprofv3 kernel trace and HBM counters (own passes, no tracing
# domain besides --kernel-trace) for every workload and transcription, SQ counters of the headline kernel.
# usage: tools/profile_round.sh <tag> [quick]     -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/<round>/
#        quick: headline workload only
tag=${1:-run}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/$tag
mkdir -p $out
cd $repo
python3 __graft_entry__.py > $out/build.log 2>&1            # un-profiled: the compiler never runs under the profiler's preload
if [ $? -ne 0 ]; then echo "profile_round: the build failed (see $out/build.log): nothing is measured"; tail -5 $out/build.log; exit 1; fi
python3 bench.py --no-build > $out/bench.json 2> $out/bench.err || { echo "profile_round: bench.py failed"; tail -5 $out/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
P="--no-build --no-cpu-baseline --no-alt"
if [ "$2" = quick ]; then LIST="c1:"; else
LIST="c1: c1ref:--start_reference c1b8192:--batch_8192 c2:--workload_c2 c3:--workload_c3 intloss:--transcription_integrate_losses irk:--transcription_irk_radau2 cvodes:--transcription_cvodes_tolerances"
fi
for w in $LIST; do
  name=${w%%:*}; args=$(echo ${w#*:} | tr '_' ' ' | sed 's/integrate losses/integrate_losses/; s/irk radau2/irk_radau2/; s/cvodes tolerances/cvodes_tolerances/')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -o $name -- python3 $repo/bench.py $P $args --steps 10 --warmup 2 > $out/trace_$name.json 2> $out/trace_$name.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$name -o fetch -- python3 $repo/bench.py $P $args --steps 3 --warmup 1 > $out/pmc_fetch_$name.json 2> $out/pmc_fetch_$name.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write_$name -o write -- python3 $repo/bench.py $P $args --steps 3 --warmup 1 > $out/pmc_write_$name.json 2> $out/pmc_write_$name.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq_$name -o sq -- python3 $repo/bench.py $P $args --steps 3 --warmup 1 > $out/pmc_sq_$name.json 2> $out/pmc_sq_$name.err
done
cd $repo
python3 tools/make_traffic_json.py $out > $out/hbm_traffic.json.tmp 2> $out/hbm_traffic.err
traffic_rc=$?
python3 tools/summarize_profiles.py $out > $out/summary.txt 2>&1
# the bench line once more with the traffic just measured on this build next to it (roofline.traffic, valu_issue) -- the tracked, digest-checked file is
# replaced only by a complete one: make_traffic_json.py must have succeeded and its output must parse
if [ $traffic_rc -eq 0 ] && python3 -c "import json, sys; d = json.load(open(sys.argv[1])); assert d" $out/hbm_traffic.json.tmp 2>/dev/null; then
  mv $out/hbm_traffic.json.tmp $out/hbm_traffic.json
  cp $out/hbm_traffic.json $repo/profiles/hbm_traffic.json.tmp && mv $repo/profiles/hbm_traffic.json.tmp $repo/profiles/hbm_traffic.json
else
  echo "profile_round: make_traffic_json.py failed (rc $traffic_rc) or wrote no valid JSON: profiles/hbm_traffic.json is left as it was"; tail -5 $out/hbm_traffic.err
fi
python3 bench.py --no-build > $out/bench_with_traffic.json 2> $out/bench_with_traffic.err
cat $out/summary.txt
