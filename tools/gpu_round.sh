#!/bin/bash
# one GPU session of round 4: tests, benchmarks, follow lists, horizon timing -> gpurun_out/<name>/     (PYTEST_ARGS, e.g. "-x")
name=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -n 12
python tools/follow_probe.py c1 c1_8192 c2 c3 c1ref 2>&1 | tail -n 6 | tee $O/follow.txt
WORKLOADS="c1: c1_8192:--batch_8192_--steps_60 c1ref:--start_reference_--steps_100 c2:--workload_c2_--steps_20 c3:--workload_c3_--steps_30 c4:--workload_c4_--steps_3" tools/ab_hot.sh $name product 2>&1 | tail -n 8
if [ -n "$VARIANTS" ]; then WORKLOADS="c1: c1ref:--start_reference_--steps_100 c2:--workload_c2_--steps_20" tools/ab_hot.sh $name $VARIANTS 2>&1 | tail -n 8; fi
python tools/horizon_timing.py 1024 2>&1 | grep -v "N  *700\|N 1000\|N 2000" | tee $O/horizon.txt | tail -n 16
