#!/bin/bash
# The GPU test suite file by file, each in a process of its own (a kernel that faults aborts the interpreter: the other files still run), the tests of
# tests/test_gpu_parity.py in groups.  usage: tools/gpu_suite.sh <outdir-name> [pytest args]      -> gpurun_out/<outdir-name>/suite.txt
name=${1:-suite}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
: > $O/suite.txt
run() {   # tag, pytest selection...
  tag=$1; shift
  python -m pytest -m gpu -q -p no:cacheprovider --timeout 1500 "$@" > $O/$tag.log 2>&1
  rc=$?
  echo "$tag rc $rc: $(grep -E 'passed|failed|error' $O/$tag.log | tail -1)" | tee -a $O/suite.txt
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Fatal Python|Memory access fault|HSA_STATUS|Aborted" $O/$tag.log | head -20 | sed 's/^/    /' | tee -a $O/suite.txt; fi
}
for f in tests/test_*.py; do
  b=$(basename $f .py)
  if [ $b = test_gpu_parity ]; then
    run parity_randomized $f -k "randomized" "$@"
    run parity_geometry $f -k "every_launch_geometry or long_horizons or short_horizons or deterministic" "$@"
    run parity_config $f -k "config or full_batches or figure or minimum_time or fixture" "$@"
    run parity_rest $f -k "not randomized and not every_launch_geometry and not long_horizons and not short_horizons and not deterministic and not config and not full_batches and not figure and not minimum_time and not fixture" "$@"
  else
    grep -q "mark.gpu\|pytestmark" $f && run $b $f "$@"
  fi
done
echo "---"; cat $O/suite.txt | grep -c "rc 0:" | xargs echo "groups green:"; grep -v "rc 0:" $O/suite.txt | grep " rc " 
