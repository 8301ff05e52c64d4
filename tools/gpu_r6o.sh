#!/bin/bash
# session: after the build -- the loss-table families (LDS-resident table head, 128 x 1 geometry): parity, timing, phase cycles
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6o; cd $R
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dynamic or loss or table or tabulated" 2>&1 | tail -n 4
python -m pytest tests/test_integrated_loss_table.py -q -m gpu -x 2>&1 | tail -n 3
for n in 100 120 300; do python tools/dyn_time.py $n 2>&1 | tail -n 1; done | tee $O/dyn_time.txt
MSD_GEOMETRY2=64x2 python tools/dyn_time.py 100 2>&1 | tail -n 1 | tee -a $O/dyn_time.txt
python tools/c1_time.py 100 1024 | head -n 1
python bench.py --no-build > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); a=b["alt"]
print(b["value"]); print({k:a["dynamic_losses_N100"][k] for k in ("solves_per_s","launch_ms","launch_ms_with_them")}, a["dynamic_losses_N300"]["solves_per_s"], a["dynamic_losses_integrated_N100"]["solves_per_s"], a["dynamic_losses_integrated_N100"]["launch_ms"])
PY
