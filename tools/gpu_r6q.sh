# session: the restoration problem's Newton system on the scan (msd_resto_scan.hpp) -- parity of the restoration / watchdog tests, loose-schedule sweeps, timing
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6q; cd $R; mkdir -p $O
python -m pytest tests/test_restoration.py tests/test_watchdog.py -q -m gpu -x 2>&1 | tail -n 4
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "randomized or long_horizons or loose" 2>&1 | tail -n 3
SWEEP_FACTORS=2.5,3 timeout 900 python tests/tools/random_sweep.py 0 200 > $O/random_sweep_loose_2.5_3.txt 2>&1; tail -n 3 $O/random_sweep_loose_2.5_3.txt
SWEEP_FACTORS=4,6 timeout 900 python tests/tools/random_sweep.py 0 200 > $O/random_sweep_very_loose_4_6.txt 2>&1; tail -n 4 $O/random_sweep_very_loose_4_6.txt
python bench.py --no-build > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); a=b["alt"]
print(b["value"]); print(a["loose_schedules_reference_start"]["restoration"], a["loose_schedules_reference_start"]["restart_only"]["kernel_ms"]); print(a["loose_schedules_long_horizons"])
print(a["c4"]["warm"]["resolves_per_s"], a["c4"]["cold"]["resolves_per_s"])
PY
