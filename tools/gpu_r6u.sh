# session: streamed first pass with the rolling stock's structure compiled in -- parity on the long horizons, timing
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6u; cd $R; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "long_horizons or every_launch_geometry or gpops or figure or loose or randomized" 2>&1 | tail -n 4
python -m pytest tests/test_restoration.py tests/test_watchdog.py tests/test_solution_fixtures.py -q -m gpu -x 2>&1 | tail -n 3
python tools/horizon_timing.py --only=560,600,700,1000,2000 2>&1 | tee $O/horizon_timing_long.txt
python tools/long_horizon_timing.py 2>&1 | tail -n 8
