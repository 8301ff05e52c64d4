# session: the loss-table family with the rolling stock's structure compiled in -- parity and timing
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6t; cd $R; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dynamic or loss or table or tabulated or rolling or config3 or override" 2>&1 | tail -n 4
python -m pytest tests/test_integrated_loss_table.py tests/test_restoration.py tests/test_reference_unit_tests.py -q -m gpu -x 2>&1 | tail -n 3
for n in 60 100 120 200 300; do python tools/dyn_time.py $n 2>&1 | tail -n 1; done | tee $O/dynamic_loss_timing.txt
timeout 900 python tests/tools/random_sweep_loss_functions.py 0 60 2>&1 | tail -n 2
python tools/c1_time.py 100 1024 | head -n 1
