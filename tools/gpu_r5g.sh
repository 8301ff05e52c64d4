MSD_LIB=$PWD/ms-eetc_amd/lib/variants/libmseetc_hip_telem.so python tools/phase_cycles.py > gpurun_out/r5g_phase.txt 2>&1
python tools/ls_stats.py > gpurun_out/r5g_ls.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err
cat gpurun_out/r5g_phase.txt gpurun_out/r5g_ls.txt; tail -c 1500 gpurun_out/r5g_bench.json
