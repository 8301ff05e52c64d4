# final session of round 6 (library built and linked beforehand: gpurun_out/r6o objects): GPU suite, the round's profiles, config-4 kernel statistics, horizon timing,
# register metadata, the sweeps, the pattern-initialised build against the product
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r6x
python -c "import __graft_entry__ as g; print('library stale:', g.stale())"
tools/gpu_suite.sh r6x | tail -6
tools/profile_round.sh r6x_prof > gpurun_out/r6x/profile_round.log 2>&1; tail -14 gpurun_out/r6x/profile_round.log
tools/gpu_c4prof.sh r6x_c4 2>&1 | tail -4
python tools/horizon_timing.py > gpurun_out/r6x/horizon_timing.txt 2>&1; tail -30 gpurun_out/r6x/horizon_timing.txt
python tools/kernel_meta.py > gpurun_out/r6x/kernel_registers.txt 2>&1
for n in 100 120 200 300; do python tools/dyn_time.py $n 2>&1 | tail -n 1; done | tee gpurun_out/r6x/dynamic_loss_timing.txt
tools/gpu_sweeps.sh r6x 2>&1 | tail -12
( time python tools/build_variant.py pattern --flags "-ftrivial-auto-var-init=pattern" ) > gpurun_out/r6x/build_pattern.log 2>&1; tail -2 gpurun_out/r6x/build_pattern.log
python tools/compare_libraries.py ms-eetc_amd/lib/libmseetc_hip.so ms-eetc_amd/lib/variants/libmseetc_hip_pattern.so 2>&1 | tee gpurun_out/r6x/compare_pattern.txt | tail -8
