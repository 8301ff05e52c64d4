# session B: product build on the GPU box, the round's profiles (kernel statistics, counters, bench line), config-4 kernel statistics, horizon timing, register metadata,
# the pattern-initialised build against the product (tolerance rule of tools/compare_libraries.py)
tools/gpu_build.sh r6k
tools/gpu_suite.sh r6k | tail -4
tools/profile_round.sh r6k_prof > gpurun_out/r6k/profile_round.log 2>&1; tail -25 gpurun_out/r6k/profile_round.log
tools/gpu_c4prof.sh r6k_c4 2>&1 | tail -14
python tools/horizon_timing.py > gpurun_out/r6k/horizon_timing.txt 2>&1; tail -20 gpurun_out/r6k/horizon_timing.txt
python tools/kernel_meta.py > gpurun_out/r6k/kernel_registers.txt 2>&1
( time python tools/build_variant.py pattern --flags "-ftrivial-auto-var-init=pattern" ) > gpurun_out/r6k/build_pattern.log 2>&1; tail -2 gpurun_out/r6k/build_pattern.log
python tools/compare_libraries.py ms-eetc_amd/lib/libmseetc_hip.so ms-eetc_amd/lib/variants/libmseetc_hip_pattern.so 2>&1 | tee gpurun_out/r6k/compare_pattern.txt | tail -12
