# session A: the suite on the product library, the pattern-initialised diagnostic build against it bit for bit, the random sweeps, the compiler-flag lottery of the hot kernel
tools/gpu_suite.sh r6j
( time python tools/build_variant.py pattern --flags "-ftrivial-auto-var-init=pattern" ) > gpurun_out/r6j/build_pattern.log 2>&1; tail -2 gpurun_out/r6j/build_pattern.log
python tools/compare_libraries.py ms-eetc_amd/lib/libmseetc_hip.so ms-eetc_amd/lib/variants/libmseetc_hip_pattern.so 2>&1 | tee gpurun_out/r6j/compare_pattern.txt | tail -12
tools/gpu_flag_lottery.sh r6j
tools/gpu_sweeps.sh r6j
