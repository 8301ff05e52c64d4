# product build on the GPU box (objects come back in obj.tgz), the 64 x 1 one-brake follow-up kernel as a variant of two units, the suite on the product, the short-horizon tests on the variant
tools/gpu_build.sh r6i
python tools/build_hot.py f64x1 --unit msd_api.hip --unit msd_kernels_rg2.hip -DMSD_FOLLOW_64X1 > gpurun_out/r6i/build_f64x1.log 2>&1; tail -1 gpurun_out/r6i/build_f64x1.log
tools/gpu_suite.sh r6i
export MSD_LIB=$PWD/ms-eetc_amd/lib/variants/libmseetc_hip_f64x1.so
echo "== 64 x 1 follow-up variant"
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "short_horizons or deterministic or one_brake or randomized" 2>&1 | tail -3
python tools/determinism_probe.py 15 0 2>&1 | tail -7
