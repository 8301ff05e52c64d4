# last session of round 6 (after a host-side change of msd_post.hip: the kernels of the solver are the ones profiled before): GPU suite, the round's profiles, config-4 statistics
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r6final
python -c "import __graft_entry__ as g; print('library stale:', g.stale())"
tools/gpu_suite.sh r6final | tail -6
tools/profile_round.sh r6final_prof > gpurun_out/r6final/profile_round.log 2>&1; tail -14 gpurun_out/r6final/profile_round.log
tools/gpu_c4prof.sh r6final_c4 2>&1 | tail -4
python tools/kernel_meta.py > gpurun_out/r6final/kernel_registers.txt 2>&1
