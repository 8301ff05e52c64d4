#!/bin/bash
# the random sweeps of the round on the final library -> gpurun_out/<name>/
name=${1:-sweeps}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
timeout 1500 python tests/tools/random_sweep.py 0 450 > $O/random_sweep_0_450.txt 2>&1; tail -n 3 $O/random_sweep_0_450.txt
SWEEP_FACTORS=2.5,3 timeout 900 python tests/tools/random_sweep.py 0 200 > $O/random_sweep_loose_2.5_3.txt 2>&1; tail -n 3 $O/random_sweep_loose_2.5_3.txt
SWEEP_FACTORS=4,6 timeout 900 python tests/tools/random_sweep.py 0 200 > $O/random_sweep_very_loose_4_6.txt 2>&1; tail -n 6 $O/random_sweep_very_loose_4_6.txt
timeout 900 python tests/tools/random_sweep_transcriptions.py 0 40 > $O/random_sweep_transcriptions.txt 2>&1; tail -n 2 $O/random_sweep_transcriptions.txt
timeout 900 python tests/tools/random_sweep_loss_functions.py 0 60 > $O/random_sweep_loss_functions.txt 2>&1; tail -n 2 $O/random_sweep_loss_functions.txt
