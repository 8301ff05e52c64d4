#!/bin/bash
# Compiler-flag A/B of the benchmark kernel on a GPU box: msd_kernels_full.hip (the fused first-pass kernels; -DMSD_HOT_ONLY_64X2: the benchmark geometry alone) in a
# handful of scheduling / allocation variants, compiled side by side, each timed on config 1 at 1024 and 8192 scenarios per launch.  A flag the compiler does not know
# fails its build and is reported as such.   usage: tools/gpu_flag_lottery.sh <outdir-name>
name=${1:-lottery}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
declare -A V
V[base]=""
V[nomisched]="-mllvm -enable-misched=false"
V[nopostmisched]="-mllvm -enable-post-misched=false"
V[relaxocc]="-mllvm -amdgpu-schedule-relaxed-occupancy=true"
V[o2]="-O2"
V[os]="-Os"
V[unroll]="__NOSOLVEFLAGS__"
V[maxilp]="-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
V[noagprspill]="-mllvm -amdgpu-spill-vgpr-to-agpr=0"
V[topdown]="-mllvm -misched-topdown"
V[bottomup]="-mllvm -misched-bottomup"
V[norewrite]="-mllvm -amdgpu-enable-rewrite-partial-reg-uses=0"
V[splitall]="-mllvm -split-spill-mode=size"
V[earlyinline]="-mllvm -amdgpu-early-inline-all=true"
for t in "${!V[@]}"; do
  f="${V[$t]}"
  if [ "$f" = "__NOSOLVEFLAGS__" ]; then ( python tools/build_hot.py lot_$t -DMSD_HOT_ONLY_64X2 --no-solve-flags > $O/build_$t.log 2>&1 ) &
  elif [ -z "$f" ]; then ( python tools/build_hot.py lot_$t -DMSD_HOT_ONLY_64X2 > $O/build_$t.log 2>&1 ) &
  else ( python tools/build_hot.py lot_$t -DMSD_HOT_ONLY_64X2 --flags "$f" > $O/build_$t.log 2>&1 ) & fi
done
wait
for rep in 1 2; do
for t in "${!V[@]}"; do
  export MSD_LIB=$R/ms-eetc_amd/lib/variants/libmseetc_hip_lot_$t.so
  if [ ! -f $MSD_LIB ]; then [ $rep = 1 ] && echo "$t: build failed: $(grep -m1 -i "error\|unknown" $O/build_$t.log | cut -c1-150)"; continue; fi
  python3 tools/c1_time.py 100 1024 2>/dev/null | grep "kernel ms median" | sed "s|^[^ ]* |$t rep$rep |" | cut -c1-150
  python3 tools/c1_time.py 100 8192 2>/dev/null | grep "kernel ms median" | sed "s|^[^ ]* |$t rep$rep |" | cut -c1-150
done; done | tee $O/lottery.txt
