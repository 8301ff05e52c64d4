#!/bin/bash
# Compiler-flag A/B of the benchmark kernel on a GPU box: msd_kernels_full.hip (the fused first-pass kernels; -DMSD_HOT_ONLY_64X2: the benchmark geometry alone) in a
# handful of scheduling / allocation variants, compiled side by side, each timed on config 1 at 1024 and 8192 scenarios per launch.  A flag the compiler does not know
# fails its build and is reported as such.   usage: tools/gpu_flag_lottery.sh <outdir-name> [variants-file]
name=${1:-lottery}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O
cd $R
declare -A V
if [ -n "$2" ]; then
  # variants from a file: lines "tag|<build_hot.py arguments: -D switches and/or --flags "...">"
  while IFS='|' read -r t f; do [ -n "$t" ] && V[$t]="$f"; done < "$2"
else
V[base]=""
V[nomisched]="--flags \"-mllvm -enable-misched=false\""
V[nopostmisched]="--flags \"-mllvm -enable-post-misched=false\""
V[relaxocc]="--flags \"-mllvm -amdgpu-schedule-relaxed-occupancy=true\""
V[o2]="--flags -O2"
V[os]="--flags -Os"
V[unroll]="--no-solve-flags"
V[noagprspill]="--flags \"-mllvm -amdgpu-spill-vgpr-to-agpr=0\""
V[norewrite]="--flags \"-mllvm -amdgpu-enable-rewrite-partial-reg-uses=0\""
V[splitall]="--flags \"-mllvm -split-spill-mode=size\""
V[earlyinline]="--flags \"-mllvm -amdgpu-early-inline-all=true\""
fi
for t in "${!V[@]}"; do
  ( eval python tools/build_hot.py lot_$t -DMSD_HOT_ONLY_64X2 ${V[$t]} > $O/build_$t.log 2>&1 ) &
done
wait
for rep in 1 2; do
for t in "${!V[@]}"; do
  export MSD_LIB=$R/ms-eetc_amd/lib/variants/libmseetc_hip_lot_$t.so
  if [ ! -f $MSD_LIB ]; then [ $rep = 1 ] && echo "$t: build failed: $(grep -m1 -i "error\|unknown" $O/build_$t.log | cut -c1-150)"; continue; fi
  python3 tools/c1_time.py 100 1024 2>/dev/null | grep "kernel ms median" | sed "s|^[^ ]* |$t rep$rep |" | cut -c1-150
  python3 tools/c1_time.py 100 8192 2>/dev/null | grep "kernel ms median" | sed "s|^[^ ]* |$t rep$rep |" | cut -c1-150
done; done | tee $O/lottery.txt
