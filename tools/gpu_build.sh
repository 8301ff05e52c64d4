#!/bin/bash
# Build the HIP library on the GPU box (16 cores there, 8 in the build container) and bring the objects back: gpurun_out/<name>/obj.tgz -> ms-eetc_amd/lib/obj/
# (then `python __graft_entry__.py` only links).  Optionally runs a session script behind the build: tools/gpu_build.sh <name> [script args...]
name=${1:-build}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$name
mkdir -p $O/obj
cd $R
( time python3 __graft_entry__.py ) > $O/build.log 2>&1; echo "build rc $?" >> $O/build.log; tail -n 4 $O/build.log
# (gpurun merges at most 64 MiB back: the objects travel compressed; unpack with `tar -xzf gpurun_out/<name>/obj.tgz -C ms-eetc_amd/lib/obj`)
rmdir $O/obj 2>/dev/null
tar -czf $O/obj.tgz -C ms-eetc_amd/lib/obj $(cd ms-eetc_amd/lib/obj && ls *.o *.stamp)
du -sh $O/obj.tgz
if [ -n "$1" ]; then "$@"; fi
