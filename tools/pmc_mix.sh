#!/bin/bash
# Instruction mix and wait cycles of the solve kernel (rocprofv3 --pmc, one pass per counter group). usage: tools/pmc_mix.sh <tag> [bench args]
tag=${1:-mix}; shift
out=$(pwd)/gpurun_out/$tag; mkdir -p $out; repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o g$i -- python3 $repo/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/g$i.json 2> $out/g$i.err
done
cd $repo
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('$out/g*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'solve_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print('%-28s %16.0f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
