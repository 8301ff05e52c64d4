#!/bin/bash
# A/B aid: config-1 kernel time and a parity sample with variant libraries (tools/build_units.py); nothing here builds the product library.
# usage: tools/gpu_ab_variant.sh "<tag> [<tag> ...]" [pytest -k expression | none]
kexpr=${2:-"config1 or config2 or config3"}
for tag in $1; do
  export MSD_LIB=$PWD/ms-eetc_amd/lib/variants/libmseetc_hip_$tag.so
  python tools/c1_time.py 100 1024; python tools/c1_time.py 100 8192 | head -1
  if [ "$kexpr" != none ]; then python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "$kexpr" 2>&1 | tail -3; fi
done
