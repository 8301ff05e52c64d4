"""
Tuning aid: build the HIP library with extra -D switches into ms-eetc_amd/lib/variants/libmseetc_hip_<tag>.so, next to the product
library (which __graft_entry__.build() owns).  Select it at run time with MSD_LIB=<path>.  Only the geometries for N <= 255 are
instantiated (MSD_MINIMAL_GEOMETRIES) to keep the build short.

    python tools/build_variant.py <tag> [-DNAME=VALUE ...] [--flags "<extra hipcc flags>"]
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry   # noqa: E402


def main():
    tag = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith('-D')]
    extra = []
    if '--flags' in sys.argv:
        extra = sys.argv[sys.argv.index('--flags') + 1].split()
    solve_flags = entry.SOLVE_KERNEL_FLAGS
    if '--no-solve-flags' in sys.argv:
        solve_flags = []
    out = entry.PKG / 'lib' / 'variants'
    obj = out / ('obj_' + tag)
    obj.mkdir(parents=True, exist_ok=True)
    csrc = entry.PKG / 'csrc'
    flags = [f for f in entry.HIP_FLAGS if f != '-shared'] + ['-DMSD_MINIMAL_GEOMETRIES=1'] + defs
    jobs = []
    for u in entry.UNITS:
        fl = flags + ((solve_flags + extra) if u.startswith('msd_kernels_') else [])
        jobs.append(subprocess.Popen([entry.HIPCC] + fl + ['-c', '-o', str(obj / (u + '.o')), str(csrc / u)]))
    codes = [j.wait() for j in jobs]
    if any(codes):
        raise SystemExit("hipcc failed: {}".format(codes))
    lib = out / 'libmseetc_hip_{}.so'.format(tag)
    subprocess.run([entry.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib)] + [str(obj / (u + '.o')) for u in entry.UNITS], check=True)
    print(lib)


if __name__ == '__main__':
    main()
