"""
Tuning aid: rebuild only the first-pass kernels of the benchmark family (msd_kernels_full.hip: the fused iteration) with extra -D switches /
flags and link them with the other objects of the product library into ms-eetc_amd/lib/variants/libmseetc_hip_<tag>.so (select with
MSD_LIB=<path>).  The product library must be built (python __graft_entry__.py) from the same headers.

    python tools/build_hot.py <tag> [-DNAME=VALUE ...] [--flags "<extra hipcc flags>"] [--unit msd_kernels_rg2.hip ...]
"""
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry   # noqa: E402


def main():
    tag = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith('-D')]
    extra = sys.argv[sys.argv.index('--flags') + 1].split() if '--flags' in sys.argv else []
    units = [sys.argv[k + 1] for k, a in enumerate(sys.argv) if a == '--unit'] or ['msd_kernels_full.hip']      # (--unit may be given several times)
    out = entry.PKG / 'lib' / 'variants'
    obj = out / ('obj_' + tag)
    obj.mkdir(parents=True, exist_ok=True)
    csrc = entry.PKG / 'csrc'
    flags = [f for f in entry.HIP_FLAGS if f != '-shared'] + ([] if '--no-solve-flags' in sys.argv else entry.SOLVE_KERNEL_FLAGS) + defs + extra
    t0 = time.time()
    jobs = [subprocess.Popen([entry.HIPCC] + flags + ['-c', '-o', str(obj / (unit + '.o')), str(csrc / unit)]) for unit in units]
    if any(j.wait() for j in jobs):
        raise SystemExit("hipcc failed")
    prod = entry.PKG / 'lib' / 'obj'
    objs = [str(obj / (u + '.o')) if u in units else str(prod / (u + '.o')) for u in entry.UNITS]
    lib = out / 'libmseetc_hip_{}.so'.format(tag)
    subprocess.run([entry.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib)] + objs, check=True)
    print(lib, '%.0f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
