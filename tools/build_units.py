"""
Tuning / diagnostic aid: rebuild some translation units with extra -D switches and link them with the other objects of the product library into
ms-eetc_amd/lib/variants/libmseetc_hip_<tag>.so (select with MSD_LIB=<path>).  The product library must be built from the same headers.

    python tools/build_units.py <tag> unit.hip [unit2.hip ...] [-DNAME=VALUE ...]
"""
import subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry   # noqa: E402

tag = sys.argv[1]
units = [a for a in sys.argv[2:] if a.endswith('.hip')]
defs = [a for a in sys.argv[2:] if a.startswith('-D')]
out = entry.PKG / 'lib' / 'variants'
obj = out / ('obj_' + tag)
obj.mkdir(parents=True, exist_ok=True)
csrc = entry.PKG / 'csrc'
t0 = time.time()
jobs = []
for u in units:
    fl = [f for f in entry.HIP_FLAGS if f != '-shared'] + (entry.SOLVE_KERNEL_FLAGS if u.startswith('msd_kernels_') else []) + defs
    jobs.append(subprocess.Popen([entry.HIPCC] + fl + ['-c', '-o', str(obj / (u + '.o')), str(csrc / u)]))
assert not any(j.wait() for j in jobs)
prod = entry.PKG / 'lib' / 'obj'
objs = [str(obj / (u + '.o')) if u in units else str(prod / (u + '.o')) for u in entry.UNITS]
lib = out / 'libmseetc_hip_{}.so'.format(tag)
subprocess.run([entry.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib)] + objs, check=True)
print(lib, '%.0f s' % (time.time() - t0))
