"""
Tuning aid: recompile only msd_kernels_full.hip (and msd_api.hip with --api) into the product library and stamp it as current.
The other units keep their objects: only valid while the edit does not change what they compile (run __graft_entry__.build(force=True) before committing).
"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as e
csrc = e.PKG / 'csrc'
objdir = e.PKG / 'lib' / 'obj'
units = ['msd_kernels_full.hip'] + (['msd_api.hip'] if '--api' in sys.argv else [])
flags = [f for f in e.HIP_FLAGS if f != '-shared']
jobs = []
for u in units:
    fl = flags + (e.SOLVE_KERNEL_FLAGS if u.startswith('msd_kernels_') else [])
    jobs.append(subprocess.Popen([e.HIPCC] + fl + ['-c', '-o', str(objdir / (u + '.o')), str(csrc / u)]))
assert not any(j.wait() for j in jobs)
lib = e.PKG / 'lib' / 'libmseetc_hip.so'
subprocess.run([e.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib)] + [str(objdir / (u + '.o')) for u in e.UNITS], check=True)
lib.with_name(lib.name + '.stamp').write_text(e.hip_digest())
print('ok')
