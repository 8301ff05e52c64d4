"""
Tuning aid: recompile the named units (default: msd_kernels_full.hip) into the product library and stamp it as current, e.g.
    python tools/quick_units.py msd_kernels_full2.hip msd_api.hip
The other units keep their objects: only valid while the edit does not change what they compile (run
`python __graft_entry__.py --force` before committing).
"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as e
csrc = e.PKG / 'csrc'
objdir = e.PKG / 'lib' / 'obj'
units = [a for a in sys.argv[1:] if a.endswith('.hip')] or ['msd_kernels_full.hip']
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
