"""
Register / spill / scratch metadata of every solve kernel in the built objects (ms-eetc_amd/lib/obj/*.o):
    python tools/kernel_meta.py [unit-substring]
Extracts the gfx950 code object from each object's .hip_fatbin section and reads the kernel descriptors' notes.
"""
import re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
LLVM = Path('/opt/rocm/lib/llvm/bin')
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for obj in sorted((ROOT / 'ms-eetc_amd' / 'lib' / 'obj').glob('*.o')):
    if pat not in obj.name:
        continue
    with tempfile.TemporaryDirectory() as td:
        fat, co = Path(td) / 'fat.bin', Path(td) / 'k.co'
        subprocess.run([str(LLVM / 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', str(obj), str(fat)], check=True)
        r = subprocess.run([str(LLVM / 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + str(fat), '--output=' + str(co), '--unbundle'],
                           capture_output=True, text=True)
        if r.returncode or not co.exists():
            continue
        notes = subprocess.run([str(LLVM / 'llvm-readelf'), '--notes', str(co)], capture_output=True, text=True).stdout
    print('==', obj.name)
    for blk in notes.split('- .agpr_count')[1:]:
        g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
        name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(.*', '', name)
        print('  %-62s vgpr %4s spill %5s sgpr_spill %4s scratch %5s B lds %6s' % (name, g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
