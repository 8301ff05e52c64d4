"""
Static instruction mix per phase of the first-pass kernel.  A unit compiled with -DMSD_TELEMETRY=1 carries a cycle-counter read (s_memtime) at every
phase boundary (Ctx::mark); the code between two of them is one phase, in program order (the cold blocks the compiler moved out of line -- the
non-benchmark integrator branches, error exits -- are counted where it put them).
    python tools/build_units.py telem msd_kernels_full.hip -DMSD_HOT_ONLY_64X2 -DMSD_TELEMETRY=1
    python tools/phase_mix.py [kernel-name-substring]      (disassembles ms-eetc_amd/lib/variants/obj_telem/msd_kernels_full.hip.o)
"""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
LLVM = Path('/opt/rocm/lib/llvm/bin')
want = sys.argv[1] if len(sys.argv) > 1 else 'Li64ELi2ELi1ELi0ELb0ELb0ELi1ELi1E'
obj = ROOT / 'ms-eetc_amd' / 'lib' / 'variants' / 'obj_telem' / 'msd_kernels_full.hip.o'
with tempfile.TemporaryDirectory() as td:
    fat, co = Path(td) / 'fat.bin', Path(td) / 'k.co'
    subprocess.run([str(LLVM / 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', str(obj), str(fat)], check=True)
    subprocess.run([str(LLVM / 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + str(fat), '--output=' + str(co), '--unbundle'], check=True)
    txt = subprocess.run([str(LLVM / 'llvm-objdump'), '-d', str(co)], capture_output=True, text=True, check=True).stdout


def cls(i):
    if i.startswith('v_accvgpr'): return 'accvgpr'
    if i.startswith('scratch_'): return 'scratch'
    if i.startswith('ds_bpermute'): return 'bpermute'
    if i.startswith('ds_'): return 'lds'
    if i.startswith(('global_', 'buffer_', 'flat_')): return 'global'
    if i.startswith(('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64')): return 'rcp/rsq'
    if i.startswith(('v_div_', )): return 'ieee div'
    if i.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')): return 'fp64 arith'
    if i.startswith(('v_max_f64', 'v_min_f64')): return 'fp64 max'
    if i.startswith('v_mov') and 'dpp' in i: return 'dpp'
    if i.startswith(('v_mov', 'v_readlane', 'v_readfirstlane', 'v_writelane')): return 'moves'
    if i.startswith('v_'): return 'valu other'
    if i.startswith('s_waitcnt'): return 'waitcnt'
    if i.startswith(('s_cbranch', 's_branch')): return 'branch'
    if i.startswith('s_'): return 'salu'
    return 'other'


cols = ['fp64 arith', 'fp64 max', 'rcp/rsq', 'ieee div', 'valu other', 'moves', 'dpp', 'accvgpr', 'lds', 'bpermute', 'scratch', 'global', 'salu', 'branch', 'waitcnt']
parts = re.split(r'\n[0-9a-f]+ <([^>]+)>:\n', txt)
for name, body in zip(parts[1::2], parts[2::2]):
    if want not in name:
        continue
    segs = [collections.Counter()]
    for l in body.split('\n'):
        m = re.match(r'\s+(\S+)', l)
        if not m:
            continue
        if m.group(1).startswith('s_memtime'):
            segs.append(collections.Counter())
            continue
        segs[-1][cls(m.group(1))] += 1
    print(subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:90])
    print('%-4s %7s | ' % ('seg', 'total') + ' '.join('%10s' % c for c in cols))
    for k, c in enumerate(segs):
        print('%-4d %7d | ' % (k, sum(c.values())) + ' '.join('%10d' % c[x] for x in cols))
