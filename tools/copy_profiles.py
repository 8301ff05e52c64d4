"""
Copy what is to be judged from a tools/profile_round.sh output directory into profiles/<round>/ (tracked): per workload the rocprofv3 kernel
statistics (`--kernel-trace --stats`) and the counter means per launch over the solve-kernel rows (one --pmc pass each), the bench line, the
summary table, the text outputs of the other tools found there; the per-workload traffic goes to profiles/hbm_traffic.json (bench.py reports
roofline.traffic from it when its kernel digest matches the library it runs).
usage: python tools/copy_profiles.py gpurun_out/<tag> profiles/<round>
"""
import collections, csv, glob, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for d in sorted(glob.glob(os.path.join(src, 'trace_*'))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len('trace_'):]
    for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
        shutil.copy(f, os.path.join(dst, name + '_kernel_stats.csv'))
    for kind in ('fetch', 'write', 'sq'):
        files = glob.glob(os.path.join(src, 'pmc_%s_%s' % (kind, name), '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(files[0])):
            if 'solve_kernel' in row['Kernel_Name']:
                acc[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
        with open(os.path.join(dst, '%s_pmc_%s.csv' % (name, kind)), 'w') as out:
            out.write('kernel,counter,mean_per_launch,launches\n')
            for (k, c), v in sorted(acc.items()):
                out.write('"%s",%s,%g,%d\n' % (k, c, sum(v)/len(v), len(v)))
for f in ('bench.json', 'summary.txt', 'horizon_timing.txt', 'geometry_sweep.txt', 'random_sweep.txt', 'random_sweep_transcriptions.txt', 'resto_probe.txt', 'resto_probe2.txt'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
if os.path.exists(os.path.join(src, 'hbm_traffic.json')):
    shutil.copy(os.path.join(src, 'hbm_traffic.json'), os.path.join(os.path.dirname(os.path.abspath(dst)), 'hbm_traffic.json'))
print('copied', src, '->', dst)
