"""
One table of a tools/profile_round.sh output directory: kernel name, calls, average duration (rocprofv3 --stats), bench launch time
(HIP events), HBM counter bytes, VALU instructions per IP iteration -- what profiles/<round>/README.md quotes.
usage: python tools/summarize_profiles.py gpurun_out/<tag>
"""
import csv, glob, json, os, sys
out = sys.argv[1]
tr = json.load(open(os.path.join(out, 'hbm_traffic.json'))) if os.path.exists(os.path.join(out, 'hbm_traffic.json')) else {"workloads": {}}
KEYS = dict(c1='c1', c1ref='c1/reference_start', c1b8192='c1/batch8192', c2='c2', c3='c3', intloss='c1/integrate_losses', irk='c1/irk_radau2', cvodes='c1/cvodes_tolerances')
print("%-8s %-52s %5s %12s %10s %11s %9s %9s %9s %9s" % ('workload', 'kernel', 'calls', 'rocprof avg', 'event ms', 'solves/s', 'FETCH MB', 'WRITE MB', 'VALU/it', 'iters'))
for name, key in KEYS.items():
    files = glob.glob(os.path.join(out, 'trace_' + name, '**', '*kernel_stats.csv'), recursive=True)
    jf = os.path.join(out, 'trace_' + name + '.json')
    if not files or not os.path.exists(jf):
        continue
    try:
        line = json.loads([l for l in open(jf).read().splitlines() if l.startswith('{')][-1])
    except Exception:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if 'solve_kernel' in r['Name']]
    if not rows:
        continue
    row = max(rows, key=lambda r: float(r['TotalDurationNs']))      # the dominant kernel (first pass of a split solve); the follow-up kernel's row is in the csv
    w = tr["workloads"].get(key, {})
    its = line['config']['ip_iterations_mean']
    valu = w.get('issue', {}).get('valu_instructions_per_launch')
    print("%-8s %-52s %5s %9.1f us %10.3f %11.0f %9s %9s %9s %9.2f" % (name, row['Name'].split('(')[0][-52:], row['Calls'], float(row['AverageNs'])/1e3, line['roofline']['launch_ms'], line['value'],
          '%.1f' % (w['fetch_size_kb']/1024) if w else '-', '%.1f' % (w['write_size_kb']/1024) if w else '-',
          '%.0f' % (valu/(line['config']['batch_per_gpu']*its)) if valu else '-', its))
