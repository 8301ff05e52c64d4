"""
HBM bytes per launch of the solve kernel for every workload profiled by tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE passes) and the
issue statistics of the SQ pass, stamped with the digest of the kernel sources they were measured on (bench.py reports
`roofline.traffic` only when the digest matches the library it runs).
usage: python tools/make_traffic_json.py gpurun_out/<tag>  > profiles/hbm_traffic.json
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

out = sys.argv[1]
KEYS = dict(c1='c1', c1ref='c1/reference_start', c1b8192='c1/batch8192', c2='c2', c3='c3', intloss='c1/integrate_losses', irk='c1/irk_radau2', cvodes='c1/cvodes_tolerances')


def part_of(kernel_name):
    "template argument PART of a solve kernel: 1 / 3 first pass of a split solve, 2 its follow-up kernel, 0 a kernel that holds everything"
    import re
    m = re.search(r'solve_kernel<([^>]*)>', kernel_name)
    args = [a.strip() for a in m.group(1).split(',')] if m else []
    return int(args[7]) if len(args) >= 8 else 0


def mean_counter(d, counter, which='launch'):
    """
    Mean of a counter per launch of the solver: the first kernel of a launch (first pass, or the one kernel) plus -- which = 'launch' -- the
    follow-up kernel of a split solve behind it.  which = 'first': the dominant kernel alone.  Returns (value, launches).
    """
    first, follow = [], []
    for f in glob.glob(os.path.join(out, d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'solve_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                (follow if part_of(r['Kernel_Name']) == 2 else first).append(float(r['Counter_Value']))
    if not first and follow:      # a launch that is the follow-up kernel's alone
        first, follow = follow, []
    if not first:
        return None, 0
    v = sum(first)/len(first)
    if which == 'launch' and follow:
        v += sum(follow)/len(first)
    return v, len(first)


rec = {"kernel_digest": entry.hip_digest(),
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (tools/profile_round.sh), means per launch of the solver over a pass -- bytes: first-pass kernel + follow-up "
                 "kernel of a split solve; issue statistics: the first-pass (dominant) kernel alone; "
                 "FETCH_SIZE as reported (the x2 gfx950 correction of MI355X_MICROARCH.md is calibrated for 16 B/lane streams; this kernel's traffic is 8 B/lane "
                 "scratch and result stores, a width the guide calls uncalibrated; bytes_per_launch_with_fetch_doubled applies it anyway -- WRITE_SIZE dominates either way)",
       "workloads": {}}
for name, key in KEYS.items():
    fetch_kb, nf = mean_counter('pmc_fetch_' + name, 'FETCH_SIZE')
    write_kb, nw = mean_counter('pmc_write_' + name, 'WRITE_SIZE')
    if fetch_kb is None or write_kb is None:
        continue
    line = json.load(open(os.path.join(out, 'pmc_fetch_' + name + '.json')))
    w = {"bytes_per_launch": int(1024*(fetch_kb + write_kb)), "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
         "bytes_per_launch_with_fetch_doubled": int(1024*(2*fetch_kb + write_kb)), "launches_measured": [nf, nw],
         "launch": "{}; {} scenarios, {:.2f} IP iterations per solve".format(line['roofline']['kernel'], line['config']['batch_per_gpu'], line['config']['ip_iterations_mean']),
         "compulsory_bytes_per_launch": line['roofline']['compulsory_bytes_per_launch'],
         "stage_iterations_per_launch": line['roofline']['stage_iterations_per_launch']}
    # issue statistics of the same kernel from the SQ pass (units of four cycles per wave, summed over the waves of a launch)
    sq = {k: mean_counter('pmc_sq_' + name, k, 'first')[0] for k in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU')}
    if all(v is not None for v in sq.values()):
        w["issue"] = {"valu_instructions_per_launch": sq['SQ_INSTS_VALU'], "lds_instructions_per_launch": sq['SQ_INSTS_LDS'], "salu_instructions_per_launch": sq['SQ_INSTS_SALU'],
                      "valu_instructions_per_stage_iteration": sq['SQ_INSTS_VALU']/w['stage_iterations_per_launch'],
                      "wave_life_executing": sq['SQ_ACTIVE_INST_ANY']/sq['SQ_WAVE_CYCLES'], "wave_life_waiting_on_counters": sq['SQ_WAIT_ANY']/sq['SQ_WAVE_CYCLES'],
                      "wave_life_waiting_for_instructions": sq['SQ_WAIT_INST_ANY']/sq['SQ_WAVE_CYCLES']}
    rec["workloads"][key] = w
if not rec["workloads"]:
    raise SystemExit("no solve_kernel rows in the counter files under " + out)
print(json.dumps(rec, indent=1))
