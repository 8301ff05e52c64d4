"""
HBM bytes per launch of the solve kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, stamped with the
digest of the kernel sources it was measured on (bench.py reports `roofline.traffic` only when the digest matches the library).
usage: python tools/make_traffic_json.py gpurun_out/<tag>  > profiles/hbm_traffic.json
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

out = sys.argv[1]


def mean_counter(tag, counter):
    vals = []
    for f in glob.glob(os.path.join(out, 'pmc_' + tag, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'solve_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals.append(float(r['Counter_Value']))
    return (sum(vals)/len(vals), len(vals)) if vals else (None, 0)


fetch_kb, nf = mean_counter('fetch', 'FETCH_SIZE')
write_kb, nw = mean_counter('write', 'WRITE_SIZE')
if fetch_kb is None or write_kb is None:
    raise SystemExit("no solve_kernel rows in the counter files under " + out)
line = json.load(open(os.path.join(out, 'pmc_fetch.json')))
rec = {
    "bytes_per_launch": int(1024*(fetch_kb + write_kb)),
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "bytes_per_launch_with_fetch_doubled": int(1024*(2*fetch_kb + write_kb)),      # upper reading: the guide's x2 for 16 B/lane streams applied anyway
    "kernel_digest": entry.hip_digest(),
    "launch": "{}; {} scenarios, {:.2f} IP iterations per solve".format(line['roofline']['kernel'], line['config']['batch_per_gpu'], line['config']['ip_iterations_mean']),
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh), mean of {} / {} solve-kernel launches; FETCH_SIZE as reported "
              "(the x2 gfx950 correction of MI355X_MICROARCH.md is calibrated for 16 B/lane streams; the kernel's traffic is 8 B/lane scratch and result stores, a width the guide calls uncalibrated; bytes_per_launch_with_fetch_doubled is the reading with the correction applied anyway -- WRITE_SIZE dominates either way)".format(nf, nw),
    "compulsory_bytes_per_launch": int(line['config']['batch_per_gpu']*(8*(5*line['config']['num_intervals'] + 2) + 168)),
}
# issue statistics of the same kernel from the SQ pass (units of four cycles per wave, summed over the waves of a launch)
sq = {k: mean_counter('sq', k)[0] for k in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU')}
if all(v is not None for v in sq.values()):
    rec["issue"] = {"valu_instructions_per_launch": sq['SQ_INSTS_VALU'], "lds_instructions_per_launch": sq['SQ_INSTS_LDS'], "salu_instructions_per_launch": sq['SQ_INSTS_SALU'],
                    "wave_life_executing": sq['SQ_ACTIVE_INST_ANY']/sq['SQ_WAVE_CYCLES'], "wave_life_waiting_on_counters": sq['SQ_WAIT_ANY']/sq['SQ_WAVE_CYCLES'],
                    "wave_life_waiting_for_instructions": sq['SQ_WAIT_INST_ANY']/sq['SQ_WAVE_CYCLES'],
                    "source": "rocprofv3 --pmc SQ_* pass of tools/profile_round.sh (one wave per SIMD: nothing hides a wave's own latencies)"}
print(json.dumps(rec, indent=1))
