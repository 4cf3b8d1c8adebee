"""Summarise a rocprofv3 kernel trace: per (kernel, grid) average duration and per-step totals.
usage: python scripts/trace_summary.py <kernel_trace.csv> <steps>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    for a, b in (('mrgan::(anonymous namespace)::', ''), ('_ZN5mrgan12_GLOBAL__N_1', ''), ('void ', '')):
        n = n.replace(a, b)
    n = n.split('(')[0][:34]
    key = (n, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
    agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0.0
print('%-36s %6s %4s %3s %4s %4s %6s %6s %8s %9s' % ('kernel', 'blk', 'gy', 'gz', 'vgpr', 'agpr', 'lds', 'calls', 'avg_us', 'us/step'))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    per_step = sum(v) / 1e3 / steps
    if per_step < 0.3:
        continue
    if len(v) < steps:       # not a kernel of the step (the profiled pass's delay kernel, fills and copies of the set-up)
        print('%-36s %6d %4s %3s %4s %4s %6s %6d %8.1f %9s' % (k[0] or '(not part of a step)', k[1], k[2], k[3], k[4], k[5], k[6], len(v), sum(v) / len(v) / 1e3, 'setup'))
        continue
    tot += per_step
    print('%-36s %6d %4s %3s %4s %4s %6s %6d %8.1f %9.1f' % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], len(v), sum(v) / len(v) / 1e3, per_step))
print('total kernel time per step: %.1f us' % tot)
