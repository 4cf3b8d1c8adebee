"""Join a --pmc SQ_VALU_MFMA_BUSY_CYCLES pass (scripts/pmc.sh) with the kernel-trace durations of the same command:
MFMA-pipe busy share per kernel.  usage: python scripts/mfma_util.py <pmc counter_collection.csv> <kernel_trace.csv> [clock GHz]"""
import collections
import csv
import sys

pmc, trace = sys.argv[1], sys.argv[2]
clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4


def short(n):
    return n.replace('mrgan::(anonymous namespace)::', '').replace('void ', '').split('(')[0]


busy, cnt, thr = collections.defaultdict(float), collections.Counter(), {}
for r in csv.DictReader(open(pmc)):
    if r['Counter_Name'] != 'SQ_VALU_MFMA_BUSY_CYCLES':
        continue
    k = (short(r['Kernel_Name']), r['Grid_Size'])
    busy[k] += float(r['Counter_Value'])
    cnt[k] += 1
dur, dcnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(trace)):
    k = (short(r['Kernel_Name']), str(int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)) if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    dur[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
    dcnt[k] += 1
print('%-60s %9s %7s %9s %12s %7s' % ('kernel', 'threads', 'calls', 'avg_us', 'mfma_busy', 'util'))
rows = []
for k in busy:
    if k not in dur or busy[k] == 0:
        continue
    avg_us, b = dur[k] / dcnt[k], busy[k] / cnt[k]
    rows.append((dur[k], k, avg_us, b, b / (avg_us * 1e-6 * clock * 1e9 * 1024)))
for _, k, avg_us, b, u in sorted(rows, reverse=True)[:16]:
    print('%-60s %9s %7d %9.1f %12.3g %7.3f' % (k[0][:60], k[1], dcnt[k], avg_us, b, u))
