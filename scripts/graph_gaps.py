"""Kernel durations and the gaps between consecutive kernels inside the replayed (D, G) graph, from a rocprofv3 kernel trace of
`python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --min-seconds 0 --profile-steps 0`-like runs (graph replay).
usage: python scripts/graph_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
# steady part: the longest run of launches whose gaps are all below 50 us
best, cur = [], []
for i, k in enumerate(ks):
    if cur and k[0] - cur[-1][1] > 50000:
        if len(cur) > len(best):
            best = cur
        cur = []
    cur.append(k)
if len(cur) > len(best):
    best = cur
dur = sum(e - s for s, e, _ in best)
gaps = [best[i + 1][0] - best[i][1] for i in range(len(best) - 1)]
span = best[-1][1] - best[0][0]
print("launches in the steady run: %d, span %.1f us" % (len(best), span / 1e3))
print("kernel time %.1f %% of the span, gaps %.1f %%; mean kernel %.2f us, mean gap %.2f us (median %.2f, max %.2f)" % (
    100.0 * dur / span, 100.0 * sum(gaps) / span, dur / len(best) / 1e3, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3, max(gaps) / 1e3))
