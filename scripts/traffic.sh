#!/bin/bash
# HBM traffic per kernel from PMC counters, one counter family per pass (TCC slots), as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --profile-steps 2 --no-cpu-baseline --no-graph --min-seconds 0 > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % c)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c: continue
        n = r['Kernel_Name'].replace('mrgan::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        agg[n] += float(r['Counter_Value']); cnt[n] += 1
    for n in agg:
        out.setdefault(n, {})[c + '_KB_per_launch'] = agg[n] / cnt[n]
        out[n]['launches'] = cnt[n]
for n, d in out.items():
    # gfx950: FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact
    d['hbm_bytes_per_launch'] = (2.0 * d.get('FETCH_SIZE_KB_per_launch', 0.0) + d.get('WRITE_SIZE_KB_per_launch', 0.0)) * 1024.0
import sys
sys.path.insert(0, '.')
from bench import source_sha
commit = __import__('os').environ.get('MRGAN_COMMIT', 'unknown')
json.dump({'commit': commit, 'source_sha': source_sha(), 'command': 'scripts/traffic.sh (two --pmc passes: FETCH_SIZE, WRITE_SIZE; bench.py --steps 10 --warmup 2 --profile-steps 2 --no-cpu-baseline --no-graph)', 'kernels': out}, open('gpurun_out/traffic.json', 'w'), indent=1, sort_keys=True)
for n, d in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
    print('%-52s launches %4d  fetch %9.0f KB  write %9.0f KB  -> %.2f MB/launch' % (n[:52], d['launches'], d.get('FETCH_SIZE_KB_per_launch', 0), d.get('WRITE_SIZE_KB_per_launch', 0), d['hbm_bytes_per_launch'] / 1e6))
PY
