#!/bin/bash
# usage: bash scripts/pmc.sh <tag> "<counters>" -- <program args...>   (counters in their own pass; no trace domains)
tag=$1; ctrs=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$tag
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_$tag/*/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('mrgan::(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    key = (n, r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
seen = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('mrgan::(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    seen[((n, r['Grid_Size']), r['Counter_Name'])] += 1
for key, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:24]:
    print(key, {c: '%.4g' % (v / seen[(key, c)]) for c, v in d.items()}, 'dispatches', max(seen[(key, c)] for c in d))
PY
