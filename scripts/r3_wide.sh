#!/bin/bash
# BASELINE configs[4] geometry on one GPU: per-kernel breakdown of the fp8 and bf16 steps
mkdir -p gpurun_out/r3w
for dt in ${DTYPES:-fp8 bf16}; do
  python bench.py --hidden 4096 --batch 8192 --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/r3w/$dt.json 2> gpurun_out/r3w/$dt.err || exit 1
  python - $dt <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3w/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']; st=r['step']
print(sys.argv[1], 'ms/step %.4f  frac %.4f  allk %.4f'%(d['ms_per_step'], r['frac'], st['all_kernels_ms']))
for k,v in list(st['kernel_ms'].items())[:18]: print('   %-66s %.4f  x%.0f'%(k[:66],v,st['kernel_launches'][k]))
PY
done
