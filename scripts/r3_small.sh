#!/bin/bash
# per-kernel breakdown of the small-batch steps
mkdir -p gpurun_out/r3s
for b in ${BATCHES:-50 512}; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch $b ${BENCH_ARGS} > gpurun_out/r3s/b$b.json 2> gpurun_out/r3s/b$b.err || exit 1
  python - $b <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3s/b%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']['step']
print('batch', sys.argv[1], 'ms/step %.4f allk %.4f launches %d'%(d['ms_per_step'], r['all_kernels_ms'], sum(r['kernel_launches'].values())))
for k,v in r['kernel_ms'].items(): print('   %-66s %.4f  x%.0f'%(k,v,r['kernel_launches'][k]))
PY
done
