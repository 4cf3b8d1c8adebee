#!/bin/bash
# wall time of the three chain launches with parts of the kernel switched off (results wrong; timing only)
mkdir -p gpurun_out/r3c
for ab in ${ABLS:-0 256 512 1024 2048 4096 3072 4864 7936}; do
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --min-seconds 0.05 --ablate $ab > gpurun_out/r3c/abl_$ab.json 2> gpurun_out/r3c/abl_$ab.err
  python - $ab <<'PY'
import json, sys
ab = sys.argv[1]
d = json.loads(open('gpurun_out/r3c/abl_%s.json' % ab).read().strip().splitlines()[-1])
k = d['roofline']['step']['kernel_ms']
print("ablate %5s  chain<0> %.4f  chain<1> %.4f  chain<2> %.4f   step %.4f" % (ab, k.get('chain_kernel<0>', 0), k.get('chain_kernel<1>', 0), k.get('chain_kernel<2>', 0), d['ms_per_step']))
PY
done
