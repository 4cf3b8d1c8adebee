#!/bin/bash
mkdir -p gpurun_out/r3c
for v in 0 1; do
  for ab in 0 7936; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --min-seconds 0.05 --ablate $ab > gpurun_out/r3c/ka_${v}_$ab.json 2> gpurun_out/r3c/ka_${v}_$ab.err
  python - $v $ab <<'PY'
import json, sys
v, ab = sys.argv[1:3]
d = json.loads(open('gpurun_out/r3c/ka_%s_%s.json' % (v, ab)).read().strip().splitlines()[-1])
k = d['roofline']['step']['kernel_ms']
print("DEV_KERNARG=%s ablate %5s  chain<0> %.4f  chain<1> %.4f  chain<2> %.4f   step %.4f  allk %.4f" % (v, ab, k.get('chain_kernel<0>', 0), k.get('chain_kernel<1>', 0), k.get('chain_kernel<2>', 0), d['ms_per_step'], d['roofline']['step']['all_kernels_ms']))
PY
  done
done
env | grep -i -E "HIP_|HSA_|ROC" | head
