#!/bin/bash
# chain kernel iteration: equality with the per-layer launches, mirror parity at the bench size, stamps, bench line
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "chain_launches or (bf16_gradients and 512-4096) or bf16_steps or pair_with_shared" > gpurun_out/r3c/tests.log 2>&1; echo tests rc=$? &&
python scripts/chain_stamps.py > gpurun_out/r3c/chain_stamps.txt 2>&1 &&
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
tail -3 gpurun_out/r3c/tests.log; grep "chain stamps" gpurun_out/r3c/chain_stamps.txt | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'])
r=d['roofline']
for k,v in r['step']['kernel_ms'].items(): print('%-70s %.4f  x%.0f'%(k,v,r['step']['kernel_launches'][k]))
print(r['step']['all_kernels_ms'], r['frac'], r['dominant_kernel']['kernel'], r['dominant_kernel']['frac'])
PY
