#!/bin/bash
# quick regression + bench line: fp32 / bf16 step parity, chain equality, bench
mkdir -p gpurun_out/r3q
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "${TESTS:-fp32_steps or bf16_steps or chain_launches or (bf16_gradients and 512-4096) or pair_with_shared or graph_replay or supervised}" > gpurun_out/r3q/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3q/tests.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/r3q/bench.json 2> gpurun_out/r3q/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3q/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'])
r=d['roofline']
for k,v in r['step']['kernel_ms'].items(): print('%-70s %.4f  x%.0f'%(k,v,r['step']['kernel_launches'][k]))
print(r['step']['all_kernels_ms'], r['frac'], r['dominant_kernel']['kernel'], r['dominant_kernel']['frac'])
PY
