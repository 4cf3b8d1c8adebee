#!/bin/bash
mkdir -p gpurun_out/r3dp
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "phase_graphs or two_rank or two_process or fp8_phase" > gpurun_out/r3dp/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3dp/tests.log
for mode in "" "--force-dp" "--force-dp --dp-graph" "--force-dp --grad-dtype bf16" "--force-dp --local-stats"; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline $mode > gpurun_out/r3dp/b.json 2> gpurun_out/r3dp/b.err
  python - "$mode" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r3dp/b.json').read().strip().splitlines()[-1])
print("%-40s %.4f ms/step  (%s)" % (sys.argv[1] or "train_pair graph", d['ms_per_step'], d['config']['launch']))
PY
done
