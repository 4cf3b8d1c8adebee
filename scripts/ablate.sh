#!/bin/bash
# per-category kernel time of one bench step under each ablation bit (eager launches, hipEvent timing)
for a in ${ABLATE_LIST:-0 1 8 9 2 4 6}; do
  echo "ablate=$a"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --ablate $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['roofline']['step']; print('  ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in s['kernel_ms'].items() if k.startswith('gemm')})"
done
