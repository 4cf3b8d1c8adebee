#!/bin/bash
# usage: bash scripts/sweep_env.sh "VAR=val VAR2=val" "..." ; runs the bench (graph mode) once per setting
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['roofline']['step']
print('  steps/s %.1f  ms/step %.4f ' % (d['value'], d['ms_per_step']), {k: round(v,3) for k,v in s['kernel_ms'].items()})"
done
