#!/bin/bash
# usage: bash scripts/sweep_env.sh "VAR=val VAR2=val" "..." ; runs the bench (graph mode) once per setting
for cfg in "$@"; do   # (the library no longer reads tuning from the environment: use Engine.set_tuning in a script)
  echo "== $cfg"
  env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['roofline']['step']
km = s['kernel_ms']
g = lambda p: round(sum(v for k, v in km.items() if k.startswith(p)), 3)
print('  steps/s %.1f  ms/step %.4f ' % (d['value'], d['ms_per_step']), {'fwd': g('gemm_bf16_kc_kernel<0'), 'dx': g('gemm_bf16_kc_kernel<1'), 'dw': g('gemm_bf16_ks'), 'aux': round(sum(v for k, v in km.items() if not k.startswith('gemm')), 3)}, d['roofline']['kernel'], d['roofline']['achieved'])"
done
