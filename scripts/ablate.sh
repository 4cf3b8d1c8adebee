#!/bin/bash
# per-category kernel time of one bench step under each ablation bit (eager launches, hipEvent timing)
for a in ${ABLATE_LIST:-0 2 4 6}; do
  echo "ablate=$a"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --ablate $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); km=d['roofline']['step']['kernel_ms']
g=lambda p: round(sum(v for k,v in km.items() if k.startswith(p)),3)
print('  ms/step %.3f'%d['ms_per_step'], {'fwd': g('gemm_bf16_kc_kernel<0'), 'dx': g('gemm_bf16_kc_kernel<1'), 'dw': g('gemm_bf16_ks')})"
done
