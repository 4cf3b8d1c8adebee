#!/bin/bash
# A/B of two builds of the library within one box (box-to-box timing differs by +-5 %): mr_gan_amd/lib/libmrgan_hip.so against
# mr_gan_amd/lib/libmrgan_hip_old.so (built from another checkout with `make LIB=../lib/libmrgan_hip_old.so BUILD=build_old`)
mkdir -p gpurun_out/r3ab
L=mr_gan_amd/lib
cp $L/libmrgan_hip.so /tmp/new.so; cp $L/libmrgan_hip_old.so /tmp/old.so
for rep in 1 2; do for v in new old; do
  cp /tmp/$v.so $L/libmrgan_hip.so
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/r3ab/${v}_$rep.json 2> gpurun_out/r3ab/${v}_$rep.err || exit 1
  python - $v $rep <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3ab/%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
k=d['roofline']['step']['kernel_ms']
print(sys.argv[1], sys.argv[2], 'ms/step %.4f allk %.4f'%(d['ms_per_step'], d['roofline']['step']['all_kernels_ms']), ' '.join('%s %.4f'%(n[:24],v) for n,v in k.items() if 'chain' in n or 'ks_group' in n))
PY
done; done
cp /tmp/new.so $L/libmrgan_hip.so
