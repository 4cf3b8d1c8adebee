#!/bin/bash
# in-kernel phase breakdowns (diagnostic build, make STAMPS=1) and launch-level ablations of the GEMM kernels at the bench shapes
mkdir -p gpurun_out/r3st
MRGAN_BENCH_LIB=mr_gan_amd/lib/libmrgan_hip_stamps.so python scripts/gemm_bench.py > gpurun_out/r3st/kc_stamps.txt 2>&1 || exit 1
python scripts/gemm_bench.py 0 4 2 7 > gpurun_out/r3st/kc_ablation.txt 2>&1 || exit 1
python scripts/chain_stamps.py > gpurun_out/r3st/chain_stamps_raw.txt 2>&1 || exit 1
grep "chain stamps" gpurun_out/r3st/chain_stamps_raw.txt | tail -3 > gpurun_out/r3st/chain_stamps.txt
CHAIN_ABLATE=7936 python scripts/chain_stamps.py 2>&1 | grep "chain stamps" | tail -3 > gpurun_out/r3st/chain_stamps_all_phases_ablated.txt
tail -3 gpurun_out/r3st/chain_stamps.txt
