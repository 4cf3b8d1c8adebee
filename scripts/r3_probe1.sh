#!/bin/bash
# round-3 probe: baseline bench line, per-shape GEMM timings with ablation bits, in-kernel stamp breakdowns
set -o pipefail
mkdir -p gpurun_out/r3p1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3p1/bench0.json 2> gpurun_out/r3p1/bench0.err &&
python scripts/gemm_bench.py 0 2 4 > gpurun_out/r3p1/gemm_bench.txt 2>&1 &&
MRGAN_BENCH_LIB=mr_gan_amd/lib/libmrgan_hip_stamps.so python scripts/gemm_bench.py 0 > gpurun_out/r3p1/gemm_stamps.txt 2>&1 &&
python scripts/chain_stamps.py > gpurun_out/r3p1/chain_stamps.txt 2>&1
echo done $?
