#!/bin/bash
# round-3 evidence for profiles/: kernel stats, MFMA-busy PMC pass, HBM traffic PMC passes, bench lines (run as the last GPU action)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3prof
# the PMC traffic passes first: the bench lines below quote profiles/r03_traffic.json when it was measured on the sources they run
MRGAN_COMMIT=${MRGAN_COMMIT:-unknown} bash scripts/traffic.sh > gpurun_out/r3prof/traffic.log 2>&1 || exit 1
cp gpurun_out/traffic.json gpurun_out/r3prof/traffic.json
cp gpurun_out/traffic.json profiles/r03_traffic.json
python bench.py > gpurun_out/r3prof/bench_final.json 2> gpurun_out/r3prof/bench_final.err || exit 1
bash scripts/profile.sh r03 > gpurun_out/r3prof/profile.log 2>&1 || exit 1
python scripts/trace_summary.py gpurun_out/prof_r03/*/*_kernel_trace.csv 265 > gpurun_out/r3prof/kernel_trace_summary.txt
cp gpurun_out/prof_r03/*/*_kernel_stats.csv gpurun_out/r3prof/kernel_stats.csv
grep -E '^\{' gpurun_out/prof_r03.log > gpurun_out/r3prof/bench_under_rocprof.json
bash scripts/pmc.sh mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" -- python3 bench.py --steps 10 --warmup 2 --profile-steps 2 --no-cpu-baseline --no-graph --min-seconds 0 > gpurun_out/r3prof/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_mfma -- python3 bench.py --steps 10 --warmup 2 --profile-steps 2 --no-cpu-baseline --no-graph --min-seconds 0 > gpurun_out/r3prof/tr_mfma.log 2>&1 || exit 1
python scripts/mfma_util.py gpurun_out/pmc_mfma/*/*counter_collection.csv gpurun_out/tr_mfma/*/*_kernel_trace.csv > gpurun_out/r3prof/mfma_util.txt
python bench.py --force-dp --steps 100 --no-cpu-baseline > gpurun_out/r3prof/bench_dp_protocol_on_one_gpu.json 2>/dev/null
python bench.py --global-batch 512 --steps 100 --no-cpu-baseline > gpurun_out/r3prof/bench_512_rows.json 2>/dev/null
python bench.py --batch 50 --rows 6000 --d 1200 --steps 200 --no-cpu-baseline > gpurun_out/r3prof/bench_reference_size.json 2>/dev/null
for shape in "--d 3632 --batch 512" "--d 2432 --batch 1024" "--d 800 --batch 1024" "--d 400 --batch 1024"; do
  python bench.py $shape --rows 65536 --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['roofline'].pop('families',None); print(json.dumps(d))" >> gpurun_out/r3prof/bench_other_shapes.jsonl
done
python bench.py --hidden 4096 --batch 8192 --dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3prof/bench_wide_fp8.json 2>/dev/null
python bench.py --hidden 4096 --batch 8192 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3prof/bench_wide_bf16.json 2>/dev/null
tail -3 gpurun_out/r3prof/mfma_util.txt; tail -4 gpurun_out/r3prof/traffic.log
