#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/profile.sh <tag> [bench args...]
# writes gpurun_out/prof_<tag>/ and prints the kernel stats; copy the csv into profiles/ afterwards.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --min-seconds 0 "$@" > gpurun_out/prof_$tag.log 2>&1
cat gpurun_out/prof_$tag/*/*_kernel_stats.csv | cut -c1-200 | head -16
grep -E '^\{' gpurun_out/prof_$tag.log | cut -c1-400
