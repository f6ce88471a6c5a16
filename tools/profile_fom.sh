#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel on the GPU box:
#   pass 1: --kernel-trace --stats        -> gpurun_out/prof/kt
#   pass 2: --pmc FETCH_SIZE              -> gpurun_out/prof/fetch      (counters in their own passes,
#   pass 3: --pmc WRITE_SIZE              -> gpurun_out/prof/write       MI355X_MICROARCH.md, rocprofv3 PMC slots)
# then condense into profiles/ with tools/summarize_profile.py <tag> (run that here, after the call).
# usage (from the repo root, on the GPU box):  bash tools/profile_fom.sh
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/kt.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/write.log 2>&1
grep -h '"metric"' $R/gpurun_out/prof/kt.log | tail -1 > $R/gpurun_out/prof/bench_under_profiler.json || true
echo "profiles collected under gpurun_out/prof"
