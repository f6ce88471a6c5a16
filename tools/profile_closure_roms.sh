#!/bin/bash
# Kernel-level time split of the host-driven ROM configs (bench.py --config quadratic / ann), rocprofv3 --kernel-trace --stats.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/closure
for c in quadratic ann; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/bench.py --config $c --steps 1 --warmup 1 --time-steps 60 --no-cpu-baseline > /tmp/prof_$c.log 2>&1
  f=$(ls /tmp/prof_$c/*/*kernel_stats.csv | head -1)
  head -25 $f > $R/gpurun_out/closure/${c}_kernel_stats.csv
  grep '"metric"' /tmp/prof_$c.log > $R/gpurun_out/closure/${c}_bench.json
done
