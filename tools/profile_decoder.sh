#!/bin/bash
# Kernel-level split of bench.py --config decoder_bf16 (rocprofv3 --kernel-trace --stats), condensed into gpurun_out/r02_profiles/.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_profiles
mkdir -p $O; rm -rf /tmp/dec_kt
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dec_kt -- python $R/bench.py --config decoder_bf16 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/dec_kt.log 2>&1
f=$(ls /tmp/dec_kt/*/*kernel_stats.csv | head -1)
python3 - "$f" "$O/r02_decoder_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(rows[0])
    for r in rows[1:15]:
        w.writerow([r[0][:110]] + r[1:])
PY
grep -h '"metric"' /tmp/dec_kt.log | tail -1 > $O/r02_bench_decoder_under_profiler.json || true
rm -rf /tmp/dec_kt
cat $O/r02_decoder_kernel_stats.csv
