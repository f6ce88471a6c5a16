#!/bin/bash
# One JSON line per BASELINE config on one MI355X -> gpurun_out/r02_bench_all_configs.jsonl (copy into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
: > gpurun_out/r02_bench_all_configs.jsonl
for c in fom pod_galerkin pod_lspg quadratic ann decoder_bf16; do
  python bench.py --config $c 2>gpurun_out/bench_$c.err | tail -1 >> gpurun_out/r02_bench_all_configs.jsonl || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_bench_all_configs.jsonl"):
    d = json.loads(l)
    print("%-14s %.4g %s  frac %.3f  parity %.2e" % (d["config"]["workload"][:14], d["value"], d["unit"], d["roofline"]["frac"], d.get("rel_l2_vs_cpu_ref", float("nan"))))
PY
