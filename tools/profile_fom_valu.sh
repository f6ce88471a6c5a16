#!/bin/bash
# VALU-issue counters of fom_fused_kernel under bench.py (one SQ pass):
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/fom_valu; mkdir -p $R/gpurun_out/fom_valu
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/fom_valu/pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/fom_valu/run.log 2>&1
python - <<PY
import csv, glob, json, os
src = sorted(glob.glob("$R/gpurun_out/fom_valu/pmc/*/*_counter_collection.csv"))[-1]
acc = {}; n = {}; dur = []
seen = set()
for r in csv.DictReader(open(src)):
    if "fom_fused_kernel" not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: acc[k] / n[k] for k in acc}
out = {"kernel": "fom_fused_kernel<16, true, true>", "launches": len(seen), "avg_ms": sum(dur) / len(dur) / 1e6, **m}
wc = m.get("SQ_WAVE_CYCLES")
if wc:
    out["valu_active_over_wave_cycles"] = m.get("SQ_ACTIVE_INST_VALU", 0) / wc
    out["wait_inst_any_over_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0) / wc
    out["wait_any_over_wave_cycles"] = m.get("SQ_WAIT_ANY", 0) / wc
    out["active_inst_any_over_wave_cycles"] = m.get("SQ_ACTIVE_INST_ANY", 0) / wc
json.dump(out, open("$R/gpurun_out/fom_valu_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $R/gpurun_out/fom_valu
