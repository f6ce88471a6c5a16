#!/bin/bash
# rocprofv3 evidence for the device-side POD-ANN loop (bench.py --config ann -> rom_ann_fused_kernel), condensed on the box
# into gpurun_out/r02_profiles/: kernel-trace statistics of a full-length pass, and one SQ counter pass (issue / LDS / wave cycles).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_profiles
mkdir -p $O; rm -rf /tmp/ann_kt /tmp/ann_pmc
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ann_kt -- python $R/bench.py --config ann --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ann_kt.log 2>&1
f=$(ls /tmp/ann_kt/*/*kernel_stats.csv | head -1)
head -8 $f | cut -c1-260 > $O/r02_rom_ann_fused_kernel_stats.csv
grep -h '"metric"' /tmp/ann_kt.log | tail -1 > $O/r02_bench_ann_under_profiler.json || true
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d /tmp/ann_pmc -- python $R/bench.py --config ann --steps 1 --warmup 1 --time-steps 40 --no-cpu-baseline > /tmp/ann_pmc.log 2>&1
python - "$O/r02_rom_ann_fused_pmc.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
rows = [r for f in glob.glob("/tmp/ann_pmc/*/*_counter_collection.csv") for r in csv.DictReader(open(f))]
per = defaultdict(lambda: defaultdict(dict)); meta = {}
for r in rows:
    k = r["Kernel_Name"]
    if "rom_ann_fused" not in k:
        continue
    d = r["Dispatch_Id"]
    per[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[(k, d)] = (int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
out = []
for k, disp in per.items():
    dmax = max(meta[(k, d)][1] for d in disp)
    keep = [d for d in disp if meta[(k, d)][1] >= 0.5 * dmax]       # the working kernel, not the repair kernel's empty pass
    n = len(keep)
    mean = lambda c: sum(disp[d].get(c, 0.0) for d in keep) / n
    o = {"kernel": k[:120], "launches": n, "avg_us": sum(meta[(k, d)][1] for d in keep) / n / 1e3,
         "vgpr": meta[(k, keep[0])][2], "agpr": meta[(k, keep[0])][3], "lds": meta[(k, keep[0])][4], "scratch": meta[(k, keep[0])][5]}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
        o[c] = mean(c)
    if o["SQ_WAVE_CYCLES"]:
        o["valu_issue_fraction_of_wave_cycles"] = o["SQ_ACTIVE_INST_VALU"] / o["SQ_WAVE_CYCLES"]
        o["lds_issue_fraction_of_wave_cycles"] = o["SQ_ACTIVE_INST_LDS"] / o["SQ_WAVE_CYCLES"]
    out.append(o)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for o in out:
    print(json.dumps(o))
PY
rm -rf /tmp/ann_kt /tmp/ann_pmc
ls -la $O
