#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (tools/profile_r03.sh) of one bench config into one JSON record per hot kernel.

usage: summarize_pmc.py <dir with sq/ fetch/ write/ sub-directories of rocprofv3 output> <kernel substring> [<substring> ...]
For every kernel whose name contains one of the substrings: the launches with the largest grid, mean counters, and
  MfmaUtil      = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)       (gfx94x derived-metric formula)
  valu_active   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                      (both in quad-cycles)
  hbm bytes     = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024                 (KiB; FETCH doubled on gfx950, MI355X_MICROARCH section HBM)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root, keys = sys.argv[1], sys.argv[2:]
out = {}
for sub in ("sq", "fetch", "write"):
    rows = [r for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")) for r in csv.DictReader(open(f))]
    per = defaultdict(lambda: defaultdict(dict)); meta = {}
    for r in rows:
        k = r["Kernel_Name"]
        if not any(s in k for s in keys):
            continue
        d = r["Dispatch_Id"]
        per[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[(k, d)] = (int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"],
                        r["LDS_Block_Size"], r["Scratch_Size"])
    for k, disp in per.items():
        gmax = max(meta[(k, d)][0] for d in disp)
        keep = [d for d in disp if meta[(k, d)][0] == gmax]
        n = len(keep)
        name = re.sub(r"^void \(anonymous namespace\)::", "", k)[:90]
        o = out.setdefault(name, {"kernel": name})
        o.update({"grid": gmax, "vgpr": meta[(k, keep[0])][2], "agpr": meta[(k, keep[0])][3], "lds_bytes": meta[(k, keep[0])][4],
                  "scratch_bytes": meta[(k, keep[0])][5]})
        o[f"{sub}_pass_launches"] = n
        o[f"{sub}_pass_avg_us"] = sum(meta[(k, d)][1] for d in keep) / n / 1e3
        for c in sorted({c for d in keep for c in disp[d]}):
            o[c] = sum(disp[d].get(c, 0.0) for d in keep) / n
for o in out.values():
    if o.get("SQ_BUSY_CU_CYCLES"):
        o["MfmaUtil"] = o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * o["SQ_BUSY_CU_CYCLES"])
    if o.get("SQ_WAVE_CYCLES"):
        o["valu_active_fraction"] = o.get("SQ_ACTIVE_INST_VALU", 0.0) / o["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in o:
            o["wait_any_fraction"] = o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in o and o.get("sq_pass_avg_us"):
        o["mfma_f64_tflops_under_profiler"] = o["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / o["sq_pass_avg_us"] / 1e6
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_fetch_bytes_per_launch"] = o["FETCH_SIZE"] * 1024 * 2
        o["hbm_write_bytes_per_launch"] = o["WRITE_SIZE"] * 1024
        o["hbm_bytes_per_launch"] = o["hbm_fetch_bytes_per_launch"] + o["hbm_write_bytes_per_launch"]
print(json.dumps(list(out.values()), indent=1))
