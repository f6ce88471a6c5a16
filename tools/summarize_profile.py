#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{kt,fetch,write}) into profiles/.

usage: summarize_profile.py <round-tag> [kernel-substring]
Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary, long PyTorch
kernel names shortened), profiles/<tag>_pmc.csv (per-dispatch FETCH_SIZE / WRITE_SIZE of the
hot kernel) and profiles/fom_pmc_summary.json, which bench.py reads for roofline.traffic.

HBM bytes follow MI355X_MICROARCH.md section HBM: the counters are in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (doubled here), WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "fom_fused_kernel"
src = os.path.join(REPO, "gpurun_out", "prof")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

def short(n):
    return n if len(n) < 140 else n[:137] + "..."

stats = sorted(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
avg_ns = None
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = short(r[0]); w.writerow(r)
            if key in r[0]:
                avg_ns = float(r[3])
summary = {"round": tag, "kernel": key, "avg_ns_kernel_trace": avg_ns,
           "config": {"config": "fom", "batch": 1024, "n": 1024, "time_steps": 500, "dt": 0.025}}     # what tools/profile_fom.sh runs
pm = {}
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["counter", "dispatch_id", "value_KiB", "vgpr", "sgpr", "lds", "scratch", "duration_ns"])
    for name in ("fetch", "write"):
        for p in sorted(glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)[:1]:
            for r in csv.DictReader(open(p)):
                if key in r["Kernel_Name"]:
                    w.writerow([r["Counter_Name"], r["Dispatch_Id"], r["Counter_Value"], r["VGPR_Count"], r["SGPR_Count"],
                                r["LDS_Block_Size"], r["Scratch_Size"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
                    pm.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    fetch = sum(pm["FETCH_SIZE"]) / len(pm["FETCH_SIZE"]) * 1024 * 2      # gfx950: x2 for wide coalesced reads
    write = sum(pm["WRITE_SIZE"]) / len(pm["WRITE_SIZE"]) * 1024
    summary.update({"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "hbm_bytes_per_launch": fetch + write,
                    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x2 (gfx950)"})
json.dump(summary, open(os.path.join(dst, "fom_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
