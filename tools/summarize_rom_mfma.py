#!/usr/bin/env python3
"""Condense the rocprofv3 PMC pass of tools/profile_rom_mfma.sh: per ROM kernel, launches with the full batch
only (the largest grid), mean counter values and the derived MFMA utilisation.

  MfmaUtil   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs)      (fraction of SIMD-cycles the matrix
               pipe is busy while the CU is busy; the gfx94x derived-metric formula)
  flop       = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512                               (one MOP = 512 flop)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sorted(glob.glob(os.path.join(REPO, "gpurun_out", "rom_mfma", "pmc", "*", "*_counter_collection.csv")), key=os.path.getmtime)
rows = [r for f in src for r in csv.DictReader(open(f))]          # both bench runs (Galerkin, LSPG)
per = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch -> counter -> value
meta = {}
for r in rows:
    k = r["Kernel_Name"]
    if "rom_reduce" not in k and "lu_solve" not in k and "rom_fused" not in k:
        continue
    d = r["Dispatch_Id"]
    per[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[(k, d)] = (int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
out = []
for k, disp in per.items():
    gmax = max(meta[(k, d)][0] for d in disp)
    full = [d for d in disp if meta[(k, d)][0] == gmax]
    durs = sorted(meta[(k, d)][1] for d in full)
    keep = [d for d in full if meta[(k, d)][1] >= 0.8 * durs[-1]]            # launches with every sample active
    n = len(keep)
    mean = lambda c: sum(disp[d].get(c, 0.0) for d in keep) / n
    dur = sum(meta[(k, d)][1] for d in keep) / n
    mfma_busy, cu_busy = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_BUSY_CU_CYCLES")
    mops = mean("SQ_INSTS_VALU_MFMA_MOPS_F64")
    out.append({"kernel": (re.search(r"(rom_reduce\w*<[^>]*>|lu_solve_kernel<[^>]*>|rom_fused_kernel<[^>]*>)", k) or [k[:80]])[0], "launches": n, "avg_us": dur / 1e3,
                "SQ_VALU_MFMA_BUSY_CYCLES": mfma_busy, "SQ_BUSY_CU_CYCLES": cu_busy,
                "MfmaUtil": mfma_busy / (4.0 * cu_busy) if cu_busy else None,
                "SQ_INSTS_VALU_MFMA_MOPS_F64": mops, "SQ_INSTS_VALU_MFMA_F64": mean("SQ_INSTS_VALU_MFMA_F64"),
                "mfma_tflops": mops * 512 / dur / 1e3 if dur else None,
                "SQ_ACTIVE_INST_VALU": mean("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": mean("SQ_WAVE_CYCLES"),
                "vgpr": meta[(k, keep[0])][2], "agpr": meta[(k, keep[0])][3], "lds": meta[(k, keep[0])][4], "scratch": meta[(k, keep[0])][5]})
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "rom_mfma_summary.json")
json.dump(out, open(dst, "w"), indent=1)
for o in out:
    print(json.dumps(o))
