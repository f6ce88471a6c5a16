#!/bin/bash
# Round-3 evidence in one GPU call, condensed ON the box into gpurun_out/r03_profiles/ (copy its files into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench.py command (FOM headline + the other_configs runs)
#   2. per ROM config (short passes): one SQ pass (matrix-pipe / VALU activity), one FETCH_SIZE pass, one WRITE_SIZE pass
#      (counters in their own passes, no trace domains besides --kernel-trace: MI355X_MICROARCH, rocprofv3 PMC slots)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_profiles
W=/tmp/r03_prof
rm -rf $W; mkdir -p $O $W
cd /tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -- python $R/bench.py --no-cpu-baseline > $W/kt.log 2>&1
f=$(ls $W/kt/*/*kernel_stats.csv | head -1)
python3 - "$f" "$O/r03_default_bench_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as g:
    w = csv.writer(g); w.writerow(rows[0])
    for r in rows[1:26]:
        w.writerow([r[0][:120]] + r[1:])
PY
grep -h '"metric"' $W/kt.log | tail -1 > $O/r03_bench_default_under_profiler.json || true
echo "kernel trace done"
for cfg in ${CONFIGS:-pod_galerkin pod_lspg quadratic ann pod_r96_galerkin pod_r96_lspg decoder_bf16}; do
  case $cfg in
    pod_r96_*) key="rom_wide_kernel"; ts=40;;
    decoder_bf16) key="decode_mlp_kernel"; ts=500;;
    pod_*) key="rom_fused_kernel"; ts=40;;
    quadratic) key="quad_fused_kernel"; ts=40;;
    ann) key="rom_ann_fused_kernel"; ts=40;;
  esac
  mkdir -p $W/$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $W/$cfg/sq -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --time-steps $ts --no-cpu-baseline > $W/$cfg/sq.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/$cfg/fetch -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --time-steps $ts --no-cpu-baseline > $W/$cfg/fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/$cfg/write -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --time-steps $ts --no-cpu-baseline > $W/$cfg/write.log 2>&1
  python3 $R/tools/summarize_pmc.py $W/$cfg $key > $O/r03_${cfg}_pmc.json
  grep -h '"metric"' $W/$cfg/sq.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('value','ms_per_step','steps') } | {'units_per_pass': d['config']['units_per_pass'], 'workload': d['config']['workload']}))" > $O/r03_${cfg}_bench_under_sq_pass.json || true
  echo "$cfg done"; cat $O/r03_${cfg}_pmc.json | head -60
done
# HBM traffic of the hot kernel at the FULL bench configuration (what bench.py reports as roofline.traffic of these configs)
if [ -z "$NO_FULL_TRAFFIC" ]; then
for cfg in ${CONFIGS:-pod_galerkin pod_lspg quadratic ann pod_r96_galerkin pod_r96_lspg decoder_bf16}; do
  case $cfg in
    pod_r96_*) key="rom_wide_kernel";;
    decoder_bf16) key="decode_mlp_kernel";;
    pod_*) key="rom_fused_kernel";;
    quadratic) key="quad_fused_kernel";;
    ann) key="rom_ann_fused_kernel";;
  esac
  mkdir -p $W/full_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/full_$cfg/fetch -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > $W/full_$cfg/fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/full_$cfg/write -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > $W/full_$cfg/write.log 2>&1
  python3 $R/tools/summarize_pmc.py $W/full_$cfg $key > $O/r03_${cfg}_full_traffic.json
  echo "$cfg full-size traffic done"
done
python3 - $O <<'PY'
import glob, json, os, sys
o = sys.argv[1]
out = {"round": "r03", "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --config NAME --steps 1 --warmup 0` (the full "
       "bench configuration); KiB -> bytes; FETCH_SIZE x2 (gfx950, MI355X_MICROARCH section HBM); the launch with the largest grid", "configs": {}}
for f in sorted(glob.glob(os.path.join(o, "r03_*_full_traffic.json"))):
    cfg = os.path.basename(f)[4:-len("_full_traffic.json")]
    recs = [r for r in json.load(open(f)) if "hbm_bytes_per_launch" in r]
    if recs:
        r = max(recs, key=lambda r: r["hbm_bytes_per_launch"])
        out["configs"][cfg] = {"kernel": r["kernel"], "hbm_bytes_per_launch": r["hbm_bytes_per_launch"], "fetch_bytes_per_launch": r["hbm_fetch_bytes_per_launch"],
                               "write_bytes_per_launch": r["hbm_write_bytes_per_launch"]}
json.dump(out, open(os.path.join(o, "rom_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
fi
rm -rf $W
ls -la $O
