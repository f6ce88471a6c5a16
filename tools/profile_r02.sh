#!/bin/bash
# Round-2 evidence in one GPU call: rocprofv3 passes of bench.py (FOM default; fused ROM kernels), condensed ON the box
# into gpurun_out/r02_profiles/ (raw traces are too large to travel back); copy that directory's files into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_fom.sh > gpurun_out/prof_fom.log 2>&1
bash tools/profile_rom_mfma.sh > gpurun_out/prof_rom.log 2>&1
mkdir -p gpurun_out/r02_profiles
python tools/summarize_profile.py r02 > gpurun_out/r02_profiles/summarize_fom.log 2>&1
cp profiles/r02_kernel_stats.csv profiles/r02_pmc.csv profiles/fom_pmc_summary.json gpurun_out/r02_profiles/
python tools/summarize_rom_mfma.py gpurun_out/r02_profiles/r02_rom_fused_mfma_pmc.json > gpurun_out/r02_profiles/summarize_rom.log 2>&1
for t in gal lspg; do
  f=$(ls gpurun_out/rom_mfma/kt_$t/*/*kernel_stats.csv | head -1)
  head -12 $f | cut -c1-220 > gpurun_out/r02_profiles/r02_rom_fused_${t}_kernel_stats.csv
  grep -h '"metric"' gpurun_out/rom_mfma/kt_$t.log | tail -1 > gpurun_out/r02_profiles/r02_bench_${t}_under_profiler.json || true
done
cp gpurun_out/prof/bench_under_profiler.json gpurun_out/r02_profiles/r02_bench_fom_under_profiler.json || true
rm -rf gpurun_out/prof gpurun_out/rom_mfma
ls -la gpurun_out/r02_profiles
