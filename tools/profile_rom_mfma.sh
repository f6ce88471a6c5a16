#!/bin/bash
# MFMA-utilisation counters of the ROM kernels (bench.py --config pod_galerkin / pod_lspg: the fused bg_rom_run kernel):
#   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64
#             SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES  (one pass, SQ block only)
# then condense with: python tools/summarize_rom_mfma.py
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/rom_mfma; mkdir -p $R/gpurun_out/rom_mfma
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/rom_mfma/pmc -- python $R/bench.py --config pod_galerkin --steps 1 --warmup 1 --time-steps 40 --no-cpu-baseline > $R/gpurun_out/rom_mfma/run_gal.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/rom_mfma/pmc -- python $R/bench.py --config pod_lspg --steps 1 --warmup 1 --time-steps 40 --no-cpu-baseline > $R/gpurun_out/rom_mfma/run_lspg.log 2>&1
# kernel-trace statistics of full-length passes (no counters)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rom_mfma/kt_gal -- python $R/bench.py --config pod_galerkin --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rom_mfma/kt_gal.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rom_mfma/kt_lspg -- python $R/bench.py --config pod_lspg --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rom_mfma/kt_lspg.log 2>&1
echo "collected"
