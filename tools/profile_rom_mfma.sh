#!/bin/bash
# MFMA-utilisation counters of the ROM reduce kernels (bench_rom.py, POD Galerkin + LSPG):
#   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64
#             SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES  (one pass, SQ block only)
# then condense with: python tools/summarize_rom_mfma.py
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/rom_mfma; mkdir -p $R/gpurun_out/rom_mfma
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/rom_mfma/pmc -- python $R/bench_rom.py --which pod_galerkin pod_lspg --time-steps 10 > $R/gpurun_out/rom_mfma/run.log 2>&1
echo "collected"
