#!/bin/bash
# One experimental build of the library with a single source recompiled under extra flags (kernel A/B timing through
# BG_LIB_PATH): build/libvar_<name>.so (git-ignored, travels with gpurun).  Needs an up-to-date product build.
# usage: tools/build_variant.sh <name> <csrc file> [hipcc flags...]     e.g.  tools/build_variant.sh b20 rom_fused.hip -DBG_ACC_BUDGET=20
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../1d-burgers-equation-roms_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c csrc/$src -o build/var_$name.o \
      -Rpass-analysis=kernel-resource-usage 2> build/var_$name.log
objs=$(ls build/*.hip.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/libvar_$name.so $objs build/var_$name.o
grep -E "Function Name|VGPRs:|AGPRs|Scratch|Occupancy|LDS Size" build/var_$name.log | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - - | grep -E "${FILTER:-.}" || true
echo built build/libvar_$name.so
