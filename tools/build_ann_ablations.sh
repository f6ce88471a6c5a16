#!/bin/bash
# Ablation builds of the fused POD-ANN kernel for tools/time_ann_fused.py: libabl_<mask>.so under build/ (git-ignored,
# travels with gpurun).  usage: tools/build_ann_ablations.sh 0 128 256 384 ...   (needs an up-to-date product build)
set -e
cd "$(dirname "$0")/../1d-burgers-equation-roms_amd"
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DBG_FUSED_ABLATE=$m -c csrc/rom_ann_fused.hip -o build/ann_abl_$m.o
  objs=$(ls build/*.hip.o | grep -v rom_ann_fused)
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libabl_$m.so $objs build/ann_abl_$m.o
  echo built build/libabl_$m.so
done
