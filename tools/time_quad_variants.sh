#!/bin/bash
# A/B timing of experimental builds of the fused quadratic-manifold kernel (tools/build_variant.sh <name> quad_fused.hip ...):
# usage: tools/time_quad_variants.sh name1 name2 ...   ("product" = the in-tree library)
for v in "$@"; do
  if [ "$v" = product ]; then unset BG_LIB_PATH; else export BG_LIB_PATH=1d-burgers-equation-roms_amd/build/libvar_$v.so; fi
  python bench.py --config quadratic --time-steps ${TS:-100} --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.3g /s'%d['value'], '%.1f ms'%d['ms_per_step'], d['rel_l2_vs_cpu_ref'], d['iters_match_cpu_ref'], d['config']['units_per_pass'])"
done
