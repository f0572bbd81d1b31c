#!/bin/bash
# Tuning aid: bench several builds of k_gather (raytracing_folder_amd/lib/variants/librt_<name>.so,
# same ABI, different -DRT_GATHER_* knobs) in one GPU session.  Usage: tools_gather_variants.sh name[:blocks] ...
for spec in "$@"; do
  name=${spec%%:*}; blocks=${spec#*:}; [ "$blocks" = "$spec" ] && blocks=1024
  RT_MI355X_LIB=$PWD/raytracing_folder_amd/lib/variants/librt_$name.so RT_GATHER_BLOCKS=$blocks \
    RT_STREAMS=${RT_STREAMS:-1} timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/var_$name_$blocks.log 2>&1
  echo "$name blocks=$blocks: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/var_$name_$blocks.log) $(grep -o 'ms_gather": [0-9.]*' gpurun_out/var_$name_$blocks.log)"
done
