#!/bin/bash
# A/B of tuning builds on the GPU box: tools_ab.sh <tag> <name> [<name> ...]   (name "shipped" = the library as built;
# others = raytracing_folder_amd/lib/variants/librt_<name>.so from `make -C raytracing_folder_amd/csrc variant NAME=.. DEFS=..`)
# writes gpurun_out/<tag>_<name>.json (the bench line) and prints one summary line per build
tag=$1; shift
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = shipped ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$PWD/raytracing_folder_amd/lib/variants/librt_$n.so; fi
  timeout -k 10 240 python bench.py --steps 5 --warmup 1 --no-cpu-baseline ${AB_ARGS} 2> gpurun_out/${tag}_$n.err | grep '^{' > gpurun_out/${tag}_$n.json || { echo "$n FAILED"; tail -3 gpurun_out/${tag}_$n.err; exit 1; }
  python - "$n" gpurun_out/${tag}_$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
print(sys.argv[1], "frame_ms", d["ms_per_step"], "gather", k["k_gather"]["ms_per_frame"], "trace", k["k_wavefront+k_bounce"]["ms_per_frame"],
      "resolve", k["k_resolve"]["ms_per_frame"], "leaf_reads", d["gather_per_frame"]["gather_leaf_reads"], "rounds", d["gather_per_frame"]["gather_rounds"], flush=True)
PY
done
