#!/bin/bash
# A/B of tuning builds on the three workloads: tools_ab3.sh <tag> <workloads, comma separated> <name> [<name> ...]
# (name "shipped" = the library as built; others = raytracing_folder_amd/lib/variants/librt_<name>.so)
tag=$1; wls=$2; shift 2
mkdir -p gpurun_out
for wl in ${wls//,/ }; do
for n in "$@"; do
  if [ "$n" = shipped ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$PWD/raytracing_folder_amd/lib/variants/librt_$n.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $wl ${AB_ARGS} 2> gpurun_out/${tag}_${wl}_$n.err | grep '^{' > gpurun_out/${tag}_${wl}_$n.json || { echo "$wl $n FAILED"; tail -5 gpurun_out/${tag}_${wl}_$n.err; exit 1; }
  python - "$wl $n" gpurun_out/${tag}_${wl}_$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
t = d["traversal_per_frame"]
print(sys.argv[1], "frame_ms", d["ms_per_step"], "gather", k["k_gather"]["ms_per_frame"], "trace", k["k_wavefront+k_bounce"]["ms_per_frame"],
      "resolve", k["k_resolve"]["ms_per_frame"], "nodes", t["bvh_nodes_visited"], "tris", t["tris_tested"], "inst", t["instance_visits"], flush=True)
PY
done
done
