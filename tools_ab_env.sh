#!/bin/bash
# A/B of an environment switch on the three workloads: tools_ab_env.sh <tag> <workloads> <VAR> <value> [<value> ...]
tag=$1; wls=$2; var=$3; shift 3
mkdir -p gpurun_out
for wl in ${wls//,/ }; do
for v in "$@"; do
  env $var=$v timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $wl ${AB_ARGS} 2> gpurun_out/${tag}_${wl}_$v.err | grep '^{' > gpurun_out/${tag}_${wl}_$v.json || { echo "$wl $v FAILED"; tail -5 gpurun_out/${tag}_${wl}_$v.err; exit 1; }
  python - "$wl $var=$v" gpurun_out/${tag}_${wl}_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"]["kernels"]
t = d["traversal_per_frame"]
print(sys.argv[1], "frame_ms", d["ms_per_step"], "gather", k["k_gather"]["ms_per_frame"], "trace", k["k_wavefront+k_bounce"]["ms_per_frame"],
      "nodes", t["bvh_nodes_visited"], "tris", t["tris_tested"], flush=True)
PY
done
done
