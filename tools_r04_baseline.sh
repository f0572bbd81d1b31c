#!/bin/bash
# round-4 first call: the measured ceilings + today's numbers of the shipped kernels on the three workloads + the share timing
mkdir -p gpurun_out
timeout -k 10 120 raytracing_folder_amd/lib/tools_peaks > gpurun_out/r04_peaks.json 2> gpurun_out/r04_peaks.err || { echo peaks FAILED; tail -3 gpurun_out/r04_peaks.err; exit 1; }
cat gpurun_out/r04_peaks.json
for wl in cornell balls gi; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $wl 2> gpurun_out/r04a_$wl.err | grep '^{' > gpurun_out/r04a_$wl.json || { echo "$wl FAILED"; tail -5 gpurun_out/r04a_$wl.err; exit 1; }
  python - gpurun_out/r04a_$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels"]
print(sys.argv[1], "frame_ms", d["ms_per_step"], "Mray/s", d["value"], {n: v["ms_per_frame"] for n, v in k.items()}, d["traversal_per_frame"], d["rays_per_frame"], flush=True)
PY
done
timeout -k 10 300 python tools_share_timing.py cornell gpurun_out/r04a_share.json 2>&1 | tail -12
