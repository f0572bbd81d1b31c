#!/usr/bin/env python3
"""Tuning aid: time one rank's share of the bench frame (tiles rank, rank+N, ...) on ONE GPU for N = 1, 2, 4, 8 --
what a rank of an N-GPU run does per frame, without the gather.  Shows the fixed per-frame cost that
limits strong scaling.  usage: python tools_share_timing.py [workload]"""
import sys
import time
import torch
sys.path.insert(0, ".")
import bench
from raytracing_folder_amd import capi
from raytracing_folder_amd.dist import ShardedRenderer
from raytracing_folder_amd import workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if wl == "balls":
    s, cam = workloads.make_balls_scene(1920, 1080)
else:
    s, cam = workloads.load_cornell(1920, 1080)
    raw, _ = s.photon_pass(1000000, 8, seed=20171203, device=0)
    s.set_photons(capi.photon_balance(raw))
p = capi.default_params(min_sample=64, max_sample=64, threshold=-1.0)
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, n // 2, n - 1}):
        R = ShardedRenderer(s, cam, p, rank, n, 0)
        R.render_own_tiles()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st = R.render_own_tiles()
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / 3 * 1e3)
    base = base or worst
    d = st.as_dict()
    print("   last rank's kernels (ms, events):", {k: round(d[k], 3) for k in ("ms_primary", "ms_bounce", "ms_gather", "ms_resolve")},
          "launches", {k: d[k] for k in ("launches_primary", "launches_bounce", "launches_gather", "launches_resolve")}, flush=True)
    print(f"{wl} N={n}: slowest sampled rank {worst:.2f} ms/frame -> efficiency bound {base / (n * worst):.3f}", flush=True)
