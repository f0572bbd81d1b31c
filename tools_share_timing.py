#!/usr/bin/env python3
"""Tuning aid: time one rank's share of the bench frame (tiles rank, rank+N, ...) on ONE GPU for N = 1, 2, 4, 8 --
what a rank of an N-GPU run does per frame, without the gather.  Shows the fixed per-frame cost that limits strong scaling.
Two ways: every frame waited for on the host (as rounds 1 and 2 timed a step), and K frames enqueued back to back on the
renderer's stream (how bench.py times its steps since round 3).
usage: python tools_share_timing.py [workload] [json-out]"""
import json
import sys
import time
import torch
sys.path.insert(0, ".")
from raytracing_folder_amd import capi
from raytracing_folder_amd.dist import ShardedRenderer
from raytracing_folder_amd import workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if wl == "balls":
    s, cam = workloads.make_balls_scene(1920, 1080)
else:
    s, cam = workloads.load_cornell(1920, 1080)
    s.generate_photons(1000000, 8, seed=20171203, device=0)
p = capi.default_params(min_sample=64, max_sample=64, threshold=-1.0)
out = {"what": "tools_share_timing.py: one rank's tiles of the bench frame on one GPU, slowest of ranks {0, N/2, N-1}, no exchange; "
               "sync = every frame waited for on the host, async = 6 frames enqueued back to back (bench.py's timed loop)", "workload": wl,
       "ms_per_frame_sync": {}, "ms_per_frame_async": {}, "kernels_last_rank_ms": {}}
base = {}
for n in (1, 2, 4, 8):
    worst = {"sync": 0.0, "async": 0.0}
    for rank in sorted({0, n // 2, n - 1}):
        R = ShardedRenderer(s, cam, p, rank, n, 0)
        st = R.render_own_tiles_packed() if n > 1 else R.render_own_tiles()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st = R.render_own_tiles_packed() if n > 1 else R.render_own_tiles()
        torch.cuda.synchronize()
        worst["sync"] = max(worst["sync"], (time.perf_counter() - t0) / 3 * 1e3)
        t0 = time.perf_counter()
        for _ in range(6):
            (R.render_own_tiles_packed if n > 1 else R.render_own_tiles)(want_stats=False, sync=False)
        R.finish()
        worst["async"] = max(worst["async"], (time.perf_counter() - t0) / 6 * 1e3)
    d = st.as_dict()
    out["kernels_last_rank_ms"][str(n)] = {k: round(d[k], 3) for k in ("ms_primary", "ms_bounce", "ms_gather", "ms_resolve")}
    for mode in ("sync", "async"):
        base.setdefault(mode, worst[mode])
        out["ms_per_frame_" + mode][str(n)] = round(worst[mode], 2)
        print(f"{wl} N={n} {mode}: slowest sampled rank {worst[mode]:.2f} ms/frame -> efficiency bound {base[mode] / (n * worst[mode]):.3f}", flush=True)
    print("   last rank's kernels (ms, events, synchronous frame):", out["kernels_last_rank_ms"][str(n)], flush=True)
out["efficiency_bound_sync"] = {k: round(out["ms_per_frame_sync"]["1"] / (int(k) * v), 3) for k, v in out["ms_per_frame_sync"].items()}
out["efficiency_bound_async"] = {k: round(out["ms_per_frame_async"]["1"] / (int(k) * v), 3) for k, v in out["ms_per_frame_async"].items()}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
