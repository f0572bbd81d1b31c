#!/usr/bin/env python3
"""Tuning aid: the tracer's time by launch class on one workload (first k_wavefront pass | its second pass + the k_bounce launches),
the fullest ray-queue level (rays that did not fit the LDS stacks), per frame with one chunk in flight.
usage: python tools_tracer_split.py [cornell|balls|gi]"""
import os, sys
os.environ["RT_STREAMS"] = "1"
sys.path.insert(0, ".")
import torch
from raytracing_folder_amd import capi, workloads
from raytracing_folder_amd.dist import ShardedRenderer
wl = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if wl == "balls":
    s, cam = workloads.make_balls_scene(1920, 1080)
elif wl == "gi":
    s, cam = workloads.load_cornell_gi(800, 600)
else:
    s, cam = workloads.load_cornell(1920, 1080)
    s.generate_photons(1000000, 8, seed=20171203, device=0)
p = capi.default_params(min_sample=64, max_sample=64, threshold=-1.0)
if wl == "gi":
    p.shade_model, p.bounce, p.hemisphere_sample = capi.SHADE_P12, 8, 1
R = ShardedRenderer(s, cam, p, 0, 1, 0)
R.render_own_tiles()
acc = {}
for _ in range(3):
    d = R.render_own_tiles().as_dict()
    for k in ("ms_primary", "ms_bounce", "ms_gather", "ms_resolve", "ms_total", "peak_rays"):
        acc[k] = acc.get(k, 0.0) + d[k] / 3.0
print(os.environ.get("RT_MI355X_LIB", "shipped").split("librt_")[-1], wl, {k: round(v, 3) for k, v in acc.items()}, flush=True)
