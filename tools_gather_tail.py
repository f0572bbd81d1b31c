#!/usr/bin/env python3
"""Tuning aid (needs the gtail instrumentation build, csrc/experiments/r04_gather_tail.patch, selected with RT_MI355X_LIB): when do the
waves of one k_gather launch start and finish?  One rank's share of the bench frame at N = 8 / 4 / 2 is one chunk = one launch.
usage: RT_MI355X_LIB=.../librt_gtail.so python tools_gather_tail.py [json-out]"""
import json
import sys
import torch
sys.path.insert(0, ".")
from raytracing_folder_amd import capi, workloads
from raytracing_folder_amd.dist import ShardedRenderer

s, cam = workloads.load_cornell(1920, 1080)
s.generate_photons(1000000, 8, seed=20171203, device=0)
p = capi.default_params(min_sample=64, max_sample=64, threshold=-1.0)
BIG = 1 << 62
out = {}
for n in (8, 4, 2):
    for rank in (0, n - 1):
        R = ShardedRenderer(s, cam, p, rank, n, 0)
        R.render_own_tiles_packed()
        d = R.render_own_tiles_packed().as_dict()
        tick = 0.01                                         # s_memrealtime: 100 MHz -> microseconds
        last_end, first_end = d["gather_rounds"], BIG - d["gather_slow"]
        first_start, last_start = BIG - d["gather_leaf_reads"], d["photons_visited"]
        waves = 1280 * 4
        span = (last_end - first_start) * tick
        r = {"ms_gather_event": round(d["ms_gather"], 3), "span_us": round(span, 1),
             "last_wave_starts_after_us": round((last_start - first_start) * tick, 1),
             "first_wave_done_before_end_us": round((last_end - first_end) * tick, 1),
             "mean_wave_lifetime_us": round(d["tris_tested"] * tick / waves, 1),
             "busy_fraction_of_span": round(d["tris_tested"] * tick / waves / span, 4),
             "longest_last_batch_us": round(d["bvh_nodes_visited"] * tick, 1),
             "mean_last_batch_us": round(d["instance_visits"] * tick / waves, 1), "queries": d["photon_queries"]}
        out[f"N={n} rank={rank}"] = r
        print(f"N={n} rank={rank}", r, flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

# per-query-round durations (the same build): histogram over one full frame
import ctypes
lib = ctypes.CDLL(__import__("os").environ["RT_MI355X_LIB"])
buf = (ctypes.c_ulonglong * (3 * 2 * 24))()
lib.rtk_exp_hist(buf, 1)
R = ShardedRenderer(s, cam, p, 0, 1, 0)
R.render_own_tiles()
torch.cuda.synchronize()
lib.rtk_exp_hist(buf, 1)
h = list(buf)
rows = []
for path, name in ((0, "plain (sparse cell)"), (1, "normal")):
    for b in range(24):
        n, t, sub = h[(0 * 2 + path) * 24 + b], h[(1 * 2 + path) * 24 + b], h[(2 * 2 + path) * 24 + b]
        if n:
            rows.append({"path": name, "from_us": 0.0 if b == 0 else round((1 << (b + 8)) * 0.01, 2), "rounds": n, "total_ms_of_wave_time": round(t * 1e-5, 2), "mean_sub_leaves": round(sub / n, 1)})
            print(rows[-1], flush=True)
if len(sys.argv) > 1:
    out["query_round_durations_full_frame"] = rows
    json.dump(out, open(sys.argv[1], "w"), indent=1)
