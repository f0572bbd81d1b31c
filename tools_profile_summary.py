#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats and the two --pmc passes of
bench.py) into the files kept under profiles/: <tag>_bench_kernel_stats.csv and <tag>_bench_pmc_hbm.json.
usage: python tools_profile_summary.py gpurun_out/prof_r1b r01b"""
import collections
import csv
import glob
import json
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{src}/stats/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_bench_kernel_stats.csv")
out = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu-baseline (one pass per counter; stats pass: --kernel-trace --stats, --steps 2 --warmup 1)",
       "workload": "Cornell 1920x1080x64spp, 1M-photon map from the GPU photon pass, 16 chunks of 8 Mi samples per frame (32 of 4 Mi before r01d)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
       "note": "gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM "
               "section); hbm_bytes_per_launch applies that x2 to the fetch side",
       "kernels": {}}
data = {}
for name, pat in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = glob.glob(f"{src}/{pat}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        k = k.split("<")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        data.setdefault(k, {})[name] = {"launches": n, "sum_KiB": round(v, 1), "per_launch_KiB": round(v / n, 1)}
for k, d in data.items():
    d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE"]["per_launch_KiB"] + d["WRITE_SIZE"]["per_launch_KiB"]) * 1024)
out["kernels"] = data
json.dump(out, open(f"profiles/{tag}_bench_pmc_hbm.json", "w"), indent=1)
print(open(f"profiles/{tag}_bench_kernel_stats.csv").read()[:900])
print({k: v["hbm_bytes_per_launch"] for k, v in data.items()})
