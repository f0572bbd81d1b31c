#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag> directory (tools_profile_run.sh: rocprofv3 --kernel-trace --stats and the --pmc passes of
bench.py) into the files kept under profiles/:
    <tag>_bench_kernel_stats.csv    rocprofv3's own per-kernel statistics of the --stats pass
    <tag>_bench_pmc_hbm.json        FETCH_SIZE / WRITE_SIZE per kernel and launch (gfx950: fetch side doubled)
    <tag>_bench_sq_counters.json    SQ / TCC / TCP counters per kernel and launch + the average launch time of the same pass
usage: python tools_profile_summary.py gpurun_out/prof_r2a r02a "<workload note>" """
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from raytracing_folder_amd import buildinfo      # noqa: E402

src, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
# which kernels these counters belong to: bench.py refuses to quote them for another build.  The profile run leaves the id of
# the build it ran on the GPU box in <src>/build_id.json (tools_profile_run.sh); the working tree's is the fallback.
BUILD = {"kernel_source_sha16": buildinfo.kernel_source_sha16(), "build_flags": buildinfo.build_flags()}
try:
    BUILD = json.load(open(os.path.join(src, "build_id.json")))
except Exception:
    pass


def kname(n):
    k = n.split("(")[0].replace("void ", "").strip()
    return k if k.startswith("k_") else None


def counters(passname):
    """{kernel: {counter: sum}}, {kernel: launches}, {kernel: avg_us} from one pmc pass"""
    files = glob.glob(f"{src}/{passname}/**/*counter_collection.csv", recursive=True)
    if not files:
        return None
    agg, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = kname(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{src}/{passname}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg, {k: len(v) for k, v in disp.items()}, {k: sum(v) / len(v) for k, v in dur.items() if v}


os.makedirs("profiles", exist_ok=True)
st = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], f"profiles/{tag}_bench_kernel_stats.csv")
    print(open(st[0]).read()[:1200])

hbm = {}
for name, passname in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    c = counters(passname)
    if not c:
        continue
    agg, n, _ = c
    for k in agg:
        hbm.setdefault(k, {})[name] = {"launches": n[k], "sum_KiB": round(agg[k][name], 1), "per_launch_KiB": round(agg[k][name] / n[k], 1)}
if hbm:
    for k, d in hbm.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE"]["per_launch_KiB"] + d["WRITE_SIZE"]["per_launch_KiB"]) * 1024)
    json.dump({"command": "tools_profile_run.sh: rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                          "--no-cpu-baseline --profile-frames 0 (one pass per counter)",
               "workload": note, **BUILD, "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
               "note": "gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section); "
                       "hbm_bytes_per_launch applies that x2 to the fetch side",
               "kernels": hbm}, open(f"profiles/{tag}_bench_pmc_hbm.json", "w"), indent=1)
    print({k: v.get("hbm_bytes_per_launch") for k, v in hbm.items()})

sq = {}
for passname in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_l2", "pmc_ta1", "pmc_ta2"):
    c = counters(passname)
    if not c:
        continue
    agg, n, avg = c
    for k in agg:
        d = sq.setdefault(k, {"launches": n[k]})
        for cname, v in agg[k].items():
            d[cname] = round(v / n[k], 1)
        if k in avg:
            d.setdefault("avg_launch_us", round(avg[k], 1))
if sq:
    for k, d in sq.items():
        if "SQ_INSTS_VALU" in d and "avg_launch_us" in d:
            # A SIMD issues one wave64 VALU instruction per 2 cycles (32 lanes per cycle: MI355X_MICROARCH.md; the 157.3 TF f32
            # vector peak is 1024 SIMDs x 2.4 GHz x 32 lanes x 2 flop) -- a SINGLE wave can issue one only every 4 cycles, and
            # SQ_ACTIVE_INST_VALU counts those 4 (one quad-cycle per instruction), which is why this fraction was priced at 4
            # cycles for a while in round 2: the build without packed ops then came out at 1.17 of that "peak".  Packed f32 ops
            # (v_pk_*) hold the pipe twice as long; they are counted once here.
            d["valu_issue_frac_of_peak"] = round(d["SQ_INSTS_VALU"] * 2 / (1024 * 2.4e3 * d["avg_launch_us"]), 4)
            # ... and against what the chip was MEASURED to issue (tools_peaks.hip -> profiles/r*_peaks.json, newest), under the profiler's clock
            pk = sorted(glob.glob("profiles/r[0-9]*_peaks.json"))
            if pk:
                d["valu_frac_of_measured_peak"] = round(d["SQ_INSTS_VALU"] / (d["avg_launch_us"] * 1e3) / json.load(open(pk[-1]))["valu_peak_Gwave_inst_per_s"], 4)
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
            d["wait_any_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        if "TCC_HIT_sum" in d and d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0) > 0:
            d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
    json.dump({"command": "tools_profile_run.sh: rocprofv3 --kernel-trace --pmc <4 counters per pass> --output-format csv -- python3 bench.py --steps 1 "
                          "--warmup 0 --no-cpu-baseline --profile-frames 0",
               "workload": note, **BUILD,
               "units": "per-launch averages; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are per-wave quad-cycles summed over all waves, SQ_INSTS_* wave "
                        "instructions; avg_launch_us from the kernel trace of the same pass (kernels of up to three chunks overlap in it)",
               "kernels": sq}, open(f"profiles/{tag}_bench_sq_counters.json", "w"), indent=1)
    for k, d in sq.items():
        print(k, {x: d[x] for x in d if x in ("launches", "avg_launch_us", "SQ_INSTS_VALU", "valu_issue_frac_of_peak", "wait_any_frac", "l2_hit_rate", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU")})
