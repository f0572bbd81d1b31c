#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: Mray/s and frame render time of the Cornell
box (FIN shading: direct light + reflection/refraction tree + k=400 photon gather), 1920x1080,
64 samples per pixel fixed, 1M-photon map, on N GPUs of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one whole frame: every rank renders its interleaved 32x8 tiles (rank t mod N) with the
scene and photon map already resident in HBM, then ONE all-gather (RCCL over xGMI) assembles the
RenderImage on every rank.  value = rays actually traced by all ranks / max-over-ranks time.
Prints one JSON line on rank 0 (see the task contract).

roofline: after the timed region the same frame is rendered --profile-frames more times with ONE chunk
in flight (RT_STREAMS=1), so that the HIP-event intervals around each kernel class are exclusive kernel
times; the line is quoted for the kernel that takes the most of them and says which of three ceilings
that kernel sits closest to -- HBM bytes (PMC counters, profiles/), L2-side bytes (what the kernel asks
its caches for), VALU issue slots (SQ counters, profiles/) -- see DESIGN.md section 6.
cpu_baseline: the plain-C oracle timed on a bounded tile sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0        # aggregate L2 rate of the 8 XCDs (MI355X_MICROARCH.md, "L2 (per XCD)")


def measured_peaks():
    """The two ceilings the guide does not give, MEASURED on an MI355X by tools_peaks.hip and committed as profiles/r*_peaks.json
    (the newest one): wave64 v_fma_f32 issue rate of the whole chip (Gwave-inst/s, at the clock the chip holds under that load)
    and 16-byte-per-lane loads that hit the vector L1 (GB/s; 63.4 B per clock and CU at 2.4 GHz), plus the rate of k_gather's own
    access shape (256-byte sub-leaf pairs at its occupancy, random in a 32 KiB window per workgroup: mostly L1 hits).  The
    line's roofline names the file (`peak_source`), so `frac` can be recomputed from that file and the counter summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_peaks.json")))
    if not files:
        return {"source": None, "valu_ginst": 1024 * 2.4 / 2, "l1_gbs": 64.0 * 256 * 2.4, "gather_shape_gbs": None}
    d = json.load(open(files[-1]))
    return {"source": os.path.basename(files[-1]), "valu_ginst": float(d["valu_peak_Gwave_inst_per_s"]), "l1_gbs": float(d["l1_peak_GBps"]),
            "gather_shape_gbs": float(d["gather_shape_5_waves_per_simd"]["window_32KiB_per_workgroup"]["chip_GBps"])}


PEAKS = measured_peaks()
L1_PEAK_GBS = PEAKS["l1_gbs"]
VALU_PEAK_GINST = PEAKS["valu_ginst"]
PARITY_NOTE = ("oracle pinned bit for bit to the reference's own code on the whole path: primitives / BVH / kNN / photon balance (compiled "
               "headers) and TraceNode, GenLight::Shadow, MtlBlinn::Shade (FIN + P13), RenderPixel, PhotonTracing / CausticTracing (the reference's "
               "main.cpp compiled with only its viewport include removed: tests/golden/main_*.npz); GPU vs those fixtures in the -m gpu suite; this "
               "run's timed frame vs the oracle: parity_check")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="wait for every timed frame on the host before the next one is started")
    ap.add_argument("--synthetic-photons", action="store_true", help="wall-sprinkled photons instead of the GPU photon pass")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--profile-frames", type=int, default=2, help="frames of the RT_STREAMS=1 pass behind the timed region (0: no roofline)")
    ap.add_argument("--workload", choices=["cornell", "balls", "gi"], default="cornell",
                    help="cornell = the headline (BASELINE C4); balls = stand-in for the absent christmas_balls scene (C5): "
                         "128 tessellated spheres, 102 402 triangles, half of them mirrors, no photon map -- a BVH-bound "
                         "frame, reported for insight only; gi = BASELINE C3: the Cornell box with the live path-traced GI of "
                         "RayTracingProj12 (its Shade, bounce 8, one hemisphere ray per hit), 800x600, 64 spp, no photon map")
    return ap.parse_args()


PARITY_GATE = {"frac_within_1": 0.995, "z_equal_frac": 0.999}       # >= 99.5 % of pixels within 1 level, z equal


def cpu_baseline(scene_export, balanced, cam, params, budget_s):
    """The oracle (single-threaded plain-C restatement of the reference's RenderPixel) on SEEDED 2x2-pixel blocks spread
    over the SAME frame (seed 7; round 2 timed two 8x8 clusters: a location-dependent number).  Two legs: `value` = the
    work the reference really does (its 30 hemisphere rays per primary hit are traced, shaded and thrown away,
    FIN/main.cpp:642-693); `pruned` = other blocks of the same frame without that dead loop (what the GPU path computes;
    the pixels are the same either way).  The oracle's pixels are KEPT: main() compares them with the last timed frame
    (`parity_check`).  Returns (cpu_baseline dict, [(x0, y0, rgb 2x2x3, z 2x2, count 2x2)])."""
    from oracle import orc
    osc = orc.scene_from_export(scene_export, balanced)
    ocam, op = orc.camera_from(cam), orc.params_from(params)
    bx, by = cam.width // 2, cam.height // 2
    order = np.random.default_rng(7).permutation(bx * by)
    kept = []
    cursor = [0]

    def leg(discarded, budget, min_blocks):
        orc.set_trace_discarded(discarded)
        orc.counters_reset()
        t0 = time.perf_counter()
        per_block = []
        while cursor[0] < len(order):
            b = int(order[cursor[0]])
            cursor[0] += 1
            x0, y0 = (b % bx) * 2, (b // bx) * 2
            tb = time.perf_counter()
            rgb, z, cnt = orc.render(osc, ocam, op, x0, y0, x0 + 2, y0 + 2)
            per_block.append(time.perf_counter() - tb)
            kept.append((x0, y0, rgb[y0:y0 + 2, x0:x0 + 2].copy(), z[y0:y0 + 2, x0:x0 + 2].copy(), cnt[y0:y0 + 2, x0:x0 + 2].copy()))
            if time.perf_counter() - t0 > budget and len(per_block) >= min_blocks:
                break
        dt = time.perf_counter() - t0
        c = orc.counters()
        rays = c["rays_primary"] + c["rays_shadow"] + c["rays_reflect"] + c["rays_refract"] + orc.discarded_rays()
        orc.set_trace_discarded(False)
        blocks = len(per_block)
        px = blocks * 4
        t = np.array(per_block)
        # spread: the blocks are a simple random sample of the frame, so the standard error of the mean block time carries over
        # to the extrapolated frame time (pixel cost varies 100x between a wall and the glass sphere)
        rel_se = float(t.std(ddof=1) / np.sqrt(blocks) / t.mean()) if blocks > 1 else None
        return {"Mray_s": round(rays / dt / 1e6, 4), "px_per_s": round(px / dt, 2), "blocks": blocks, "pixels": px, "seconds": round(dt, 1),
                "rays": int(rays), "frame_s_extrapolated": round(cam.width * cam.height / (px / dt), 1),
                "block_ms": {"mean": round(float(t.mean()) * 1e3, 2), "min": round(float(t.min()) * 1e3, 2), "max": round(float(t.max()) * 1e3, 2),
                             "rel_std_error_of_mean": round(rel_se, 4) if rel_se is not None else None}}
    full = leg(True, budget_s * 0.7, 8)
    pruned = leg(False, budget_s * 0.3, 64)
    return {"value": full["Mray_s"], "unit": "Mray/s", "cores": 1, "kind": "port",
            "port_of": "RenderPixel as the reference executes it: incl. FIN's discarded 30-ray hemisphere loop at every primary hit "
                       "(traced, shaded, dropped: FIN/main.cpp:642-693); same pixels as the GPU path",
            "sample": f"{full['blocks']} seeded 2x2-pixel blocks (seed 7, uniform over the frame) of the same {cam.width}x{cam.height}x{params.max_sample}spp frame, "
                      f"{full['seconds']} s, {full['rays']} rays (the CPU path traces 4 identical shadow rays per light, the GPU 1)",
            "px_per_s": full["px_per_s"], "frame_s_extrapolated": full["frame_s_extrapolated"], "spread": full["block_ms"],
            "speedup_basis": "compare px_per_s (or frame seconds), not Mray/s: the CPU leg counts the 30 discarded hemisphere rays per primary hit",
            "pruned": {**pruned, "note": "the same port with the dead hemisphere loop skipped (what round 1 reported as the baseline)"},
            "reference_measured": {"px_per_s": 159, "Mray_s": 0.135, "threads": 1,
                                   "note": "the reference's own RenderPixel, FIN scene, 1 M photons, adaptive 4->8 spp, rows 400-407, "
                                           "build container (BASELINE.md section 2) -- other sampling, other host: context only"}}, kept


def cpu_worker(path, index, stride, seconds):
    """one worker PROCESS of cpu_baseline_threads (bench.py --cpu-worker ...): the port on its share of the seeded blocks"""
    import pickle
    from oracle import orc
    from raytracing_folder_amd import capi
    d = pickle.load(open(path, "rb"))
    cam, params = capi.Camera.from_buffer_copy(d["cam"]), capi.Params.from_buffer_copy(d["params"])
    osc = orc.scene_from_export(d["export"], d["balanced"])
    ocam, op = orc.camera_from(cam), orc.params_from(params)
    bx, by = cam.width // 2, cam.height // 2
    order = np.random.default_rng(7).permutation(bx * by)
    order = order[len(order) // 2 + index::stride]                                        # (the second half of the seeded order: blocks the single-thread legs do not reach)
    orc.set_trace_discarded(True)
    px, t0 = 0, time.perf_counter()
    for b in order:
        x0, y0 = (int(b) % bx) * 2, (int(b) // bx) * 2
        orc.render(osc, ocam, op, x0, y0, x0 + 2, y0 + 2)
        px += 4
        if time.perf_counter() - t0 > seconds:
            break
    print(json.dumps({"px": px, "seconds": time.perf_counter() - t0}), flush=True)


def cpu_baseline_threads(scene_export, balanced, cam, params, seconds=6.0):
    """The CPU figure AS THE REFERENCE SHIPS IT: BeginRender starts hardware_concurrency() * 2 workers on the shared pixel counter
    (FIN/main.cpp:987, 71-78).  Here: that many worker PROCESSES of the port (its RNG context is a global, so threads of one
    process would race; the workers share nothing but the seeded block order), each on its own blocks of the same frame for
    `seconds`; throughput = all their pixels / the slowest worker's time.  Reported beside cores: 1, never the target."""
    import pickle
    import subprocess
    import tempfile
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(2, min(2 * cores, 32))                                    # (the pool allows a GPU box 16 cores: at most 32 workers)
    td = tempfile.mkdtemp(prefix="rt_cpu_leg_")
    path = os.path.join(td, "leg.pkl")
    with open(path, "wb") as f:
        pickle.dump({"export": scene_export, "balanced": balanced, "cam": bytes(cam), "params": bytes(params)}, f)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(i), str(workers), str(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT) for i in range(workers)]
    outs = []
    for pr in procs:
        o, _ = pr.communicate(timeout=20 * seconds + 120)
        try:
            outs.append(json.loads(o.decode().strip().splitlines()[-1]))
        except Exception:
            pass
    os.remove(path)
    if len(outs) != workers:
        return {"error": f"{workers - len(outs)} of {workers} workers failed"}
    px, wall = sum(o["px"] for o in outs), max(o["seconds"] for o in outs)
    return {"workers": workers, "host_cores": cores, "px_per_s": round(px / wall, 1), "pixels": px, "seconds": round(wall, 1),
            "frame_s_extrapolated": round(cam.width * cam.height / (px / wall), 1),
            "what": "hardware_concurrency() * 2 workers as BeginRender starts them (FIN/main.cpp:987): worker processes of the port, incl. the discarded hemisphere loop"}


def reference_here(raw_photons, width, height, seconds=6.0):
    """The REFERENCE's own RenderPixel (oracle/_ref/ref_main_harness_fin: main.cpp of RayTracingFinal compiled in the build
    container with only its viewport include removed; it travels to the GPU box as a built file) timed on THIS host, one
    thread, on seeded 32-pixel row segments of the same Cornell frame with the same photons -- at the sample counts its
    #defines fix (adaptive 4 -> 8, FIN/main.cpp:19-21: not the 64 spp of the timed workload, which is why `value` stays the
    port's) -- and the port on the very same pixels and settings: how fast the port is relative to the code it restates,
    and that it gives the same pixels.  None when the binary is absent (it needs the reference tree at build time)."""
    import struct
    import subprocess
    import tempfile
    from oracle import orc
    from raytracing_folder_amd import workloads
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_main_harness_fin")
    if not os.path.exists(exe) or raw_photons is None or len(raw_photons) < 2:
        return None
    ph = raw_photons[1:]
    dx = ph["dir_x"].astype(np.float32) / np.float32(0x7FFF)
    dy = ph["dir_y"].astype(np.float32) / np.float32(0x7FFF)
    dz = np.sqrt(np.maximum(0.0, 1.0 - dx.astype(np.float64) ** 2 - dy.astype(np.float64) ** 2)).astype(np.float32)
    dz = np.where(ph["plane_and_dirz"] & 8, -dz, dz)
    power = ph["color"].astype(np.float32) / np.float32(255.0) * ph["power"][:, None]
    payload = np.concatenate([ph["position"], np.stack([dx, dy, dz], 1), power], axis=1).astype(np.float32)
    rng = np.random.default_rng(11)
    n_seg = max(4, int(seconds * 150 / 32))                                 # about 150 px/s on one core with a 1 M-photon map
    rows = np.sort(rng.choice(height, n_seg, replace=False))
    segs = np.array([(int(y) * width + int(rng.integers(0, width - 32)), 32) for y in rows], np.int32)
    data = os.path.dirname(workloads.CORNELL_XML)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", len(payload)) + payload.tobytes() + struct.pack("<iii", width, height, len(segs)) + segs.tobytes())
        try:
            subprocess.run([exe, "pixels", os.path.basename(workloads.CORNELL_XML), fin, fout], check=True, cwd=data, timeout=20 * seconds + 60,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception as e:
            return {"error": f"reference harness failed: {e}"}
        out = open(fout, "rb").read()
    n = struct.unpack_from("<i", out, 0)[0]
    bal = np.frombuffer(out, orc.PHOTON, n + 1, 4).copy()
    off = 4 + (n + 1) * 24 + 8
    px = int(segs[:, 1].sum())
    ref_rgb, ref_z = [], []
    for _, c in segs:
        ref_rgb.append(np.frombuffer(out, np.uint8, 3 * c, off).reshape(c, 3)); off += 3 * c
        ref_z.append(np.frombuffer(out, np.float32, c, off)); off += 4 * c
        off += c
    ref_seconds = struct.unpack_from("<d", out, off)[0]
    # the port on the same pixels, the same (reference-balanced) photons, the reference's settings, its dead hemisphere loop included
    s, cam = workloads.load_cornell(width, height)
    osc = orc.scene_from_export(s.export(), bal)
    ocam, op = orc.camera_from(cam), orc.default_params()
    orc.set_trace_discarded(True)
    t0 = time.perf_counter()
    same = 0
    for (start, c), rr, rz in zip(segs, ref_rgb, ref_z):
        y, x0 = divmod(int(start), width)
        rgb, z, _ = orc.render(osc, ocam, op, x0, y, x0 + int(c), y + 1)
        same += int(((rgb[y, x0:x0 + c] == rr).all(axis=1) & (z[y, x0:x0 + c] == rz)).sum())
    port_seconds = time.perf_counter() - t0
    orc.set_trace_discarded(False)
    return {"what": "the reference's own RenderPixel (adaptive 4->8 spp as its #defines fix) vs the port on the same pixels, photons and settings, this host, 1 thread",
            "pixels": px, "segments": int(len(segs)), "photons": int(n), "reference_px_per_s": round(px / ref_seconds, 1), "port_px_per_s": round(px / port_seconds, 1),
            "port_over_reference_speed": round(ref_seconds / port_seconds, 3), "pixels_identical": same, "reference_seconds": round(ref_seconds, 2)}


def parity_check(kept, frame):
    """The oracle's pixels of the CPU leg against the same pixels of the LAST TIMED frame (SURVEY 8c gate)."""
    rgb, z, cnt = (t.cpu().numpy() for t in frame)
    d, zeq, ceq = [], [], []
    for x0, y0, orgb, oz, ocnt in kept:
        d.append(np.abs(rgb[y0:y0 + 2, x0:x0 + 2].astype(int) - orgb.astype(int)).max(axis=2).ravel())
        zeq.append((z[y0:y0 + 2, x0:x0 + 2] == oz).ravel())
        ceq.append((cnt[y0:y0 + 2, x0:x0 + 2] == ocnt).ravel())
    d, zeq, ceq = np.concatenate(d), np.concatenate(zeq), np.concatenate(ceq)
    out = {"blocks": len(kept), "px": int(d.size), "frac_within_1": round(float((d <= 1).mean()), 5), "max_abs_diff": int(d.max()),
           "frac_equal": round(float((d == 0).mean()), 5), "z_equal_frac": round(float(zeq.mean()), 5), "count_equal_frac": round(float(ceq.mean()), 5),
           "gate": PARITY_GATE, "against": "oracle pixels of the cpu_baseline leg vs the last timed frame"}
    out["pass"] = bool(out["frac_within_1"] >= PARITY_GATE["frac_within_1"] and out["z_equal_frac"] >= PARITY_GATE["z_equal_frac"])
    return out


def profile_figures(workload_tag):
    """What cannot be read from inside bench.py -- HBM bytes (FETCH_SIZE / WRITE_SIZE) and VALU instructions (SQ_INSTS_VALU)
    per launch -- from the NEWEST committed counter summaries (profiles/r*<tag>_bench_pmc_hbm.json, r*<tag>_bench_sq_counters.json:
    separate rocprofv3 --pmc passes of this very command, condensed by tools_profile_summary.py, gfx950 x2 fetch
    correction applied; tag "" = the headline, "_balls" = the 102 k-triangle frame at 256 spp, "_gi" = C3).  Only quoted for the
    exact workload they were measured on (workload_tag None: nothing is quoted), and only for the kernel build they
    were measured on: each summary carries the hash of the kernel sources + build flags of its run (`kernel_source_sha16`);
    when that differs from the build in this tree, or the live launch time of a kernel differs from the file's by more than
    5 %, the figures are reported as STALE instead of quoted (main() then emits frac: null, stale_profile: true)."""
    import glob
    from raytracing_folder_amd import buildinfo
    out = {"hbm": {}, "hbm_launches": {}, "valu_insts": {}, "l1_accesses": {}, "file_launch_us": {}, "hbm_source": None, "valu_source": None,
           "build": buildinfo.kernel_source_sha16(), "stale": []}
    if workload_tag is None:
        return out
    for key, pat in (("hbm", "_bench_pmc_hbm.json"), ("valu", "_bench_sq_counters.json")):
        # tags of the headline workload carry no workload suffix: r03v_bench_..., not r03v_balls_bench_...
        import re
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*" + pat)) if re.match(r"^r\d+[a-z]*" + re.escape(workload_tag + pat) + "$", os.path.basename(f)))
        if not files:
            continue
        try:
            d = json.load(open(files[-1]))
            if d.get("kernel_source_sha16") != out["build"]:
                out["stale"].append(f"{os.path.basename(files[-1])}: collected on kernel build {d.get('kernel_source_sha16')}, this tree is {out['build']}")
            for k, v in d["kernels"].items():
                name = k.split("<")[0]
                if key == "hbm":
                    out["hbm"][name] = float(v["hbm_bytes_per_launch"])
                    out.setdefault("hbm_launches", {})[name] = float(v["FETCH_SIZE"]["launches"])
                elif "SQ_INSTS_VALU" in v and v.get("avg_launch_us"):
                    # wave instructions and microseconds per profiled FRAME (the counter passes profile exactly one frame)
                    out["valu_insts"][name] = float(v["SQ_INSTS_VALU"]) * float(v.get("launches", 1))
                    if "TCP_TOTAL_CACHE_ACCESSES_sum" in v:      # 64-byte accesses of the vector L1s
                        out["l1_accesses"][name] = float(v["TCP_TOTAL_CACHE_ACCESSES_sum"]) * float(v.get("launches", 1))
                    out["file_launch_us"][name] = float(v["avg_launch_us"]) * float(v.get("launches", 1))
            out[key + "_source"] = os.path.basename(files[-1])
        except Exception as e:
            out["stale"].append(f"{os.path.basename(files[-1])}: unreadable ({e})")
    return out


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    a = parse()
    lib_path = os.path.join(ROOT, "raytracing_folder_amd", "lib", "librt_mi355x.so")
    if not os.path.exists(lib_path):              # fresh checkout: built artefacts are git-ignored
        if int(os.environ.get("RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):                  # rank 0 is building
                if os.path.exists(lib_path):
                    break
                time.sleep(1.0)
            time.sleep(2.0)
    import torch
    import torch.distributed as dist
    from raytracing_folder_amd import capi, photons, workloads
    from raytracing_folder_amd.dist import ShardedRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py needs an MI355X (gfx950); the render path has no CPU fallback")
    # rehearsal on a one-GPU box: RT_BENCH_REHEARSAL=1 puts every rank on device 0 and gathers over
    # gloo (RCCL refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    # ---- synthetic inputs, resident in HBM before anything is timed -------------------------
    balanced = None
    if a.workload == "gi" and (a.width, a.height) == (1920, 1080):
        a.width, a.height = 800, 600                       # BASELINE config C3's size unless another one was asked for
    if a.workload == "balls":
        s, cam = workloads.make_balls_scene(a.width, a.height)
    elif a.workload == "gi":
        s, cam = workloads.load_cornell_gi(a.width, a.height)
    else:
        s, cam = workloads.load_cornell(a.width, a.height)
    if a.workload in ("balls", "gi"):
        pass
    elif a.synthetic_photons:
        balanced = photons.synth_cornell_photon_map(a.photons, seed=20171203)
    setup_ms, dump_path = None, None
    if a.workload == "cornell" and not a.synthetic_photons:
        # generatePhotonMap as a whole on the GPU (FIN/main.cpp:350-402; counter RNG, seed 20171203): photon pass ->
        # compaction -> the few photons balancing would put out of LocatePhotons' reach (host) -> gather structure; identical
        # on every rank.  Untimed set-up of the frame metric, reported as setup_ms (the reference's own timer spans it,
        # viewport.cpp:442).  Run twice: the first call also pays the one-time allocations.
        s.generate_photons(a.photons, 8, seed=20171203, device=local)
        setup_ms = s.generate_photons(a.photons, 8, seed=20171203, device=local).as_dict()
        if world == 1 and not a.no_cpu_baseline:
            # once more with the dump generatePhotonMap leaves behind (FIN/main.cpp:397-400): the CPU leg feeds it to the
            # reference's own binary
            import tempfile
            dump_path = os.path.join(tempfile.mkdtemp(prefix="rt_bench_"), "photonmap.dat")
            s.generate_photons(a.photons, 8, seed=20171203, device=local, dat_path=dump_path)
    elif balanced is not None:
        s.set_photons(balanced)
    n_photons = s.counts()["photons"]
    p = capi.default_params(min_sample=a.spp, max_sample=a.spp, threshold=-1.0)
    if a.workload == "gi":
        # RayTracingProj12 main.cpp:17-25 (BOUNCE 8, HEMISPHERE_SAMPLE 1) and its Shade (:341-588)
        p.shade_model, p.bounce, p.hemisphere_sample = capi.SHADE_P12, 8, 1
    R = ShardedRenderer(s, cam, p, rank, world, local, host_gather=rehearsal)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up frames are synchronous (statistics, queue sizes from measurement); the K timed frames are ENQUEUED back to back
    # on the renderer's stream -- render, all-gather and un-interleave of a frame follow each other in stream order, a frame
    # is ordered behind the one before it on the GPU -- with no host round trip in between; finish() waits for all of them
    # and raises if any dropped a ray.  (--sync-steps: every step waits for its frame, as rounds 1 and 2 timed it.)
    st_warm = None
    for _ in range(max(a.warmup, 0)):
        st_warm, _ = R.step()
    if st_warm is not None and not a.sync_steps:
        # untimed: the first ENQUEUED frame of a process pays one-off costs on the host (measured once: 60 ms inside the first
        # asynchronous call on a fresh box) that belong to no step
        R.step(sync=False)
        R.finish()
    barrier()
    async_steps = not a.sync_steps and st_warm is not None
    if async_steps:
        s.render_counters(local, reset=True)            # the device-side work counters now count the timed frames and nothing else
    t0 = time.perf_counter()
    stats, step_ms = [], []
    for _ in range(a.steps):
        t_step = time.perf_counter()
        st, frame = R.step(sync=not async_steps)                             # (without a warm-up the first timed frame is the synchronous one)
        st_warm = st if st is not None else st_warm
        stats.append(st_warm.as_dict())                                      # timings / stream count of the last synchronous frame; the work counts: below
        step_ms.append(round((time.perf_counter() - t_step) * 1e3, 2))      # this rank's steps as the host saw them (enqueue time when asynchronous)
    R.finish()
    barrier()
    dt = time.perf_counter() - t0
    timed_counts = s.render_counters(local).as_dict() if async_steps else None     # ONE read-back behind the timed region: what the K timed frames really traced
    if a.steps > 0:
        frame = tuple(t.clone() for t in frame)        # the last TIMED frame (the roofline leg below renders into the same buffers)

    keys = ["rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "photon_queries", "photons_visited",
            "bvh_nodes_visited", "tris_tested", "instance_visits", "gather_rounds", "gather_slow", "gather_leaf_reads"]
    tot = {k: float(timed_counts[k]) if timed_counts is not None else float(sum(x[k] for x in stats)) for k in keys}
    gather_ms = float(sum(R.gather_ms)) / max(len(R.gather_ms), 1) if getattr(R, "gather_ms", None) else 0.0
    vec = torch.tensor([dt, gather_ms] + [tot[k] for k in keys], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if world > 1:
        mx = vec.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dt, gather_ms = float(mx[0]), float(mx[1])
        for i, k in enumerate(keys):
            tot[k] = float(vec[2 + i])
    rays = tot["rays_primary"] + tot["rays_shadow"] + tot["rays_reflect"] + tot["rays_refract"]

    # ---- roofline leg: the same frame with ONE chunk in flight, so that event intervals are exclusive kernel times.
    # Every rank takes part (same number of renders everywhere); only rank 0's figures are quoted.
    prof = []
    if a.profile_frames > 0:
        old_streams = os.environ.get("RT_STREAMS")
        os.environ["RT_STREAMS"] = "1"
        for _ in range(a.profile_frames):
            prof.append(R.render_own_tiles(want_stats=True).as_dict())
        if old_streams is None:
            del os.environ["RT_STREAMS"]
        else:
            os.environ["RT_STREAMS"] = old_streams
        barrier()

    if rank == 0:
        roof = None
        if prof:
            assert all(x["streams"] == 1 for x in prof)
            n = float(len(prof))
            P = {k: sum(float(x[k]) for x in prof) / n for k in prof[0]}          # per-frame averages of rank 0's share
            # the counter summaries under profiles/ belong to three exact commands: the headline, the 102 k-triangle frame at its
            # stated 256 spp, and C3
            prof_tag = None
            if world == 1 and not a.synthetic_photons:
                if (a.workload, a.width, a.height, a.spp, a.photons) == ("cornell", 1920, 1080, 64, 1000000): prof_tag = ""
                elif (a.workload, a.width, a.height, a.spp) == ("balls", 1920, 1080, 256): prof_tag = "_balls"
                elif (a.workload, a.width, a.height, a.spp) == ("gi", 800, 600, 64): prof_tag = "_gi"
            figs = profile_figures(prof_tag)
            p_rays = P["rays_primary"] + P["rays_shadow"] + P["rays_reflect"] + P["rays_refract"]
            classes = {
                # algorithmic bytes: SURVEY.md section 8(d) / BASELINE.md section 3.4; l2 bytes: what the kernel asks its
                # caches for in the DEVICE layout (rt_dev.h): 32-B photon slots, 64-B BVH nodes, 48-B triangles, 96-B node
                # transforms, 64-B ray-queue records (written once, read once), 17 B/sample + 8 B/pixel in the resolve
                "k_gather": {"ms": P["ms_gather"], "launches": P["launches_gather"],
                             "alg": P["photon_queries"] * 48.0 + P["photons_visited"] * 24.0,
                             "l2": P["photon_queries"] * 48.0 + P["gather_leaf_reads"] * 32.0 * 32.0},      # slots read, counted in units of 32 slots of 32 B
                # the tracer: k_wavefront (whole ray tree, LDS ray stacks) + the k_bounce launches behind it (rays that did not
                # fit the LDS stacks: normally none); with RT_TRACER=levels: k_primary + one k_bounce per level
                "k_wavefront+k_bounce": {"ms": P["ms_primary"] + P["ms_bounce"], "launches": P["launches_primary"] + P["launches_bounce"],
                                       "alg": p_rays * 48.0 + P["instance_visits"] * 84.0 + P["bvh_nodes_visited"] * 28.0 + P["tris_tested"] * 48.0,
                                       "l2": (P["rays_reflect"] + P["rays_refract"]) * 128.0 + P["instance_visits"] * 96.0 +
                                             P["bvh_nodes_visited"] * 64.0 + P["tris_tested"] * 48.0 + P["photon_queries"] * 48.0 + P["samples"] * 17.0},
                "k_resolve": {"ms": P["ms_resolve"], "launches": P["launches_resolve"],
                              "alg": P["pixels"] * 8.0, "l2": P["samples"] * 17.0 + P["pixels"] * 8.0},
            }
            exclusive_ms = sum(c["ms"] for c in classes.values())
            table = {}
            for name, c in classes.items():
                sec = c["ms"] * 1e-3
                row = {"ms_per_frame": round(c["ms"], 3), "launches_per_frame": round(c["launches"], 1),
                       "share_of_kernel_time": round(c["ms"] / exclusive_ms, 3) if exclusive_ms > 0 else None,
                       "algorithmic_GBps": round(c["alg"] / sec / 1e9, 1) if sec > 0 else None,
                       "l2_GBps": round(c["l2"] / sec / 1e9, 1) if sec > 0 else None,
                       "l2_frac": round(c["l2"] / sec / 1e9 / L2_PEAK_GBS, 4) if sec > 0 else None}
                parts = name.split("+")
                if "k_wavefront" in parts and "k_wavefront" not in figs["hbm"] and "k_primary" in figs["hbm"]:
                    parts = ["k_primary" if q == "k_wavefront" else q for q in parts]
                if all(q in figs["hbm"] for q in parts) and sec > 0:
                    # PMC bytes are per launch of each kernel; the counter passes profile exactly ONE frame of this workload, so a
                    # kernel's launches in them are its launches per frame (k_wavefront's second pass is a launch of its own there,
                    # while the library's event classes count it with the level launches)
                    hb = sum(figs["hbm"][q] * figs["hbm_launches"][q] for q in parts)
                    row["hbm_bytes_per_frame"] = int(hb)
                    row["hbm_frac"] = round(hb / sec / 1e9 / HBM_PEAK_GBS, 4)
                q0 = parts[0]
                if q0 in figs["valu_insts"] and sec > 0 and c["launches"] > 0:
                    # VALU issue fraction of the DOMINANT kernel of the class: counted instructions per launch (profile) / LIVE
                    # exclusive time per launch, against the chip's measured v_fma_f32 issue rate (measured_peaks()); NOT clamped.
                    live_us = c["ms"] * 1e3                                     # this class, per frame, exclusive
                    row["valu_frac"] = round(figs["valu_insts"][q0] / (live_us * 1e3) / VALU_PEAK_GINST, 4)
                    if q0 in figs["l1_accesses"]:
                        # the vector-memory pipe: 64-byte L1 accesses counted in the profile / live time, against the measured rate of L1 hits;
                        # for k_gather also against the measured rate of its own access shape (what loads alone can do at its occupancy)
                        row["l1_frac"] = round(figs["l1_accesses"][q0] * 64.0 / (live_us * 1e-6) / 1e9 / L1_PEAK_GBS, 4)
                        if name == "k_gather" and PEAKS["gather_shape_gbs"]:
                            row["access_shape_frac"] = round(figs["l1_accesses"][q0] * 64.0 / (live_us * 1e-6) / 1e9 / PEAKS["gather_shape_gbs"], 4)
                    row["live_us_per_frame"] = round(live_us, 1)
                    row["profile_us_per_frame"] = round(figs["file_launch_us"][q0], 1)
                    # (the tracer class also holds the k_bounce launches behind k_wavefront: compared with a wider margin)
                    margin = 0.05 if name != "k_wavefront+k_bounce" else 0.10
                    if abs(live_us - figs["file_launch_us"][q0]) > margin * figs["file_launch_us"][q0]:
                        figs["stale"].append(f"{q0}: live {live_us:.0f} us per frame vs {figs['file_launch_us'][q0]:.0f} us in {figs['valu_source']}")
                table[name] = row
            dom = max(classes, key=lambda k: classes[k]["ms"])
            d, c = table[dom], classes[dom]
            fr = {"hbm": d.get("hbm_frac"), "l2": d["l2_frac"], "valu": d.get("valu_frac"), "l1": d.get("l1_frac")}
            bound = max((k for k in fr if fr[k] is not None), key=lambda k: fr[k])
            launches = max(c["launches"], 1.0)
            sec_launch = c["ms"] * 1e-3 / launches
            if bound == "hbm":
                ach, peak, unit = d["hbm_bytes_per_frame"] / (c["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            elif bound == "l2":
                ach, peak, unit = d["l2_GBps"], L2_PEAK_GBS, "GB/s"
            elif bound == "l1":
                ach, peak, unit = fr["l1"] * L1_PEAK_GBS, L1_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = fr["valu"] * VALU_PEAK_GINST, VALU_PEAK_GINST, "Gwave-inst/s"
            stale = bool(figs["stale"])
            if stale:
                # counter-derived fractions belong to other kernels than the ones that just ran: not quoted
                fr = {"hbm": None, "l2": fr["l2"], "valu": None, "l1": None}
            roof = {"kernel": dom, "bound": bound if not stale else "l2", "achieved": round(ach, 2) if not stale else d["l2_GBps"],
                    "peak": round(peak, 1) if not stale else L2_PEAK_GBS, "unit": unit if not stale else "GB/s",
                    "frac": round(fr[bound], 4) if not stale else None, "stale_profile": stale, "stale_reasons": figs["stale"],
                    "kernel_build": figs["build"], "peak_source": PEAKS["source"],
                    "peaks": {"hbm_GBps": HBM_PEAK_GBS, "l2_GBps": L2_PEAK_GBS, "l1_GBps": L1_PEAK_GBS, "valu_Gwave_inst_per_s": VALU_PEAK_GINST,
                              "gather_access_shape_GBps": PEAKS["gather_shape_gbs"],
                              "note": "hbm, l2: MI355X_MICROARCH.md; l1, valu, gather_access_shape: measured by tools_peaks.hip (peak_source)"},
                    "traffic": int(figs["hbm"][dom]) if dom in figs["hbm"] and not stale else None,
                    "traffic_source": figs["hbm_source"], "valu_source": figs["valu_source"],
                    "fractions": fr,
                    "avg_launch_ms": round(sec_launch * 1e3, 4), "launches_per_frame": round(launches, 1),
                    "timing": f"exclusive: {len(prof)} frame(s) with one chunk in flight (RT_STREAMS=1) after the timed region, HIP events on the "
                              f"launch stream; the timed region ran with {int(stats[-1]['streams'])} chunk(s) in flight",
                    "exclusive_kernel_ms_per_frame": round(exclusive_ms, 2), "frame_ms_one_stream": round(P["ms_total"], 2),
                    "algorithmic": {"bytes_per_launch": int(c["alg"] / launches), "GBps": d["algorithmic_GBps"],
                                    "frac_of_hbm_peak": round(d["algorithmic_GBps"] / HBM_PEAK_GBS, 3),
                                    "note": "SURVEY 8(d) formula (48 B/query + 24 B/photon examined; 48 B/ray + 84 B/instance + 28 B/node + 48 B/triangle): "
                                            "it prices cache hits as bytes, so it is NOT an HBM rate -- the photon structure (34 MB) and the scene are "
                                            "L2 / Infinity-Cache resident; see hbm_frac for what really crosses the fabric"},
                    "kernels": table}
        out = {
            "metric": "Mray/s", "value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C5 stand-in: 128 tessellated spheres (102 402 triangles, half mirrors) on a ground quad under a PNG sky "
                                    f"(environment + background), FIN shading, {a.width}x{a.height}, {a.spp} spp fixed, no photon map, bounce 4") if a.workload == "balls" else
                                   (f"C3: the Cornell box RayTracingProj12's main() loads (glass teapot, glossy sphere), its shading (live path-traced GI: one cosine-hemisphere ray per hit), {a.width}x{a.height}, "
                                    f"{a.spp} spp fixed, bounce 8, no photon map") if a.workload == "gi" else
                                   f"Cornell box, FIN shading, {a.width}x{a.height}, {a.spp} spp fixed, "
                                   f"{n_photons}-photon map ({'sprinkled on the walls' if a.synthetic_photons else 'GPU photon pass, Philox seed 20171203, 8 bounces'}), "
                                   f"k=400 r=1, bounce 4",
                       "tiles": "32x8 interleaved, tile t -> rank t mod N", "exchange": "one all_gather of 8 B/pixel per frame"},
            "frame_ms": round(dt / a.steps * 1e3, 2),
            "render_attempts": int(max(x.get("attempts", 1) for x in stats)) if stats else None,
            "setup_ms": setup_ms,
            "step_ms_rank0": step_ms,
            "gather_ms": round(gather_ms, 3),
            "parity_note": PARITY_NOTE,
            **({"rehearsal": "all ranks on device 0, gloo gather -- not a scaling measurement"} if rehearsal else {}),
            "ray_counts_from": "device counters read once after the timed frames (rt_render_counters)" if timed_counts is not None else "per-step statistics of the synchronous steps",
            "rays_per_frame": {k: int(tot[k] / a.steps) for k in keys[:4]},
            "photon_queries_per_frame": int(tot["photon_queries"] / a.steps),
            "gather_per_frame": {k: int(tot[k] / a.steps) for k in ("photons_visited", "gather_rounds", "gather_slow", "gather_leaf_reads")},
            "traversal_per_frame": {k: int(tot[k] / a.steps) for k in ("instance_visits", "bvh_nodes_visited", "tris_tested")},
            "roofline": roof,
        }
        failed = False
        if world == 1 and not a.no_cpu_baseline:
            # the checker needs the reference's balanced heap (its LocatePhotons walks it): made from the same photons on request
            out["cpu_baseline"], kept = cpu_baseline(s.export(), s.get_photons() if n_photons else None, cam, p, a.cpu_seconds)
            out["cpu_baseline"]["as_shipped_multi_worker"] = cpu_baseline_threads(s.export(), s.get_photons() if n_photons else None, cam, p)
            if a.workload == "cornell" and n_photons:
                # (the unbalanced photons as generated: what balancing them on the host starts from)
                raw = capi.photons_read_dat(dump_path) if dump_path else None
                out["cpu_baseline"]["reference_here"] = reference_here(raw, a.width, a.height)
            # the frame that was timed is the frame that is checked: the oracle's pixels against the last timed step's image
            out["parity_check"] = parity_check(kept, frame)
            failed = not out["parity_check"]["pass"]
        if world > 1:
            # the gathered frame must equal what the tiles say: spot-check against rank 0's own tiles
            frgb, fz, fcnt = frame
            own = R.rgb.cpu() if rehearsal else R.rgb
            assert bool((frgb[:8, :32].cpu() == own[:8, :32].cpu()).all()), "gathered frame disagrees with rank 0's tile"
            out["gathered_frame_nonzero_fraction"] = round(float((fz.float() != 0).float().mean()), 4)
            if rehearsal:
                # every rank sits on this one GPU: render the N = 1 frame here and hold the gathered one against it
                R1 = ShardedRenderer(s, cam, p, 0, 1, local)
                R1.step()
                one = tuple(t.cpu() for t in (R1.rgb, R1.z, R1.cnt))
                g = tuple(t.cpu() for t in frame)
                drgb = (g[0].int() - one[0].int()).abs()
                out["rehearsal_check"] = {"z_equal": bool((g[1] == one[1]).all()), "count_equal": bool((g[2] == one[2]).all()),
                                          "rgb_max_abs_diff": int(drgb.max()), "against": "the N = 1 frame rendered on the same device"}
                assert out["rehearsal_check"]["z_equal"] and out["rehearsal_check"]["count_equal"] and out["rehearsal_check"]["rgb_max_abs_diff"] <= 1, out["rehearsal_check"]
        print(json.dumps(out), flush=True)
        if failed:
            sys.exit("bench.py: the timed frame FAILS the parity gate against the oracle (see parity_check in the line above)")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
