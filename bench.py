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
Prints one JSON line on rank 0 (see the task contract): roofline = the dominant kernel (photon
gather), cpu_baseline = the plain-C oracle timed on a bounded tile sample (rank 0, N=1 only).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--synthetic-photons", action="store_true", help="wall-sprinkled photons instead of the GPU photon pass")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--workload", choices=["cornell", "balls"], default="cornell",
                    help="cornell = the headline (BASELINE C4); balls = stand-in for the absent christmas_balls scene (C5): "
                         "128 tessellated spheres, 102 402 triangles, half of them mirrors, no photon map -- a BVH-bound "
                         "frame, reported for insight only")
    return ap.parse_args()


def cpu_baseline(scene_export, balanced, cam, params, budget_s):
    """The oracle (single-threaded plain-C restatement of the reference's RenderPixel) on a seeded
    sample of 8x8-pixel blocks of the SAME workload, for about budget_s seconds."""
    from oracle import orc
    osc = orc.scene_from_export(scene_export, balanced)
    ocam, op = orc.camera_from(cam), orc.params_from(params)
    rng = np.random.default_rng(7)
    bx, by = cam.width // 8, cam.height // 8
    order = rng.permutation(bx * by)
    orc.counters_reset()
    t0 = time.perf_counter()
    blocks = 0
    for b in order:
        x0, y0 = (b % bx) * 8, (b // bx) * 8
        orc.render(osc, ocam, op, x0, y0, x0 + 8, y0 + 8)
        blocks += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    c = orc.counters()
    rays = c["rays_primary"] + c["rays_shadow"] + c["rays_reflect"] + c["rays_refract"]
    px = blocks * 64
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mray/s", "cores": 1, "kind": "port",
            "sample": f"{blocks} seeded 8x8-pixel blocks ({px} px) of the same {cam.width}x{cam.height}x{params.max_sample}spp frame, "
                      f"{dt:.1f} s, {rays} rays (the CPU path traces 4 identical shadow rays per light, the GPU 1)",
            "px_per_s": round(px / dt, 2),
            "frame_s_extrapolated": round(cam.width * cam.height / (px / dt), 1)}


def measured_traffic(kernel, default_workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_bench_pmc_hbm.json,
    made by tools_profile_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    command, with the gfx950 x2 fetch correction).  PMC counters cannot be read from inside bench.py, so the
    figure is only quoted for the default workload it was measured on; otherwise null."""
    import glob
    if not default_workload:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_hbm.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return float(d["kernels"][kernel]["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception:
        return None, None


def main():
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
    if a.workload == "balls":
        s, cam = workloads.make_balls_scene(a.width, a.height)
    else:
        s, cam = workloads.load_cornell(a.width, a.height)
    if a.workload == "balls":
        pass
    elif a.synthetic_photons:
        balanced = photons.synth_cornell_photon_map(a.photons, seed=20171203)
    else:
        # generatePhotonMap on the GPU (counter RNG, seed 20171203), balanced on the host like the
        # reference does; identical on every rank
        raw, attempts = s.photon_pass(a.photons, 8, seed=20171203, device=local)
        balanced = capi.photon_balance(raw)
    if balanced is not None:
        s.set_photons(balanced)
    p = capi.default_params(min_sample=a.spp, max_sample=a.spp, threshold=-1.0)
    R = ShardedRenderer(s, cam, p, rank, world, local, host_gather=rehearsal)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 0)):
        R.step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(a.steps):
        st, frame = R.step()
        stats.append(st.as_dict())
    barrier()
    dt = time.perf_counter() - t0

    keys = ["rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "photon_queries", "photons_visited",
            "bvh_nodes_visited", "tris_tested", "instance_visits", "gather_rounds", "gather_slow", "gather_leaf_reads"]
    tot = {k: float(sum(x[k] for x in stats)) for k in keys}
    ms = {k: float(sum(x[k] for x in stats)) for k in ("ms_trace", "ms_gather", "ms_resolve", "ms_total")}
    launches_g = float(sum(x["launches_gather"] for x in stats))
    vec = torch.tensor([dt] + [tot[k] for k in keys] + [ms["ms_gather"], ms["ms_trace"], launches_g], dtype=torch.float64,
                       device="cpu" if rehearsal else "cuda")
    if world > 1:
        mx = vec.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        for i, k in enumerate(keys):
            tot[k] = float(vec[1 + i])
        # roofline figures are quoted for rank 0's kernels
    rays = tot["rays_primary"] + tot["rays_shadow"] + tot["rays_reflect"] + tot["rays_refract"]

    if rank == 0:
        my = {k: float(sum(x[k] for x in stats)) for k in keys}
        # algorithmic bytes of the photon gather: 24 B query + 24 B result per query,
        # 24 B (the reference's photon record) per photon examined
        g_bytes = my["photon_queries"] * 48.0 + my["photons_visited"] * 24.0
        g_gbs = g_bytes / (ms["ms_gather"] * 1e-3) / 1e9 if ms["ms_gather"] > 0 else 0.0
        # algorithmic bytes of the trace+shade kernels: 48 B/ray + 84 B per object transform +
        # 28 B per BVH node visit + 48 B per triangle test
        my_rays = my["rays_primary"] + my["rays_shadow"] + my["rays_reflect"] + my["rays_refract"]
        t_bytes = my_rays * 48.0 + my["instance_visits"] * 84.0 + my["bvh_nodes_visited"] * 28.0 + my["tris_tested"] * 48.0
        t_gbs = t_bytes / (ms["ms_trace"] * 1e-3) / 1e9 if ms["ms_trace"] > 0 else 0.0
        # Up to three chunks of a frame are in flight at once, so the per-stream event intervals of different
        # kernels overlap in time and their sums no longer partition the frame; the roofline is quoted for the
        # kernel class that moves the most algorithmic bytes (with RT_STREAMS=1, where intervals are exclusive,
        # that is also the one taking the most time: profiles/r01d)
        gather_dominant = g_bytes >= t_bytes
        default_workload = (a.workload, a.width, a.height, a.spp, a.photons, a.synthetic_photons, world) == ("cornell", 1920, 1080, 64, 1000000, False, 1)
        traffic, traffic_src = measured_traffic("k_gather", default_workload and gather_dominant)
        roof = {"bound": "hbm", "kernel": "k_gather" if gather_dominant else "k_primary+k_bounce",
                "achieved": round(g_gbs if gather_dominant else t_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round((g_gbs if gather_dominant else t_gbs) / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(g_bytes / max(launches_g, 1)) if gather_dominant else None,
                "avg_launch_ms": round(ms["ms_gather"] / max(launches_g, 1), 4) if gather_dominant else None,
                "launches": int(launches_g) if gather_dominant else int(sum(x["launches_trace"] for x in stats)),
                # frac near or above 1 is not an accounting slip: the formula prices every photon examined at
                # the reference's 24 B against the HBM peak, but the 34 MB photon structure is served from
                # L2 / Infinity Cache (`traffic` = the HBM bytes actually moved per launch); the kernel is
                # bound by the latency of the dependent chain inside a query
                "note": ("photon structure is L2/MALL resident: see traffic; kernel is latency bound, not HBM bound" if gather_dominant else
                         "scene (BVH, triangles, transforms) is L2 resident; achieved = algorithmic bytes of traversal / kernel time"),
                "other": {"k_gather_GBps": round(g_gbs, 2), "trace_shade_GBps": round(t_gbs, 2),
                          "ms_gather": round(ms["ms_gather"], 2), "ms_trace_shade": round(ms["ms_trace"], 2),
                          "ms_resolve": round(ms["ms_resolve"], 3)}}
        out = {
            "metric": "Mray/s", "value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C5 stand-in: 128 tessellated spheres (102 402 triangles, half mirrors) on a ground quad, FIN shading, "
                                    f"{a.width}x{a.height}, {a.spp} spp fixed, no photon map, bounce 4") if a.workload == "balls" else
                                   f"Cornell box, FIN shading, {a.width}x{a.height}, {a.spp} spp fixed, "
                                   f"{len(balanced) - 1}-photon map ({'sprinkled on the walls' if a.synthetic_photons else 'GPU photon pass, Philox seed 20171203, 8 bounces'}), "
                                   f"k=400 r=1, bounce 4",
                       "tiles": "32x8 interleaved, tile t -> rank t mod N", "exchange": "one all_gather of 8 B/pixel per frame"},
            "frame_ms": round(dt / a.steps * 1e3, 2),
            **({"rehearsal": "all ranks on device 0, gloo gather -- not a scaling measurement"} if rehearsal else {}),
            "rays_per_frame": {k: int(tot[k] / a.steps) for k in keys[:4]},
            "photon_queries_per_frame": int(tot["photon_queries"] / a.steps),
            "gather_per_frame": {k: int(tot[k] / a.steps) for k in ("photons_visited", "gather_rounds", "gather_slow", "gather_leaf_reads")},
            "traversal_per_frame": {k: int(tot[k] / a.steps) for k in ("instance_visits", "bvh_nodes_visited", "tris_tested")},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s.export(), balanced, cam, p, a.cpu_seconds)
        if world > 1:
            # the gathered frame must equal what the tiles say: spot-check against rank 0's own tiles
            frgb, fz, fcnt = frame
            own = R.rgb.cpu() if rehearsal else R.rgb
            t = 0                                    # tile 0 belongs to rank 0
            assert bool((frgb[:8, :32].cpu() == own[:8, :32].cpu()).all()), "gathered frame disagrees with rank 0's tile"
            out["gathered_frame_nonzero_fraction"] = round(float((fz.float() != 0).float().mean()), 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
