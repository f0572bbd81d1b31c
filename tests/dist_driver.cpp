// dist_driver.cpp -- a C++ host that shards one frame over the gfx950 devices of a node and assembles it with RCCL,
// written against include/rt_mi355x.h only (INTEGRATION.md section 4 as code; what raytracing_folder_amd/dist.py does
// through torch.distributed, without Python):
//
//   per device r of N:  rt_scene_generate_photons (same seed everywhere: identical maps, no exchange)
//                       rt_render_tiles_packed_device: tiles r, r+N, ... rendered straight into the buffer this device
//                                                      contributes (8-byte pixel records, tile by tile)
//   once:               ncclAllGather of the N contributions (ncclGroupStart/End, one communicator per device)
//   per device:         rt_tiles_unpack_device: gathered tiles -> the three RenderImage planes
//
// The reference spreads the pixels over its threads with one shared counter (FIN/main.cpp:71-78); nothing is exchanged there.
// One process drives all devices here (ncclCommInitAll needs no bootstrap), one host thread per device for the render.
// Built by tests/test_host.py (compile + link against /opt/rocm/include/rccl: no GPU needed) and run by
// tests/test_gpu_parity.py with the devices the box has (on a one-GPU box N = 1: the all-gather is then a copy, every call
// is still made).
//   dist_driver <scene.xml> <width> <height> <spp> <photons> <out.png> [max_devices]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_mi355x.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 11; } } while (0)
#define RTOK(x) do { rt_status s_ = (x); if (s_ != RT_OK) { fprintf(stderr, "%s: %s\n", #x, rt_last_error()); return 12; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: dist_driver scene.xml width height spp photons out.png [max_devices]\n"); return 2; }
    const int width = atoi(argv[2]), height = atoi(argv[3]), spp = atoi(argv[4]), n_photons = atoi(argv[5]);
    int N = rt_device_count();
    if (argc > 7 && atoi(argv[7]) > 0 && atoi(argv[7]) < N) N = atoi(argv[7]);
    if (N < 1) { fprintf(stderr, "no gfx950 device (the render path has no CPU fallback)\n"); return 3; }

    rt_scene *scene = nullptr;
    RTOK(rt_scene_create(&scene));
    RTOK(rt_scene_load_xml(scene, argv[1]));
    rt_camera cam;
    RTOK(rt_scene_get_camera(scene, &cam));
    cam.width = width; cam.height = height;
    rt_params p;
    rt_params_default(&p);
    p.min_sample = p.max_sample = spp; p.threshold = -1.0f;

    std::vector<int> devs(N);
    for (int r = 0; r < N; r++) devs[r] = r;
    std::vector<ncclComm_t> comms(N);
    NCCLOK(ncclCommInitAll(comms.data(), N, devs.data()));

    // every device holds the whole scene and photon map (generated once, uploaded to the others by the library)
    if (n_photons > 0) RTOK(rt_scene_generate_photons(scene, 0, (uint32_t)n_photons, p.photon_bounce, p.seed, nullptr, nullptr));

    const rt_tile_range first = {32, 8, 0, N};
    uint64_t per_rank_bytes = 0;
    int32_t per_rank_tiles = 0;
    RTOK(rt_tiles_packed_size(width, height, &first, &per_rank_bytes, &per_rank_tiles));   // rank 0 owns the most tiles: everyone contributes that many
    const size_t npx = (size_t)width * height;
    std::vector<hipStream_t> streams(N);
    std::vector<void *> mine(N), gathered(N);
    std::vector<uint8_t *> rgb(N), cnt(N);
    std::vector<float *> z(N);
    for (int r = 0; r < N; r++) {
        HIPOK(hipSetDevice(r));
        HIPOK(hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking));
        HIPOK(hipMalloc(&mine[r], per_rank_bytes));
        HIPOK(hipMemset(mine[r], 0, per_rank_bytes));                                        // a rank one tile short leaves its last slot zero
        HIPOK(hipMalloc(&gathered[r], per_rank_bytes * N));
        HIPOK(hipMalloc((void **)&rgb[r], npx * 3)); HIPOK(hipMalloc((void **)&z[r], npx * 4)); HIPOK(hipMalloc((void **)&cnt[r], npx));
    }

    double frame_ms = 0;
    rt_stats total;
    memset(&total, 0, sizeof total);
    for (int frame = 0; frame < 2; frame++) {           // frame 0 warms up (allocations, queue sizes from measurement)
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<rt_status> status(N, RT_OK);
        std::vector<rt_stats> stats(N);
        std::vector<std::thread> th;
        for (int r = 0; r < N; r++)
            th.emplace_back([&, r]() {
                const rt_tile_range tiles = {32, 8, r, N};
                status[r] = rt_render_tiles_packed_device(scene, &cam, &p, &tiles, r, streams[r], mine[r], per_rank_bytes, 1, &stats[r]);
            });
        for (auto &t : th) t.join();
        for (int r = 0; r < N; r++) if (status[r] != RT_OK) { fprintf(stderr, "render on device %d failed (%d)\n", r, status[r]); return 13; }
        // ONE all-gather of 8 B/pixel: rank r's tiles land at offset r of every device's gathered buffer
        NCCLOK(ncclGroupStart());
        for (int r = 0; r < N; r++) NCCLOK(ncclAllGather(mine[r], gathered[r], per_rank_bytes, ncclUint8, comms[r], streams[r]));
        NCCLOK(ncclGroupEnd());
        for (int r = 0; r < N; r++)
            RTOK(rt_tiles_unpack_device(r, streams[r], gathered[r], N, per_rank_tiles, width, height, 32, 8, rgb[r], z[r], cnt[r]));
        for (int r = 0; r < N; r++) { HIPOK(hipSetDevice(r)); HIPOK(hipStreamSynchronize(streams[r])); }
        frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        memset(&total, 0, sizeof total);
        for (int r = 0; r < N; r++) {
            total.rays_primary += stats[r].rays_primary; total.rays_shadow += stats[r].rays_shadow; total.rays_reflect += stats[r].rays_reflect;
            total.rays_refract += stats[r].rays_refract; total.photon_queries += stats[r].photon_queries; total.pixels += stats[r].pixels;
        }
    }
    // every device now holds the whole frame: the last one's copy is written out (so that N > 1 would show a broken gather)
    std::vector<uint8_t> host(npx * 3);
    HIPOK(hipSetDevice(N - 1));
    HIPOK(hipMemcpy(host.data(), rgb[N - 1], npx * 3, hipMemcpyDeviceToHost));
    RTOK(rt_image_write_png(argv[6], host.data(), width, height, 3));
    printf("devices %d pixels %llu rays %llu photon_queries %llu frame_ms %.3f bytes_per_rank %llu\n", N, (unsigned long long)total.pixels,
           (unsigned long long)(total.rays_primary + total.rays_shadow + total.rays_reflect + total.rays_refract),
           (unsigned long long)total.photon_queries, frame_ms, (unsigned long long)per_rank_bytes);
    for (int r = 0; r < N; r++) {
        (void)hipSetDevice(r);
        (void)hipFree(mine[r]); (void)hipFree(gathered[r]); (void)hipFree(rgb[r]); (void)hipFree(z[r]); (void)hipFree(cnt[r]);
        (void)hipStreamDestroy(streams[r]);
        ncclCommDestroy(comms[r]);
    }
    rt_scene_destroy(scene);
    return 0;
}
