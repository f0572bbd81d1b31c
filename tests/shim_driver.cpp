// shim_driver.cpp -- test infrastructure: drives rt::Renderer the way the reference's viewport drives
// BeginRender / StopRender / saveImage (FIN/viewport.cpp:35-37, 390-409, 443-453): load the scene file,
// BeginRender() must return at once, poll renderImage.IsRenderDone() while reading GetPixels(), then
// saveImage().  Built and run by tests/test_host.py (no GPU: BeginRender must fail loudly) and
// tests/test_gpu_parity.py (the PNGs must equal a render through the C ABI).
//   shim_driver <scene.xml> <image.png> <samples.png> <z.png> [stop | dump=<photons.dat> | photons=<count>]
// BeginRender runs the photon pass first, like the reference's (FIN/main.cpp:984-998 -> generatePhotonMap :350-402), on every
// gfx950 device of the node (one job per device, interleaved tiles into the same RenderImage).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../raytracing_folder_amd/csrc/host/rt_shim.h"

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: shim_driver scene.xml image.png samples.png z.png [stop]\n"); return 2; }
    rt::Renderer r;
    if (!r.LoadScene(argv[1])) { fprintf(stderr, "LoadScene failed: %s\n", r.LastError().c_str()); return 3; }
    bool stop = false;
    for (int i = 5; i < argc; i++) {
        if (!strcmp(argv[i], "stop")) stop = true;
        else if (!strncmp(argv[i], "dump=", 5)) r.photonDump = argv[i] + 5;
        else if (!strncmp(argv[i], "photons=", 8)) r.params.photon_count = atoi(argv[i] + 8);
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (!r.BeginRender()) { fprintf(stderr, "BeginRender failed: %s\n", r.LastError().c_str()); return 4; }
    const double begin_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int polls = 0, partial = 0;
    while (!r.renderImage.IsRenderDone()) {
        polls++;
        const int n = r.renderImage.GetNumRenderedPixels();
        if (n > 0 && n < r.renderImage.GetWidth() * r.renderImage.GetHeight()) partial++;
        if (stop && n > 0) { r.StopRender(); break; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (polls > 500000) { fprintf(stderr, "render did not finish\n"); return 5; }
    }
    if (!r.WaitRender()) { fprintf(stderr, "render failed: %s\n", r.LastError().c_str()); return 6; }
    r.saveImage(argv[2], argv[3], argv[4]);
    printf("begin_ms %.3f polls %d partial %d pixels %d of %d rays %llu photon_queries %llu photon_pass_ms %.3f structure_ms %.3f\n", begin_ms, polls, partial,
           r.renderImage.GetNumRenderedPixels(), r.renderImage.GetWidth() * r.renderImage.GetHeight(),
           (unsigned long long)(r.Stats().rays_primary + r.Stats().rays_shadow + r.Stats().rays_reflect + r.Stats().rays_refract),
           (unsigned long long)r.Stats().photon_queries, r.PhotonPassMs().total, r.PhotonPassMs().structure_build);
    return 0;
}
