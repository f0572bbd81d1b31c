// ref_binding_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Links and runs the reference-side binding (raytracing_folder_amd/binding/rt_binding.cpp) inside the REFERENCE's own
// program: main.cpp of RayTracingFinal compiled where it lies (the same recipe as ref_main_harness.cpp: its text minus the
// `#include "viewport.cpp"` and `ShowViewport();` lines, through a temporary file outside the repository), with its three
// driver functions renamed by macros for this translation unit -- BeginRender, StopRender and saveImage are exactly what
// the binding replaces (FIN/main.cpp:984-1012).  So: the reference's LoadScene builds the reference's Node / Material /
// Light objects, the binding lowers THEM through the C ABI, librt_mi355x renders into the reference's RenderImage, and the
// reference's saveImage writes the PNGs with its lodepng.  The four empty OpenGL display hooks are those of
// ref_main_harness.cpp (see its header).  Runs on the GPU box (oracle/_ref/ travels there like any built file) on the
// product's own copy of the Cornell scene -- the reference tree itself is not needed at run time.
//
//   ref_binding_harness <scene.xml> <out.bin> [width height]      (cwd = where the PNGs go; OBJ names are cwd-relative)
//   out.bin: int32 width, height, rendered pixels; Color24[w*h]; float z[w*h]; uint8 sampleCount[w*h]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <vector>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <iostream>
#include <cmath>

#define main ref_main
#define BeginRender ref_BeginRender
#define StopRender ref_StopRender
#define saveImage ref_saveImage
#include REF_MAIN
#undef main
#undef BeginRender
#undef StopRender
#undef saveImage
#undef gamma

void Sphere::ViewportDisplay(const Material *) const {}
void Plane::ViewportDisplay(const Material *) const {}
void TriObj::ViewportDisplay(const Material *) const {}
void PointLight::SetViewportLight(int) const {}

// the binding's three functions (rt_binding.cpp, its own translation unit)
void BeginRender();
void StopRender();
void saveImage();
int RenderProgress();
bool RenderRunning();

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_binding_harness scene.xml out.bin [width height]\n"); return 1; }
    FILE *quiet = freopen("/dev/null", "w", stdout);          // the loader prints the scene it parsed
    (void)quiet;
    pIt.Init();
    if (!LoadScene(argv[1])) { fprintf(stderr, "LoadScene(%s) failed\n", argv[1]); return 2; }
    if (argc > 4) { camera.imgWidth = atoi(argv[3]); camera.imgHeight = atoi(argv[4]); renderImage.Init(camera.imgWidth, camera.imgHeight); }
    const int w = camera.imgWidth, h = camera.imgHeight;
    const auto t0 = std::chrono::steady_clock::now();
    BeginRender();                                             // must return at once (viewport.cpp:36)
    const double begin_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!RenderRunning()) { fprintf(stderr, "BeginRender did not start a render\n"); return 5; }
    int polls = 0;
    while (RenderProgress() < w * h && polls < 600000) { polls++; std::this_thread::sleep_for(std::chrono::microseconds(200)); }
    const int done = RenderProgress();
    saveImage();                                               // waits for the jobs, then the reference's own image code
    FILE *fp = fopen(argv[2], "wb");
    if (!fp) return 3;
    const int32_t hdr[3] = {w, h, done};
    fwrite(hdr, 4, 3, fp);
    fwrite(renderImage.GetPixels(), 3, (size_t)w * h, fp);
    fwrite(renderImage.GetZBuffer(), 4, (size_t)w * h, fp);
    fwrite(renderImage.GetSampleCount(), 1, (size_t)w * h, fp);
    fclose(fp);
    fprintf(stderr, "begin_ms %.3f polls %d pixels %d of %d\n", begin_ms, polls, done, w * h);
    return done == w * h ? 0 : 4;
}
