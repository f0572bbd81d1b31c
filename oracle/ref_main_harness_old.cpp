// ref_main_harness_old.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The same recipe as ref_main_harness.cpp (which see) for the EARLIER snapshots whose MtlBlinn::Shade variants the product
// offers as shading models: RayTracingProj12 (live path-traced GI, BASELINE config C3), RayTracingProj6 (C2) and
// RayTracingProj3 (C1).  Their main.cpp is compiled where it lies with only the `#include "viewport.cpp"` and
// `ShowViewport();` lines removed; the OpenGL display hooks that are key functions of their classes get empty bodies
// (RayTracingProj12 / 6: Sphere, Plane, TriObj::ViewportDisplay; RayTracingProj3: Sphere::ViewportDisplay -- PointLight's
// hook is inline in these snapshots).  Drives TraceNode(rootNode), hit.node->GetMaterial()->Shade(...) and GenLight::Shadow.
//
// RayTracingProj12's Shade draws from rand() (hemisphere rays, RayTracingProj12 main.cpp:393-446): every case is run under
// its own srand(seed) and the values it consumed are kept (found by locating the next ones in a pre-drawn copy of the
// stream), so that the oracle can be fed the very same draws in the very same order.
//
// usage: ref_main_harness_old shade <scene.xml> <in.bin> <out.bin>      (cwd = the scene's directory)
//        ref_main_harness_old pixels <scene.xml> <in.bin> <out.bin>     (RayTracingProj6 / 3: see cmd_pixels)
// in:  int32 n; n x {float ray[6]; int32 bounce; uint32 seed}; int32 ns; ns x {float ray[6]; float t_max}; int32 capture
// out: n x HitRec; n x float rgb[3]; n x int32 consumed (rand() calls of the case's Shade; -1: more than `capture`);
//      sum(consumed) x int32 raw rand values, case after case; ns x float shadow
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
#include REF_MAIN
#undef main
#undef gamma

void Sphere::ViewportDisplay(const Material *) const {}
#ifndef REF_P3
void Plane::ViewportDisplay(const Material *) const {}
void TriObj::ViewportDisplay(const Material *) const {}
#endif

#if defined(REF_P3)
#define REF_SHADE(ray, hi, bounce) (hi).node->GetMaterial()->Shade(ray, hi, lights)
#else
#define REF_SHADE(ray, hi, bounce) (hi).node->GetMaterial()->Shade(ray, hi, lights, bounce)
#endif

static std::vector<char> slurp(const char *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<char> b(n);
    if (n && fread(b.data(), 1, n, fp) != (size_t)n) { fprintf(stderr, "short read\n"); exit(2); }
    fclose(fp);
    return b;
}
static void walk(const Node *n, std::vector<const Node *> &order)
{
    order.push_back(n);
    for (int i = 0; i < n->GetNumChild(); i++) walk(n->GetChild(i), order);
}
static std::vector<const Node *> g_order;
struct HitRec { int32_t hit; float z; float p[3]; float N[3]; int32_t front; int32_t node; };
struct ShadowProbe : public GenLight { static float call(Ray r, float t_max) { return Shadow(r, t_max); } };

// ---- pixels <scene.xml> <in> <out> (RayTracingProj6 / RayTracingProj3 only) ----------------------------------------
// The snapshot's own RenderPixel(pixelIterator &) (RayTracingProj6 main.cpp:102-153, RayTracingProj3 main.cpp:85-129: one ray
// through the pixel centre, Shade, Color24) run over the WHOLE frame on this thread -- their iterators cannot be stopped
// (RayTracingProj3 has no flag, RayTracingProj6's GetPixel returns true without a pixel once its flag is set, :38).
// in:  int32 width, height (camera.imgWidth / imgHeight override; 0 = the file's)
// out: int32 width, height; width*height x uint8 rgb[3]; width*height x float z
#if defined(REF_P6) || defined(REF_P3)
static int cmd_pixels(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    int32_t w = *(const int32_t *)b.data(), h = *(const int32_t *)(b.data() + 4);
    if (w > 0 && h > 0) { camera.imgWidth = w; camera.imgHeight = h; renderImage.Init(w, h); }
    w = camera.imgWidth; h = camera.imgHeight;
    pIt.Init();
    RenderPixel(pIt);
    FILE *fo = fopen(out, "wb");
    if (!fo) return 3;
    fwrite(&w, 4, 1, fo); fwrite(&h, 4, 1, fo);
    fwrite(renderImage.GetPixels(), 3, (size_t)w * h, fo);
    fwrite(renderImage.GetZBuffer(), 4, (size_t)w * h, fo);
    fclose(fo);
    return renderImage.GetNumRenderedPixels() == w * h ? 0 : 4;
}
#endif

int main(int argc, char **argv)
{
    if (argc < 5 || (strcmp(argv[1], "shade") && strcmp(argv[1], "pixels"))) { fprintf(stderr, "usage: ref_main_harness_old <shade|pixels> scene.xml in out\n"); return 1; }
    if (!freopen("/dev/null", "w", stdout)) return 2;
    if (!LoadScene(argv[2])) { fprintf(stderr, "LoadScene(%s) failed\n", argv[2]); return 2; }
#if defined(REF_P6) || defined(REF_P3)
    if (!strcmp(argv[1], "pixels")) return cmd_pixels(argv[3], argv[4]);
#endif
    if (strcmp(argv[1], "shade")) { fprintf(stderr, "this snapshot's harness has no `%s`\n", argv[1]); return 1; }
    walk(&rootNode, g_order);
    std::vector<char> b = slurp(argv[3]);
    const char *c = b.data();
    const int32_t n = *(const int32_t *)c; c += 4;
    const char *cases = c; c += (size_t)n * 32;
    const int32_t ns = *(const int32_t *)c; c += 4;
    const float *sh = (const float *)c; c += (size_t)ns * 28;
    const int32_t capture = *(const int32_t *)c;
    FILE *fo = fopen(argv[4], "wb");
    if (!fo) return 3;
    std::vector<HitRec> recs(n);
    std::vector<float> rgb((size_t)n * 3, 0.0f);
    std::vector<int32_t> consumed(n, 0), raw_all;
    std::vector<int32_t> raw(capture + 8);
    for (int i = 0; i < n; i++) {
        const float *r = (const float *)(cases + 32 * (size_t)i);
        const int32_t bounce = *(const int32_t *)(cases + 32 * (size_t)i + 24);
        const uint32_t seed = *(const uint32_t *)(cases + 32 * (size_t)i + 28);
        Ray ray(Point3(r[0], r[1], r[2]), Point3(r[3], r[4], r[5]));
        HitInfo hi; hi.Init();
        const bool h = TraceNode(rootNode, ray, hi);
        HitRec &q = recs[i]; memset(&q, 0, sizeof q);
        q.hit = h; q.z = hi.z; q.node = -1;
        if (!h) continue;
        q.p[0] = hi.p.x; q.p[1] = hi.p.y; q.p[2] = hi.p.z; q.N[0] = hi.N.x; q.N[1] = hi.N.y; q.N[2] = hi.N.z; q.front = hi.front;
        for (size_t k = 0; k < g_order.size(); k++) if (g_order[k] == hi.node) q.node = (int32_t)k;
        srand(seed);
        for (size_t k = 0; k < raw.size(); k++) raw[k] = rand();
        srand(seed);
        (void)bounce;
        const Color col = REF_SHADE(ray, hi, bounce);
        rgb[3 * i] = col.r; rgb[3 * i + 1] = col.g; rgb[3 * i + 2] = col.b;
        const int32_t v0 = rand(), v1 = rand(), v2 = rand(), v3 = rand();
        int32_t used = -1;
        for (int k = 0; k + 3 < (int)raw.size(); k++) if (raw[k] == v0 && raw[k + 1] == v1 && raw[k + 2] == v2 && raw[k + 3] == v3) { used = k; break; }
        if (used > capture) used = -1;
        consumed[i] = used;
        if (used > 0) raw_all.insert(raw_all.end(), raw.begin(), raw.begin() + used);
    }
    fwrite(recs.data(), sizeof(HitRec), n, fo);
    fwrite(rgb.data(), 4, rgb.size(), fo);
    fwrite(consumed.data(), 4, n, fo);
    if (!raw_all.empty()) fwrite(raw_all.data(), 4, raw_all.size(), fo);
    for (int i = 0; i < ns; i++) {
        const float *r = sh + 7 * (size_t)i;
        const float v = ShadowProbe::call(Ray(Point3(r[0], r[1], r[2]), Point3(r[3], r[4], r[5])), r[6]);
        fwrite(&v, 4, 1, fo);
    }
    fclose(fo);
    return 0;
}
