// ref_main_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the REFERENCE's own main.cpp -- TraceNode, GenLight::Shadow, MtlBlinn::Shade, RenderPixel,
// PhotonTracing / CausticTracing, PointLight::RandomPhoton -- to produce the golden vectors that pin the
// shading half of oracle/rt_oracle.c.  Nothing of the reference is copied: oracle/Makefile feeds g++ the
// text of <snapshot>/main.cpp where it lies, minus exactly two lines (`#include "viewport.cpp"`, the GLUT
// window, and the `ShowViewport();` call in main()), through a temporary file outside the repository that
// is deleted after the compile; REF_MAIN names it.  main.cpp itself includes xmlload.cpp, so the scene is
// loaded by the reference's own LoadScene (TinyXML, cyTriMesh OBJ loader, cyBVH) too.  No GL header, no
// GLUT stand-in: with viewport.cpp gone nothing refers to OpenGL any more.
//
// What viewport.cpp ALSO held are the OpenGL display hooks of the plugin classes.  Four of them are the
// key functions of their classes (first non-inline virtual), so without a definition no vtable is emitted
// and the scene's objects could not be constructed:
//     Sphere::ViewportDisplay, Plane::ViewportDisplay, TriObj::ViewportDisplay   (objects.h:72,114,135)
//     PointLight::SetViewportLight                                               (lights.h:160)
// They are defined below as EMPTY bodies (test doubles, like the recording GenLight::Shadow of
// ref_harness.cpp): they draw a preview in a window that does not exist here, execute no arithmetic and
// are never called by the render path.  The other viewport hooks (MtlBlinn::SetViewportMaterial,
// TextureFile/TextureChecker::SetViewportTexture, GenLight::SetViewportParam) stay undefined; the binary is
// linked --unresolved-symbols=ignore-all like ref_harness and never calls them.
//
// rand(): the reference draws from libc rand().  Runs are made reproducible the plain way -- srand(seed),
// draw and keep the raw values, srand(seed) again, call the reference -- and the oracle is fed the kept
// values (orc_script_begin).  How many the reference consumed is found afterwards by drawing a few more
// and locating them in the kept sequence.
//
// usage: ref_main_harness <command> <scene.xml> <in.bin> <out.bin>      (cwd = the scene's directory)
// All files are raw little-endian arrays; layouts are documented next to each command and mirrored by
// oracle/gen_golden.py.
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

// ---- the four empty display hooks (see the header) ------------------------------------------------
void Sphere::ViewportDisplay(const Material *) const {}
void Plane::ViewportDisplay(const Material *) const {}
void TriObj::ViewportDisplay(const Material *) const {}
void PointLight::SetViewportLight(int) const {}

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
struct Out {
    FILE *fp;
    explicit Out(const char *p) { fp = fopen(p, "wb"); if (!fp) { fprintf(stderr, "cannot write %s\n", p); exit(2); } }
    ~Out() { fclose(fp); }
    template <class T> void put(const T &v) { fwrite(&v, sizeof(T), 1, fp); }
    void bytes(const void *p, size_t n) { if (n) fwrite(p, 1, n, fp); }
};

// nodes in the order TraceNode visits them (a node, then its children in turn): the index the product's rt_node array uses
static void walk(const Node *n, std::vector<const Node *> &order)
{
    order.push_back(n);
    for (int i = 0; i < n->GetNumChild(); i++) walk(n->GetChild(i), order);
}
static std::vector<const Node *> g_order;
static int32_t node_index(const Node *n)
{
    for (size_t i = 0; i < g_order.size(); i++) if (g_order[i] == n) return (int32_t)i;
    return -1;
}

struct HitRec { int32_t hit; float z; float p[3]; float N[3]; int32_t front; int32_t node; };
static HitRec rec(bool h, const HitInfo &hi)
{
    HitRec r; memset(&r, 0, sizeof r);
    r.hit = h; r.z = hi.z; r.node = -1;
    if (h) { r.p[0] = hi.p.x; r.p[1] = hi.p.y; r.p[2] = hi.p.z; r.N[0] = hi.N.x; r.N[1] = hi.N.y; r.N[2] = hi.N.z; r.front = hi.front; r.node = node_index(hi.node); }
    return r;
}

// ---- photon maps -----------------------------------------------------------------------------------
// AddPhoton leaves bits 0-2 of planeAndDirZ uninitialised (a stack Photon, cyPhotonMap.h:184-192); SetPlane writes bits
// 0-1 of every internal node while balancing, the rest is never read.  Zeroed so that dumps are reproducible.
// NumPhotons() is photons.size() - 1 (cyPhotonMap.h:86): an unsigned wrap-around on a map nothing was added to
static unsigned stored(const cy::PhotonMap &pm) { const unsigned n = pm.NumPhotons(); return n == ~0u ? 0u : n; }
static void clean_plane_bits(cy::PhotonMap &pm)
{
    const unsigned n = stored(pm);
    if (!n) return;
    unsigned char *raw = (unsigned char *)(pm.GetPhotons() - 1);       // photons[0] .. photons[n]
    memset(raw, 0, 24);
    for (unsigned i = 1; i <= n; i++) raw[24 * i + 19] &= 0x8;
}
// in: int32 n; n x {float pos[3], dir[3], power[3]}  ->  AddPhoton each, PrepareForIrradianceEstimation
static const char *load_photons(cy::PhotonMap &pm, const char *c)
{
    int32_t n = *(const int32_t *)c; c += 4;
    const float *f = (const float *)c; c += (size_t)n * 36;
    pm.AllocatePhotons(n);
    for (int i = 0; i < n; i++) pm.AddPhoton(Point3(f[9 * i], f[9 * i + 1], f[9 * i + 2]), Point3(f[9 * i + 3], f[9 * i + 4], f[9 * i + 5]), Color(f[9 * i + 6], f[9 * i + 7], f[9 * i + 8]));
    clean_plane_bits(pm);
    pm.PrepareForIrradianceEstimation();
    return c;
}
static void dump_balanced(Out &o, cy::PhotonMap &pm)
{
    int32_t n = (int32_t)stored(pm);
    o.put(n);
    if (n) o.bytes(pm.GetPhotons() - 1, (size_t)(n + 1) * 24);
}

// GenLight::Shadow is a protected static member (lights.h): reached the way the lights reach it, from a derived class
struct ShadowProbe : public GenLight { static float call(Ray r, float t_max) { return Shadow(r, t_max); } };

// ---- shade <scene.xml> <in> <out> ---------------------------------------------------------------------
// in:  photons (see load_photons); int32 n; n x {float ray[6]; int32 bounce}; int32 ns; ns x {float ray[6]; float t_max}
// out: int32 np + balanced photon array ((np+1) x 24 B, index 0 unused) as the reference's own
//      PrepareForIrradianceEstimation left it; n x HitRec = TraceNode(rootNode, ray, hit) (FIN/main.cpp:108-130);
//      n x float rgb[3] = hit.node->GetMaterial()->Shade(ray, hit, lights, bounce, 0) for the rays that hit (else 0);
//      ns x float = GenLight::Shadow(ray, t_max) (FIN/main.cpp:499-513)
static int cmd_shade(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = load_photons(photonmap, b.data());
    int32_t n = *(const int32_t *)c; c += 4;
    const char *cases = c; c += (size_t)n * 28;
    int32_t ns = *(const int32_t *)c; c += 4;
    const float *sh = (const float *)c;
    Out o(out);
    dump_balanced(o, photonmap);
    std::vector<float> rgb((size_t)n * 3, 0.0f);
    srand(12345);                                        // FIN's discarded hemisphere loop draws; its result is dropped
    for (int i = 0; i < n; i++) {
        const float *r = (const float *)(cases + 28 * (size_t)i);
        const int32_t bounce = *(const int32_t *)(cases + 28 * (size_t)i + 24);
        Ray ray(Point3(r[0], r[1], r[2]), Point3(r[3], r[4], r[5]));
        HitInfo hi; hi.Init();
        const bool h = TraceNode(rootNode, ray, hi);
        o.put(rec(h, hi));
        if (h) {
            const Color col = hi.node->GetMaterial()->Shade(ray, hi, lights, bounce, 0);
            rgb[3 * i] = col.r; rgb[3 * i + 1] = col.g; rgb[3 * i + 2] = col.b;
        }
    }
    o.bytes(rgb.data(), rgb.size() * 4);
    for (int i = 0; i < ns; i++) {
        const float *r = sh + 7 * (size_t)i;
        const float v = ShadowProbe::call(Ray(Point3(r[0], r[1], r[2]), Point3(r[3], r[4], r[5])), r[6]);
        o.put(v);
    }
    return 0;
}

// ---- pixels <scene.xml> <in> <out> --------------------------------------------------------------------
// RenderPixel (FIN/main.cpp:202-344, P13/main.cpp:191-333) on pixel segments of a frame, ONE worker thread.
// in:  photons; int32 width, height (camera.imgWidth/imgHeight override; 0 = keep the file's); int32 nseg;
//      nseg x {int32 start, count} (row-major pixel indices, ascending, disjoint)
// out: int32 np + balanced photons; int32 width, height; per segment count x {uint8 rgb[3]}, count x float z,
//      count x uint8 sampleCount; then one double: seconds spent inside RenderPixel over all segments (bench.py's
//      `reference_here` figure: the reference's own loop timed on the host it runs on)
// The shared pixel index is advanced to `start` with the iterator's own public GetPixel (as SURVEY 8c notes), the
// worker is stopped with pIt.setFlag() once `count` pixels are done; pixels it finishes beyond the segment are ignored.
static int cmd_pixels(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = load_photons(photonmap, b.data());
    int32_t w = *(const int32_t *)c; c += 4;
    int32_t h = *(const int32_t *)c; c += 4;
    int32_t nseg = *(const int32_t *)c; c += 4;
    const int32_t *seg = (const int32_t *)c;
    if (w > 0 && h > 0) { camera.imgWidth = w; camera.imgHeight = h; renderImage.Init(w, h); }
    w = camera.imgWidth; h = camera.imgHeight;
    Out o(out);
    dump_balanced(o, photonmap);
    o.put(w); o.put(h);
    pIt.Init();
    int next = 0;                                         // the iterator's next index
    double render_seconds = 0;
    srand(4242);
    for (int s = 0; s < nseg; s++) {
        const int start = seg[2 * s], count = seg[2 * s + 1];
        if (start < next || start + count > w * h) { fprintf(stderr, "bad segment %d\n", s); return 3; }
        pIt.clearFlag();
        int x, y;
        while (next < start) { if (!pIt.GetPixel(x, y)) { fprintf(stderr, "iterator ran out\n"); return 3; } next++; }
        const int before = renderImage.GetNumRenderedPixels();
        const auto t0 = std::chrono::steady_clock::now();
        std::thread th(RenderPixel, std::ref(pIt));
        // the segment is complete when `count` pixels have been COUNTED; P13 does not count the rows its debug skip
        // (main.cpp:219) passes over, so segments there must start at row 327 or below
        while (renderImage.GetNumRenderedPixels() - before < count) std::this_thread::sleep_for(std::chrono::microseconds(200));
        pIt.setFlag();
        th.join();
        const int done = renderImage.GetNumRenderedPixels() - before;
        // (the worker finishes the pixel it is in: `done` pixels were rendered in this time, `count` are reported)
        render_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * (double)count / (double)(done > 0 ? done : 1);
        next = start + done;                              // every counted pixel took one index (no skipped rows inside a segment)
        o.bytes(renderImage.GetPixels() + start, (size_t)count * 3);
        o.bytes(renderImage.GetZBuffer() + start, (size_t)count * 4);
        o.bytes(renderImage.GetSampleCount() + start, (size_t)count);
    }
    o.put(render_seconds);
    return 0;
}

// ---- photontrace <scene.xml> <in> <out> ---------------------------------------------------------------
// generatePhotonMap's loop (FIN/main.cpp:350-396) with the reference's own RandomPhoton / TraceNode / IsPhotonSurface /
// PhotonTracing.  generatePhotonMap itself cannot be called: it indexes photonLights.at(1) (:370: throws with one light)
// and fwrite()s to a NULL FILE* (:398-400); the loop is restated here with those two lines made safe -- the light is
// still chosen by one rand() draw compared with 0.5 (second light only if there is one), the dump is skipped.
// in:  uint32 seed; int32 max_photons; int32 bounce (PHOTON_BOUNCE); int32 mode (0 photon map, 1 caustic map: the loop the
//      reference keeps in a comment, :405-428, CausticTracing :461-487, stop when max_photons diffuse hits are COUNTED);
//      int32 capture (rand() values to keep)
// out: int32 consumed (rand() calls the loop made; -1 if beyond `capture`); consumed x int32 raw rand values;
//      int64 attempts; int32 n stored; n x 24 B photons after ScalePhotonPowers(4*pi/n), BEFORE balancing (the .dat dump)
static int cmd_photontrace(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    const uint32_t seed = *(const uint32_t *)c; c += 4;
    const int32_t max_photons = *(const int32_t *)c; c += 4;
    const int32_t bounce = *(const int32_t *)c; c += 4;
    const int32_t mode = *(const int32_t *)c; c += 4;
    const int32_t capture = *(const int32_t *)c; c += 4;
    std::vector<int32_t> raw(capture + 8);
    srand(seed);
    for (size_t i = 0; i < raw.size(); i++) raw[i] = rand();
    srand(seed);
    cy::PhotonMap &pm = mode == 1 ? causticmap : photonmap;
    pm.AllocatePhotons(max_photons);
    std::vector<Light *> photonLights;
    for (int i = 0; i < (int)lights.size(); i++) if (!lights[i]->IsAmbient()) photonLights.push_back(lights[i]);
    if (photonLights.empty()) { fprintf(stderr, "no photon source\n"); return 3; }
    int numofphoton = 0;
    int64_t attempts = 0;
    while (numofphoton < max_photons) {
        attempts++;
        PointLight *pL;
        if ((rand() / (float)RAND_MAX) < 0.5) pL = (PointLight *)(photonLights.at(0));
        else pL = (PointLight *)(photonLights.at(photonLights.size() > 1 ? 1 : 0));
        Color lightColor = pL->GetPhotonIntensity();
        Ray rayFromL = pL->RandomPhoton();
        HitInfo hitInfo;
        hitInfo.Init();
        if (TraceNode(rootNode, rayFromL, hitInfo)) {
            const MtlBlinn *mtlb = static_cast<const MtlBlinn *>(hitInfo.node->GetMaterial());
            if (mode == 0) {
                if (mtlb->IsPhotonSurface()) PhotonTracing(rayFromL, hitInfo, lightColor, numofphoton, bounce);
            } else {
                if (mtlb->IsPhotonSurface()) CausticTracing(rayFromL, hitInfo, lightColor, numofphoton, bounce, 0);
                else CausticTracing(rayFromL, hitInfo, lightColor, numofphoton, bounce, 1);
            }
        }
    }
    // how many rand() calls was that: the next values of the stream are located in the kept sequence
    int32_t consumed = -1;
    {
        const int32_t v0 = rand(), v1 = rand(), v2 = rand(), v3 = rand();
        for (int i = 0; i + 3 < (int)raw.size(); i++)
            if (raw[i] == v0 && raw[i + 1] == v1 && raw[i + 2] == v2 && raw[i + 3] == v3) { consumed = i; break; }
        if (consumed > capture) consumed = -1;
    }
    if (stored(pm)) pm.ScalePhotonPowers(1.0 * 4 * M_PI / pm.NumPhotons());
    clean_plane_bits(pm);
    Out o(out);
    o.put(consumed);
    if (consumed > 0) o.bytes(raw.data(), (size_t)consumed * 4);
    o.put(attempts);
    int32_t n = (int32_t)stored(pm);
    o.put(n);
    if (n) o.bytes(pm.GetPhotons(), (size_t)n * 24);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: ref_main_harness <shade|pixels|photontrace> scene.xml in out\n"); return 1; }
    const std::string cmd = argv[1];
    // the reference's loader prints the scene it parsed (xmlload.cpp:148-436): keep stdout for that, quietly
    if (!freopen("/dev/null", "w", stdout)) return 2;
    pIt.Init();
    if (!LoadScene(argv[2])) { fprintf(stderr, "LoadScene(%s) failed\n", argv[2]); return 2; }
    walk(&rootNode, g_order);
    if (cmd == "shade") return cmd_shade(argv[3], argv[4]);
    if (cmd == "pixels") return cmd_pixels(argv[3], argv[4]);
    if (cmd == "photontrace") return cmd_photontrace(argv[3], argv[4]);
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 1;
}
