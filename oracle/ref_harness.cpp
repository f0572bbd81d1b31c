// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the REFERENCE's own code -- compiled from the headers/sources where they lie under
// /root/reference (nothing is copied) -- to produce golden vectors that pin oracle/rt_oracle.c.
// Built by oracle/Makefile into oracle/_ref/ (git-ignored).  Runs only in the build container.
//
// What can be built: scene.h (incl. RenderImage::ComputeZBufferImage/ComputeSampleCountImage),
// scene.cpp (Box::IntersectRay), objects.h (Sphere, Plane, TriObj + cyTriMesh OBJ loader + cyBVH),
// cyPhotonMap.h, cyColor.h/cyPoint.h/cyMatrix.h, lights.h (the inline PointLight::Illuminate of
// either snapshot) and materials.h (the inline MtlBlinn::RandomPhotonBounce, Attenuation,
// createCoordinateSystem of RayTracingFinal).
// main.cpp itself (TraceNode, Shade, RenderPixel, GenLight::Shadow, PhotonTracing) is driven by
// ref_main_harness.cpp (round 3): compiled with only its viewport include removed, no GL header.
//
// GenLight::Shadow is only DECLARED by lights.h (its body, FIN/main.cpp:499-513, is in the
// unbuildable translation unit).  PointLight::Illuminate calls it once per shadow sample, so to
// exercise Illuminate this file defines GenLight::Shadow as a RECORDING TEST DOUBLE: it logs the
// ray it is handed and returns the next value of a script supplied by the caller.  What the
// `illum` vectors pin is therefore Illuminate's own arithmetic (sample placement, rand()
// consumption, the 4 -> 16 escalation rule, the fall-off); Shadow itself is pinned by ref_main_harness.
// libc rand() is made reproducible the plain way: srand(seed), draw and store the raw values,
// srand(seed) again, call the reference; the stored values are what the oracle is fed.
//
// The three Object subclasses declare a GLUT-drawing virtual (ViewportDisplay) that only
// viewport.cpp defines, so their vtables are never emitted.  The harness is therefore linked as
// an executable with --unresolved-symbols=ignore-all and only ever makes class-qualified
// (non-virtual) calls, e.g. obj.Sphere::IntersectRay(...).
//
// usage: ref_harness <command> <in.bin> <out.bin> [path]
// All files are raw little-endian arrays; layouts are documented next to each command and
// mirrored by oracle/gen_golden.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
#include <iostream>
#include <algorithm>
#include <cmath>

#include "scene.h"
#include "objects.h"
#include "cyPhotonMap.h"
#include "texture.h"
#include "lights.h"
#include "materials.h"

// the reference declares these as globals in main.cpp; objects.h has `extern` declarations only
Sphere theSphere;
Plane thePlane;

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
    void bytes(const void *p, size_t n) { fwrite(p, 1, n, fp); }
};

struct HitRec { int32_t hit; float z; float p[3]; float N[3]; int32_t front; };
static HitRec rec(bool h, const HitInfo &hi)
{
    HitRec r; memset(&r, 0, sizeof r);
    r.hit = h; r.z = hi.z;
    if (h) { r.p[0] = hi.p.x; r.p[1] = hi.p.y; r.p[2] = hi.p.z; r.N[0] = hi.N.x; r.N[1] = hi.N.y; r.N[2] = hi.N.z; r.front = hi.front; }
    return r;
}

// in: int32 n; n x {float ray[6]; float z0}   out: n x HitRec (sphere), n x HitRec (plane)
static int cmd_prims(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    int32_t n = *(int32_t *)b.data();
    const float *d = (const float *)(b.data() + 4);
    Out o(out);
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < n; i++) {
            Ray r(Point3(d[7 * i], d[7 * i + 1], d[7 * i + 2]), Point3(d[7 * i + 3], d[7 * i + 4], d[7 * i + 5]));
            HitInfo hi; hi.Init(); hi.z = d[7 * i + 6];
            bool h = pass == 0 ? theSphere.Sphere::IntersectRay(r, hi) : thePlane.Plane::IntersectRay(r, hi);
            o.put(rec(h, hi));
        }
    return 0;
}

// in: int32 n; n x {float box[6]; float ray[6]; float tmax}   out: n x int32
static int cmd_box(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    int32_t n = *(int32_t *)b.data();
    const float *d = (const float *)(b.data() + 4);
    Out o(out);
    for (int i = 0; i < n; i++) {
        const float *q = d + 13 * i;
        Box bx(q);
        Ray r(Point3(q[6], q[7], q[8]), Point3(q[9], q[10], q[11]));
        int32_t h = bx.IntersectRay(r, q[12]);
        o.put(h);
    }
    return 0;
}

// mesh <rays.bin> <out.bin> <file.obj>
// in: int32 n; n x {float ray[6]; float z0}
// out: int32 nv,nf,nvn,nnodes(with node 0); v[nv*3] f[nf*3] vn[nvn*3] fn[nf*3]
//      nodes[nnodes] x {float box[6]; uint32 data}; elements[nf]; then n x HitRec
static int cmd_mesh(const char *in, const char *out, const char *objpath, bool loadMtl = false, const char *mtlout = nullptr)
{
    TriObj *tobj = new TriObj;
    if (!tobj->Load(objpath, loadMtl)) { fprintf(stderr, "load failed\n"); return 3; }
    if (mtlout) {
        // int32 nm, nvt; per material: Kd[3] Ks[3] Tf[3] Ns Ni (floats), illum (int32), cumulative face count (int32),
        //                 256-byte map_Kd, 256-byte map_Ks; then vt[nvt*3] floats and ft[nf*3] uint32
        Out m(mtlout);
        int32_t nm = tobj->NM(), nvt = tobj->NVT();
        m.put(nm); m.put(nvt);
        int cum = 0;
        for (int i = 0; i < nm; i++) {
            const cyTriMesh::Mtl &mt = tobj->M(i);
            m.bytes(mt.Kd, 12); m.bytes(mt.Ks, 12); m.bytes(mt.Tf, 12); m.put(mt.Ns); m.put(mt.Ni);
            int32_t il = mt.illum; m.put(il);
            cum += tobj->GetMaterialFaceCount(i);
            int32_t c = cum; m.put(c);
            char buf[256];
            memset(buf, 0, 256); if (mt.map_Kd.data) strncpy(buf, mt.map_Kd.data, 255); m.bytes(buf, 256);
            memset(buf, 0, 256); if (mt.map_Ks.data) strncpy(buf, mt.map_Ks.data, 255); m.bytes(buf, 256);
        }
        for (int i = 0; i < nvt; i++) { m.put(tobj->VT(i).x); m.put(tobj->VT(i).y); m.put(tobj->VT(i).z); }
        if (nvt) for (unsigned i = 0; i < tobj->NF(); i++) for (int k = 0; k < 3; k++) { uint32_t x = tobj->FT(i).v[k]; m.put(x); }
    }
    // TriObj::bvh is private: rebuild an identical one through the public class (deterministic build)
    cyBVHTriMesh bvh;
    bvh.SetMesh(tobj, 4);
    std::vector<unsigned> ids; ids.push_back(bvh.GetRootNodeID());
    unsigned maxid = 1;
    const unsigned *ebase = nullptr;
    for (size_t k = 0; k < ids.size(); k++) {
        unsigned id = ids[k];
        if (id > maxid) maxid = id;
        if (bvh.IsLeafNode(id)) {
            const unsigned *e = bvh.GetNodeElements(id);
            if (!ebase || e < ebase) ebase = e;
        } else { ids.push_back(bvh.GetFirstChildNode(id)); ids.push_back(bvh.GetSecondChildNode(id)); }
    }
    int32_t nv = tobj->NV(), nf = tobj->NF(), nvn = tobj->NVN(), nnodes = maxid + 1;
    Out o(out);
    o.put(nv); o.put(nf); o.put(nvn); o.put(nnodes);
    for (int i = 0; i < nv; i++) { o.put(tobj->V(i).x); o.put(tobj->V(i).y); o.put(tobj->V(i).z); }
    for (int i = 0; i < nf; i++) for (int k = 0; k < 3; k++) { uint32_t x = tobj->F(i).v[k]; o.put(x); }
    for (int i = 0; i < nvn; i++) { o.put(tobj->VN(i).x); o.put(tobj->VN(i).y); o.put(tobj->VN(i).z); }
    for (int i = 0; i < nf; i++) for (int k = 0; k < 3; k++) { uint32_t x = tobj->FN(i).v[k]; o.put(x); }
    for (int id = 0; id < nnodes; id++) {
        float box[6] = {0, 0, 0, 0, 0, 0}; uint32_t data = 0;
        if (id >= 1) {
            memcpy(box, bvh.GetNodeBounds(id), sizeof box);
            if (bvh.IsLeafNode(id)) {
                uint32_t cnt = bvh.GetNodeElementCount(id);
                uint32_t off = (uint32_t)(bvh.GetNodeElements(id) - ebase);
                data = 0x80000000u | ((cnt - 1) << 28) | off;
            } else data = bvh.GetFirstChildNode(id);
        }
        o.bytes(box, sizeof box); o.put(data);
    }
    for (int i = 0; i < nf; i++) { uint32_t e = ebase[i]; o.put(e); }
    std::vector<char> b = slurp(in);
    int32_t n = *(int32_t *)b.data();
    const float *d = (const float *)(b.data() + 4);
    std::vector<float> uv;
    for (int i = 0; i < n; i++) {
        Ray r(Point3(d[7 * i], d[7 * i + 1], d[7 * i + 2]), Point3(d[7 * i + 3], d[7 * i + 4], d[7 * i + 5]));
        HitInfo hi; hi.Init(); hi.z = d[7 * i + 6];
        bool h = tobj->TriObj::IntersectRay(r, hi);
        o.put(rec(h, hi));
        uv.push_back(hi.uvw.x); uv.push_back(hi.uvw.y); uv.push_back(hi.uvw.z);
    }
    if (mtlout) {       // HitInfo::uvw after each ray (0.5,0.5,0.5 from Init() unless the triangle wrote it)
        Out u((std::string(mtlout) + ".uvw").c_str());
        if (!uv.empty()) u.bytes(uv.data(), uv.size() * sizeof(float));
    }
    return 0;
}

// in: int32 nops; nops x {int32 kind(0 scale,1 rotate,2 translate); float a[4]}  (rotate: axis xyz + degrees)
//     int32 n; n x {float ray[6]}; n x {float p[3], N[3]}
// out: float tm[9], itm[9], pos[3]; n x float ray_local[6]; n x {float p[3], N[3]}
static int cmd_node(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t nops = *(int32_t *)c; c += 4;
    Node node;
    for (int i = 0; i < nops; i++) {
        int32_t kind = *(int32_t *)c; c += 4;
        const float *a = (const float *)c; c += 16;
        if (kind == 0) node.Scale(a[0], a[1], a[2]);
        else if (kind == 1) { Point3 s(a[0], a[1], a[2]); s.Normalize(); node.Rotate(s, a[3]); }   // xmlload.cpp LoadTransform
        else node.Translate(Point3(a[0], a[1], a[2]));
    }
    int32_t n = *(int32_t *)c; c += 4;
    const float *rays = (const float *)c; c += (size_t)n * 24;
    const float *hits = (const float *)c;
    Out o(out);
    o.bytes(node.GetTransform().data, 36);
    o.bytes(node.GetInverseTransform().data, 36);
    o.bytes(&node.GetPosition().x, 12);
    for (int i = 0; i < n; i++) {
        Ray r(Point3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), Point3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
        Ray l = node.ToNodeCoords(r);
        o.put(l.p.x); o.put(l.p.y); o.put(l.p.z); o.put(l.dir.x); o.put(l.dir.y); o.put(l.dir.z);
    }
    for (int i = 0; i < n; i++) {
        HitInfo hi; hi.Init();
        hi.p = Point3(hits[6 * i], hits[6 * i + 1], hits[6 * i + 2]);
        hi.N = Point3(hits[6 * i + 3], hits[6 * i + 4], hits[6 * i + 5]);
        node.FromNodeCoords(hi);
        o.put(hi.p.x); o.put(hi.p.y); o.put(hi.p.z); o.put(hi.N.x); o.put(hi.N.y); o.put(hi.N.z);
    }
    return 0;
}

// in: int32 n; n x float rgb[3]    out: n x float halton2, n x float halton3 (index i), n x uint8 rgb[3]
static int cmd_misc(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    int32_t n = *(int32_t *)b.data();
    const float *d = (const float *)(b.data() + 4);
    Out o(out);
    for (int i = 0; i < n; i++) o.put(Halton(i, 2));
    for (int i = 0; i < n; i++) o.put(Halton(i, 3));
    for (int i = 0; i < n; i++) {
        Color24 c24 = (Color24)Color(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        o.put(c24.r); o.put(c24.g); o.put(c24.b);
    }
    return 0;
}

struct PM : public cy::PhotonMap {
    std::vector<Photon> &vec() { return photons; }
    int half() const { return halfStoredPhotons; }
};

template <int K> static void estimate(const PM &pm, float radius, const float *q, float *res)
{
    Color irr; Point3 dir;
    Point3 N(q[3], q[4], q[5]);
    pm.EstimateIrradiance<K>(irr, dir, radius, Point3(q[0], q[1], q[2]), &N, 1.f, cy::PhotonMap::FILTER_TYPE_CONSTANT);
    res[0] = irr.r; res[1] = irr.g; res[2] = irr.b; res[3] = dir.x; res[4] = dir.y; res[5] = dir.z;
}

// in: int32 np; np x {float pos[3], dir[3], power[3]}; float scale; int32 k; float radius;
//     int32 nq; nq x {float pos[3], normal[3]}
// out: np x 24B packed photons (after AddPhoton + ScalePhotonPowers(scale), before balancing);
//      (np+1) x 24B balanced array (index 0 unused); int32 halfStoredPhotons;
//      np x float decoded dir[3] + power rgb[3] of the packed photons; nq x float {irr[3], dir[3]}
static int cmd_photon(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t np = *(int32_t *)c; c += 4;
    const float *ph = (const float *)c; c += (size_t)np * 36;
    float scale = *(float *)c; c += 4;
    int32_t k = *(int32_t *)c; c += 4;
    float radius = *(float *)c; c += 4;
    int32_t nq = *(int32_t *)c; c += 4;
    const float *q = (const float *)c;
    PM pm;
    pm.AllocatePhotons(np);
    for (int i = 0; i < np; i++) {
        const float *p = ph + 9 * i;
        pm.AddPhoton(Point3(p[0], p[1], p[2]), Point3(p[3], p[4], p[5]), Color(p[6], p[7], p[8]));
    }
    // AddPhoton leaves bits 0-2 of planeAndDirZ uninitialised (a stack Photon); zero them so the
    // dump is reproducible -- SetPlane overwrites bits 0-1 for every internal node anyway.
    {
        std::vector<cy::PhotonMap::Photon> &v = pm.vec();
        for (size_t i = 0; i < v.size(); i++) { unsigned char *raw = (unsigned char *)&v[i]; raw[19] &= 0x8; }
        unsigned char *raw0 = (unsigned char *)&v[0]; memset(raw0, 0, 24);
    }
    pm.ScalePhotonPowers(scale);
    Out o(out);
    o.bytes(pm.GetPhotons(), (size_t)np * 24);
    std::vector<float> dec((size_t)np * 6);
    for (int i = 0; i < np; i++) {
        Point3 d; Color pw;
        pm[i].GetDirection(d); pm[i].GetPower(pw);
        dec[6 * i] = d.x; dec[6 * i + 1] = d.y; dec[6 * i + 2] = d.z; dec[6 * i + 3] = pw.r; dec[6 * i + 4] = pw.g; dec[6 * i + 5] = pw.b;
    }
    pm.PrepareForIrradianceEstimation();
    o.bytes(pm.vec().data(), (size_t)(np + 1) * 24);
    int32_t half = pm.half();
    o.put(half);
    o.bytes(dec.data(), dec.size() * 4);
    for (int i = 0; i < nq; i++) {
        float res[6];
        switch (k) {
        case 1: estimate<1>(pm, radius, q + 6 * i, res); break;
        case 8: estimate<8>(pm, radius, q + 6 * i, res); break;
        case 50: estimate<50>(pm, radius, q + 6 * i, res); break;
        case 100: estimate<100>(pm, radius, q + 6 * i, res); break;
        case 200: estimate<200>(pm, radius, q + 6 * i, res); break;
        case 400: estimate<400>(pm, radius, q + 6 * i, res); break;
        default: fprintf(stderr, "k=%d not instantiated\n", k); return 4;
        }
        o.bytes(res, sizeof res);
    }
    return 0;
}

// photondat <in.bin> <out.bin> <photonmap.dat>
// the reference's own photon dump (main.cpp:397-400: fwrite of Photon[NumPhotons]) read the way its viewer
// reads it (PhotonMapViz.cpp:172-193), put into a cyPhotonMap as is, balanced and queried.
// in: int32 k; float radius; int32 nq; nq x float[6] (pos, normal)
// out: int32 n; (n+1) x Photon balanced; int32 half; nq x float[6]
static int cmd_photondat(const char *in, const char *out, const char *dat)
{
    FILE *fp = fopen(dat, "rb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", dat); return 3; }
    int n = 0;
    cy::PhotonMap::Photon buffer;
    for (; !feof(fp); n++) { if (fread(&buffer, sizeof buffer, 1, fp) != 1) {} }
    n--;
    if (n <= 0) return 4;
    PM pm;
    pm.AllocatePhotons(n);
    pm.vec().resize(n + 1);
    rewind(fp);
    memset((void *)&pm.vec()[0], 0, sizeof buffer);
    if ((int)fread((void *)&pm.vec()[1], sizeof buffer, n, fp) != n) return 5;
    fclose(fp);
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t k = *(int32_t *)c; c += 4;
    float radius = *(float *)c; c += 4;
    int32_t nq = *(int32_t *)c; c += 4;
    const float *q = (const float *)c;
    pm.PrepareForIrradianceEstimation();
    Out o(out);
    int32_t n32 = n; o.put(n32);
    o.bytes(pm.vec().data(), (size_t)(n + 1) * 24);
    int32_t half = pm.half(); o.put(half);
    for (int i = 0; i < nq; i++) {
        float res[6];
        switch (k) {
        case 50: estimate<50>(pm, radius, q + 6 * i, res); break;
        case 100: estimate<100>(pm, radius, q + 6 * i, res); break;
        case 400: estimate<400>(pm, radius, q + 6 * i, res); break;
        default: fprintf(stderr, "k=%d not instantiated\n", k); return 4;
        }
        o.bytes(res, sizeof res);
    }
    return 0;
}

// Texture that returns the coordinate it is asked for: shows what TextureMap / SampleEnvironment compute
struct ProbeTexture : public Texture { Color Sample(const Point3 &uvw) const { return Color(uvw.x, uvw.y, uvw.z); } };

// tex <in.bin> <out.bin> <image.png|ppm>
// in: int32 nops; nops x {int32 kind; float a[4]} (TextureMap transform, as cmd_node); float c1[3], c2[3];
//     int32 n; n x float uvw[3]
// out: int32 w, h; w*h x float rgb[3] (texels, via Sample at texel origins);
//      n x float rgb[3] TextureFile::Sample(uvw); n x rgb TextureChecker::Sample(uvw);
//      n x float[3] TextureMap::TransformTo(uvw); n x float[3] coordinate SampleEnvironment(uvw as dir) looks up
static int cmd_tex(const char *in, const char *out, const char *image)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t nops = *(int32_t *)c; c += 4;
    ProbeTexture probe;
    TextureMap *map = new TextureMap(&probe);
    for (int i = 0; i < nops; i++) {
        int32_t kind = *(int32_t *)c; c += 4;
        const float *a = (const float *)c; c += 16;
        if (kind == 0) map->Scale(a[0], a[1], a[2]);
        else if (kind == 1) { Point3 s(a[0], a[1], a[2]); s.Normalize(); map->Rotate(s, a[3]); }
        else map->Translate(Point3(a[0], a[1], a[2]));
    }
    const float *cc = (const float *)c; c += 24;
    int32_t n = *(int32_t *)c; c += 4;
    const float *uvw = (const float *)c;
    TextureFile *tf = (TextureFile *)calloc(1, sizeof(TextureFile));   // no vtable exists for it (see header note)
    new ((void *)&((char *)tf)[0]) ItemBase();                            // name storage only
    tf->SetName(image);
    if (!tf->TextureFile::Load()) { fprintf(stderr, "cannot load %s\n", image); return 3; }
    // width/height are private: recover them by probing the tiling period is overkill -- lodepng again
    unsigned w = 0, h = 0;
    { std::vector<unsigned char> d; lodepng::decode(d, w, h, image, LCT_RGB); }
    Out o(out);
    int32_t wi = (int32_t)w, hi = (int32_t)h;
    o.put(wi); o.put(hi);
    for (unsigned y = 0; y < h; y++) for (unsigned x = 0; x < w; x++) {
        Color t = tf->TextureFile::Sample(Point3((x + 0.0f) / w, (y + 0.0f) / h, 0));
        o.put(t.r); o.put(t.g); o.put(t.b);
    }
    for (int i = 0; i < n; i++) { Color t = tf->TextureFile::Sample(Point3(uvw[3 * i], uvw[3 * i + 1], uvw[3 * i + 2])); o.put(t.r); o.put(t.g); o.put(t.b); }
    TextureChecker *tc = (TextureChecker *)calloc(1, sizeof(TextureChecker));
    tc->SetColor1(Color(cc[0], cc[1], cc[2])); tc->SetColor2(Color(cc[3], cc[4], cc[5]));
    for (int i = 0; i < n; i++) { Color t = tc->TextureChecker::Sample(Point3(uvw[3 * i], uvw[3 * i + 1], uvw[3 * i + 2])); o.put(t.r); o.put(t.g); o.put(t.b); }
    for (int i = 0; i < n; i++) { Point3 t = map->TransformTo(Point3(uvw[3 * i], uvw[3 * i + 1], uvw[3 * i + 2])); o.put(t.x); o.put(t.y); o.put(t.z); }
    TexturedColor env(1, 1, 1);
    env.SetTexture(new TextureMap(&probe));
    for (int i = 0; i < n; i++) { Color t = env.SampleEnvironment(Point3(uvw[3 * i], uvw[3 * i + 1], uvw[3 * i + 2])); o.put(t.r); o.put(t.g); o.put(t.b); }
    return 0;
}


// ---- RenderImage side products (scene.h:591-637) ----------------------------------------------
// zimg <in.bin> <out.bin>
// in: int32 w, h; float z[w*h]; uint8 sampleCount[w*h]
// out: uint8 zbufferImg[w*h]; uint8 sampleCountImg[w*h]; int32 smax (return value of ComputeSampleCountImage)
static int cmd_zimg(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t w = *(int32_t *)c; c += 4;
    int32_t h = *(int32_t *)c; c += 4;
    const size_t n = (size_t)w * h;
    RenderImage img;
    img.Init(w, h);
    memcpy(img.GetZBuffer(), c, n * 4); c += n * 4;
    memcpy(img.GetSampleCount(), c, n);
    img.ComputeZBufferImage();
    int32_t smax = img.ComputeSampleCountImage();
    Out o(out);
    o.bytes(img.GetZBufferImage(), n);
    o.bytes(img.GetSampleCountImage(), n);
    o.put(smax);
    return 0;
}

// ---- the recording double for GenLight::Shadow (see the file header) ---------------------------
static std::vector<float> g_shadow_log;       // 7 floats per call: ray.p, ray.dir, t_max
static std::vector<float> g_shadow_script;    // values returned, in call order (cycled)
static size_t g_shadow_calls = 0;
float GenLight::Shadow(Ray ray, float t_max)
{
    const float v[7] = {ray.p.x, ray.p.y, ray.p.z, ray.dir.x, ray.dir.y, ray.dir.z, t_max};
    g_shadow_log.insert(g_shadow_log.end(), v, v + 7);
    const float r = g_shadow_script.empty() ? 1.0f : g_shadow_script[g_shadow_calls % g_shadow_script.size()];
    g_shadow_calls++;
    return r;
}

#define RAND_CAPTURE 64
// illum <in.bin> <out.bin>
// in: int32 n; n x {float intensity[3], position[3], size, p[3]; uint32 seed; int32 nscript; float script[8]}
// out: n x {float result[3]; int32 ncalls; float log[20][7] (first ncalls rows used); int32 rand[RAND_CAPTURE]}
static int cmd_illum(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t n = *(int32_t *)c; c += 4;
    Out o(out);
    for (int i = 0; i < n; i++) {
        const float *f = (const float *)c; c += 40;
        uint32_t seed = *(uint32_t *)c; c += 4;
        int32_t nscript = *(int32_t *)c; c += 4;
        const float *script = (const float *)c; c += 32;
        PointLight pl;
        pl.SetIntensity(Color(f[0], f[1], f[2]));
        pl.SetPosition(Point3(f[3], f[4], f[5]));
        pl.SetSize(f[6]);
        int32_t raw[RAND_CAPTURE];
        srand(seed);
        for (int k = 0; k < RAND_CAPTURE; k++) raw[k] = rand();
        srand(seed);
        g_shadow_log.clear(); g_shadow_calls = 0;
        g_shadow_script.assign(script, script + nscript);
        Color r = pl.PointLight::Illuminate(Point3(f[7], f[8], f[9]), Point3(0, 0, 1));
        o.put(r.r); o.put(r.g); o.put(r.b);
        int32_t nc = (int32_t)g_shadow_calls; o.put(nc);
        g_shadow_log.resize(20 * 7, 0.0f);
        o.bytes(g_shadow_log.data(), 20 * 7 * 4);
        o.bytes(raw, sizeof raw);
    }
    return 0;
}

// lights <in.bin> <out.bin>: AmbientLight / DirectLight::Illuminate and every Light::Direction (lights.h:28-57, :159)
// in: int32 n; n x {int32 type (0 ambient, 1 direct, 2 point); float intensity[3], vec[3] (direction or position), p[3], shadow}
// out: n x {float illum[3] (point lights: zeros, see `illum`), dir[3]; int32 ncalls; float log[7]}
static int cmd_lights(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c = b.data();
    int32_t n = *(int32_t *)c; c += 4;
    Out o(out);
    for (int i = 0; i < n; i++) {
        int32_t type = *(int32_t *)c; c += 4;
        const float *f = (const float *)c; c += 40;
        g_shadow_log.clear(); g_shadow_calls = 0;
        g_shadow_script.assign(1, f[9]);
        Color r(0, 0, 0); Point3 d(0, 0, 0);
        const Point3 p(f[6], f[7], f[8]), N(0, 0, 1);
        if (type == 0) { AmbientLight l; l.SetIntensity(Color(f[0], f[1], f[2])); r = l.AmbientLight::Illuminate(p, N); d = l.AmbientLight::Direction(p); }
        else if (type == 1) { DirectLight l; l.SetIntensity(Color(f[0], f[1], f[2])); l.SetDirection(Point3(f[3], f[4], f[5])); r = l.DirectLight::Illuminate(p, N); d = l.DirectLight::Direction(p); }
        else { PointLight l; l.SetPosition(Point3(f[3], f[4], f[5])); d = l.PointLight::Direction(p); }
        o.put(r.r); o.put(r.g); o.put(r.b); o.put(d.x); o.put(d.y); o.put(d.z);
        int32_t nc = (int32_t)g_shadow_calls; o.put(nc);
        g_shadow_log.resize(7, 0.0f);
        o.bytes(g_shadow_log.data(), 28);
    }
    return 0;
}

#ifdef REF_FIN
// pbounce <in.bin> <out.bin>   (RayTracingFinal only: P13's materials.h has no body for it)
// in: int32 n; n x {float diffuse[3], specular[3], reflection[3], refraction[3], absorption[3], glossiness, ior;
//                   float ray_p[3], ray_dir[3]; float hit_p[3], hit_N[3], hit_z; int32 front; float c[3]; uint32 seed;
//                   float reflectionGlossiness, refractionGlossiness}
// out: n x {int32 ret; float ray_p[3], ray_dir[3], c[3]; int32 rand[8]}; then n x float atten[3] =
//      Attenuation(absorption, hit_z); then n x float Nt[3], Nb[3] = createCoordinateSystem(hit_N)
static int cmd_pbounce(const char *in, const char *out)
{
    std::vector<char> b = slurp(in);
    const char *c0 = b.data();
    int32_t n = *(int32_t *)c0;
    const size_t rec = 17 * 4 + 6 * 4 + 7 * 4 + 4 + 3 * 4 + 4 + 8;
    Out o(out);
    std::vector<float> att, cs;
    for (int i = 0; i < n; i++) {
        const char *c = c0 + 4 + rec * i;
        const float *m = (const float *)c; c += 17 * 4;
        const float *ry = (const float *)c; c += 24;
        const float *ht = (const float *)c; c += 28;
        int32_t front = *(int32_t *)c; c += 4;
        const float *col = (const float *)c; c += 12;
        uint32_t seed = *(uint32_t *)c; c += 4;
        const float *gl = (const float *)c;
        MtlBlinn mtl;
        mtl.SetReflectionGlossiness(gl[0]); mtl.SetRefractionGlossiness(gl[1]);
        mtl.SetDiffuse(Color(m[0], m[1], m[2])); mtl.SetSpecular(Color(m[3], m[4], m[5]));
        mtl.SetReflection(Color(m[6], m[7], m[8])); mtl.SetRefraction(Color(m[9], m[10], m[11]));
        mtl.SetAbsorption(Color(m[12], m[13], m[14])); mtl.SetGlossiness(m[15]); mtl.SetRefractionIndex(m[16]);
        Ray r(Point3(ry[0], ry[1], ry[2]), Point3(ry[3], ry[4], ry[5]));
        HitInfo hi; hi.Init();
        hi.p = Point3(ht[0], ht[1], ht[2]); hi.N = Point3(ht[3], ht[4], ht[5]); hi.z = ht[6]; hi.front = front != 0;
        Color cc(col[0], col[1], col[2]);
        int32_t raw[8];
        srand(seed);
        for (int k = 0; k < 8; k++) raw[k] = rand();
        srand(seed);
        int32_t ret = mtl.MtlBlinn::RandomPhotonBounce(r, cc, hi) ? 1 : 0;
        o.put(ret);
        o.put(r.p.x); o.put(r.p.y); o.put(r.p.z); o.put(r.dir.x); o.put(r.dir.y); o.put(r.dir.z);
        o.put(cc.r); o.put(cc.g); o.put(cc.b);
        o.bytes(raw, sizeof raw);
        Color a = Attenuation(Color(m[12], m[13], m[14]), ht[6]);
        att.push_back(a.r); att.push_back(a.g); att.push_back(a.b);
        Point3 Nt, Nb;
        createCoordinateSystem(hi.N, Nt, Nb);
        const float q[6] = {Nt.x, Nt.y, Nt.z, Nb.x, Nb.y, Nb.z};
        cs.insert(cs.end(), q, q + 6);
    }
    if (n) { o.bytes(att.data(), att.size() * 4); o.bytes(cs.data(), cs.size() * 4); }
    return 0;
}
#endif

// time <in.bin> <out.bin>: same input as `photon`; prints seconds spent in the nq queries (k=400)
static int cmd_time(const char *in, const char *out)
{
    (void)out;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ref_harness <prims|box|mesh|node|misc|photon> in out [obj]\n"); return 1; }
    std::string cmd = argv[1];
    if (cmd == "prims") return cmd_prims(argv[2], argv[3]);
    if (cmd == "box") return cmd_box(argv[2], argv[3]);
    if (cmd == "mesh") { if (argc < 5) return 1; return cmd_mesh(argv[2], argv[3], argv[4]); }
    if (cmd == "meshmtl") { if (argc < 6) return 1; return cmd_mesh(argv[2], argv[3], argv[4], true, argv[5]); }
    if (cmd == "node") return cmd_node(argv[2], argv[3]);
    if (cmd == "misc") return cmd_misc(argv[2], argv[3]);
    if (cmd == "photon") return cmd_photon(argv[2], argv[3]);
    if (cmd == "photondat") { if (argc < 5) return 1; return cmd_photondat(argv[2], argv[3], argv[4]); }
    if (cmd == "tex") { if (argc < 5) return 1; return cmd_tex(argv[2], argv[3], argv[4]); }
    if (cmd == "zimg") return cmd_zimg(argv[2], argv[3]);
    if (cmd == "illum") return cmd_illum(argv[2], argv[3]);
    if (cmd == "lights") return cmd_lights(argv[2], argv[3]);
#ifdef REF_FIN
    if (cmd == "pbounce") return cmd_pbounce(argv[2], argv[3]);
#endif
    if (cmd == "time") return cmd_time(argv[2], argv[3]);
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 1;
}
