// rt_scene.cpp -- host scene model: math, OBJ loading, BVH build, photon balancing, lowering.
// Each routine names the reference code whose behaviour it reproduces (FIN = /root/reference/
// RayTracingFinal/RayTracingFinal).  Compiled with -ffp-contract=off: transforms, BVH boxes and
// kd-tree layouts must come out bit-identical to the reference's (tests/test_host_*.py).
#include "rt_scene.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>

namespace rt {

// ---- Matrix3 --------------------------------------------------------------------------------
// cyMatrix3::SetRotation(axis, angle), FIN/include/cyMatrix.h:412-430
void Matrix3::SetRotation(const Point3 &axis, float angle)
{
    const float s = sinf(angle), c = cosf(angle);
    const float t = 1.0f - c;
    const float tx = t * axis.x, ty = t * axis.y, tz = t * axis.z;
    const float txy = tx * axis.y, txz = tx * axis.z, tyz = ty * axis.z;
    const float sx = s * axis.x, sy = s * axis.y, sz = s * axis.z;
    data[0] = tx * axis.x + c; data[1] = txy + sz;        data[2] = txz - sy;
    data[3] = txy - sz;        data[4] = ty * axis.y + c; data[5] = tyz + sx;
    data[6] = txz + sy;        data[7] = tyz - sx;        data[8] = tz * axis.z + c;
}

// cyMatrix3::GetInverse, FIN/include/cyMatrix.h:612-633 (adjugate, then each entry / det)
void Matrix3::GetInverse(Matrix3 &inv) const
{
    const float *d = data;
    inv.data[0] = d[4] * d[8] - d[5] * d[7];
    inv.data[1] = d[2] * d[7] - d[1] * d[8];
    inv.data[2] = d[1] * d[5] - d[2] * d[4];
    inv.data[3] = d[5] * d[6] - d[3] * d[8];
    inv.data[4] = d[0] * d[8] - d[2] * d[6];
    inv.data[5] = d[2] * d[3] - d[0] * d[5];
    inv.data[6] = d[3] * d[7] - d[4] * d[6];
    inv.data[7] = d[1] * d[6] - d[0] * d[7];
    inv.data[8] = d[0] * d[4] - d[1] * d[3];
    const float det = d[0] * inv.data[0] + d[1] * inv.data[3] + d[2] * inv.data[6];
    for (float &e : inv.data) e = e / det;
}

// ---- lowering of lights / materials ----------------------------------------------------------
static void put3(float *o, const Color &c) { o[0] = c.r; o[1] = c.g; o[2] = c.b; }
static void put3(float *o, const Point3 &p) { o[0] = p.x; o[1] = p.y; o[2] = p.z; }

void AmbientLight::Lower(rt_light &o) const { memset(&o, 0, sizeof o); o.type = RT_LIGHT_AMBIENT; put3(o.intensity, intensity); }
void DirectLight::Lower(rt_light &o) const { memset(&o, 0, sizeof o); o.type = RT_LIGHT_DIRECT; put3(o.intensity, intensity); put3(o.direction, direction); }
void PointLight::Lower(rt_light &o) const { memset(&o, 0, sizeof o); o.type = RT_LIGHT_POINT; put3(o.intensity, intensity); put3(o.position, position); o.size = size; }

bool TextureFile::Load(std::string *err)
{
    data.clear(); width = height = 0;
    return ReadImageRGB(name.c_str(), width, height, data, err);
}
bool TextureFile::Lower(rt_texture &o, std::vector<uint8_t> &texels) const
{
    memset(&o, 0, sizeof o);
    o.type = RT_TEX_FILE; o.width = width; o.height = height; o.texel_offset = (uint32_t)texels.size();
    texels.insert(texels.end(), data.begin(), data.end());
    while (texels.size() % 4) texels.push_back(0);
    return true;
}
bool TextureChecker::Lower(rt_texture &o, std::vector<uint8_t> &) const
{
    memset(&o, 0, sizeof o);
    o.type = RT_TEX_CHECKER; put3(o.color1, color1); put3(o.color2, color2);
    return true;
}

bool MtlBlinn::Lower(rt_blinn &o) const
{
    put3(o.diffuse, diffuse.GetColor()); put3(o.specular, specular.GetColor()); put3(o.reflection, reflection);
    put3(o.refraction, refraction); put3(o.emission, emission); put3(o.absorption, absorption);
    o.glossiness = glossiness; o.ior = ior;
    o.reflection_glossiness = reflectionGlossiness; o.refraction_glossiness = refractionGlossiness;
    return true;
}

bool MultiMtl::Lower(rt_blinn &o) const
{
    if (mtls.empty()) return false;
    return mtls[0]->Lower(o);
}

// ---- OBJ loading -----------------------------------------------------------------------------
// Behaviour of cyTriMesh::LoadFromFileObj (FIN/include/cyTriMesh.h:263-462) for geometry:
// whitespace-collapsed lines of at most 1023 characters, '#' comment lines, v / vt / vn /
// f records, polygons fan-triangulated around their first vertex, 1-based or negative
// (relative) indices, "v", "v/vt", "v//vn", "v/vt/vn" corner forms.  Normal faces exist from
// the first `vn` line or the first corner with a normal index onwards, texture faces likewise
// from the first `vt`.  With loadMtl (the node names no scene material, xmlload.cpp:204) `usemtl`
// groups the faces by material in order of first use and `mtllib` files are read for
// Kd/Ks/Tf/Ns/Ni/illum/map_Kd/map_Ks (:437-447, 466-544).
namespace {
struct LineReader {
    FILE *fp;
    char data[1024];
    int ReadLine()
    {
        int c = fgetc(fp);
        while (c != EOF) {
            while (c != EOF && isspace(c)) c = fgetc(fp);
            if (c == '#') { while (c != EOF && c != '\n' && c != '\r' && c != '\0') c = fgetc(fp); }
            else break;
        }
        int i = 0;
        bool inspace = false;
        while (i < 1024 - 1) {
            if (c == EOF || c == '\n' || c == '\r' || c == '\0') break;
            if (isspace(c)) inspace = true;
            else {
                if (inspace) data[i++] = ' ';
                inspace = false;
                data[i++] = (char)c;
            }
            c = fgetc(fp);
        }
        data[i] = '\0';
        return i;
    }
    bool IsCommand(const char *cmd) const
    {
        int i = 0;
        while (cmd[i] != '\0') { if (cmd[i] != data[i]) return false; i++; }
        return data[i] == '\0' || data[i] == ' ';
    }
};
}  // namespace

bool TriObj::LoadFromFileObj(const char *filename, bool loadMtl, std::string *err)
{
    FILE *fp = fopen(filename, "r");
    if (!fp) { if (err) *err = std::string("cannot open OBJ file ") + filename; return false; }
    v.clear(); vn.clear(); vt.clear(); f.clear(); fn.clear(); ft.clear(); mtls.clear(); mcfc.clear();
    LineReader in; in.fp = fp;
    std::vector<uint32_t> rf, rfn, rft;     // faces in file order
    std::vector<int> faceMtl;               // material of every triangle (-1: none)
    struct MtlData { std::string name; uint32_t firstFace = 0, faceCount = 0; };
    std::vector<MtlData> mtlList;
    std::vector<std::string> mtlFiles;
    int currentMtl = -1;
    bool hasNormals = false, hasTextures = false;
    while (int rb = in.ReadLine()) {
        if (in.IsCommand("v") || in.IsCommand("vt") || in.IsCommand("vn")) {
            float p[3] = {0, 0, 0};
            if (rb > 2) sscanf(in.data + 2, "%f %f %f", &p[0], &p[1], &p[2]);
            std::vector<float> &dst = in.data[1] == 't' ? vt : (in.data[1] == 'n' ? vn : v);
            dst.insert(dst.end(), p, p + 3);
            if (in.data[1] == 'n') hasNormals = true;
            if (in.data[1] == 't') hasTextures = true;
        } else if (in.IsCommand("f")) {
            int facevert = -1;
            bool inspace = true, negative = false;
            int type = 0;
            uint32_t index = 0;
            uint32_t face[3] = {0, 0, 0}, nface[3] = {0, 0, 0}, tface[3] = {0, 0, 0};
            const size_t before = rf.size() / 3;
            auto emit = [&]() {
                rf.insert(rf.end(), face, face + 3);
                if (hasTextures) rft.insert(rft.end(), tface, tface + 3);
                if (hasNormals) rfn.insert(rfn.end(), nface, nface + 3);
                faceMtl.push_back(currentMtl);
            };
            for (int i = 2; i < rb; i++) {
                const char ch = in.data[i];
                if (ch == ' ') { inspace = true; continue; }
                if (inspace) {
                    inspace = false; negative = false; type = 0; index = 0;
                    if (facevert < 2) {
                        if (facevert == -1) for (int k = 0; k < 3; k++) face[k] = nface[k] = tface[k] = 0;
                        facevert++;
                    } else {
                        // emit the triangle gathered so far and keep (v0, last) for the fan
                        emit();
                        face[1] = face[2]; tface[1] = tface[2]; nface[1] = nface[2];
                    }
                }
                if (ch == '/') { type++; index = 0; }
                if (ch == '-') negative = true;
                if (ch >= '0' && ch <= '9') {
                    index = index * 10 + (uint32_t)(ch - '0');
                    switch (type) {
                    case 0: face[facevert] = negative ? (uint32_t)(v.size() / 3) - index : index - 1; break;
                    case 1: tface[facevert] = negative ? (uint32_t)(vt.size() / 3) - index : index - 1; hasTextures = true; break;
                    case 2: nface[facevert] = negative ? (uint32_t)(vn.size() / 3) - index : index - 1; hasNormals = true; break;
                    }
                }
            }
            emit();
            if (currentMtl >= 0) mtlList[currentMtl].faceCount += (uint32_t)(rf.size() / 3 - before);
        } else if (loadMtl) {
            if (in.IsCommand("usemtl")) {
                // MtlList::CreateMtl (cyTriMesh.h:335-345); an empty name means material 0 there, which
                // is only defined once a material exists
                const char *nm = rb > 7 ? in.data + 7 : "";
                int idx = -1;
                for (size_t i = 0; i < mtlList.size(); i++) if (mtlList[i].name == nm) { idx = (int)i; break; }
                if (nm[0] == '\0') idx = mtlList.empty() ? -1 : 0;
                else if (idx < 0) { MtlData m; m.name = nm; m.firstFace = (uint32_t)(rf.size() / 3); mtlList.push_back(m); idx = (int)mtlList.size() - 1; }
                currentMtl = idx;
            }
            if (in.IsCommand("mtllib") && rb > 7) mtlFiles.push_back(in.data + 7);
        }
        if (feof(fp)) break;
    }
    fclose(fp);
    const size_t nfaces = rf.size() / 3;
    if (vn.empty()) rfn.clear();
    if (vt.empty()) rft.clear();
    // the reference sizes fn/ft to the face count once normals/texture vertices exist and copies the
    // faces gathered so far to their front; the tail (faces read before the first one) is
    // uninitialised there and zero here
    if (!vn.empty() && rfn.size() < rf.size()) rfn.insert(rfn.end(), rf.size() - rfn.size(), 0u);
    if (!vt.empty() && rft.size() < rf.size()) rft.insert(rft.end(), rf.size() - rft.size(), 0u);
    if (!mtlList.empty()) {
        // faces regrouped by material, materials in order of first use, unassigned faces last
        // (cyTriMesh.h:466-491)
        f.reserve(rf.size());
        auto take = [&](size_t i) {
            f.insert(f.end(), rf.begin() + 3 * i, rf.begin() + 3 * i + 3);
            if (!rfn.empty()) fn.insert(fn.end(), rfn.begin() + 3 * i, rfn.begin() + 3 * i + 3);
            if (!rft.empty()) ft.insert(ft.end(), rft.begin() + 3 * i, rft.begin() + 3 * i + 3);
        };
        for (size_t m = 0; m < mtlList.size(); m++) {
            for (size_t i = mtlList[m].firstFace, j = 0; j < mtlList[m].faceCount && i < nfaces; i++)
                if (faceMtl[i] == (int)m) { take(i); j++; }
            mcfc.push_back((uint32_t)(f.size() / 3));
        }
        if (f.size() / 3 < nfaces) for (size_t i = 0; i < nfaces; i++) if (faceMtl[i] < 0) take(i);
        if (f.size() != rf.size()) { if (err) *err = "OBJ material grouping lost faces"; return false; }
    } else { f.swap(rf); fn.swap(rfn); ft.swap(rft); }
    for (uint32_t idx : f) if (idx >= v.size() / 3) { if (err) *err = "OBJ face index out of range"; return false; }
    for (uint32_t idx : fn) if (idx >= vn.size() / 3) { if (err) *err = "OBJ normal index out of range"; return false; }
    for (uint32_t idx : ft) if (idx >= vt.size() / 3) { if (err) *err = "OBJ texture index out of range"; return false; }

    // the .mtl files, looked up next to the OBJ (cyTriMesh.h:497-544)
    if (loadMtl) {
        mtls.resize(mtlList.size());
        std::string base = filename;
        size_t cut = base.find_last_of('\\');
        if (cut == std::string::npos) cut = base.find_last_of('/');
        base = cut == std::string::npos ? std::string() : base.substr(0, cut + 1);
        for (const std::string &lib : mtlFiles) {
            FILE *mf = fopen((base + lib).c_str(), "r");
            if (!mf) { fprintf(stderr, "rt_mi355x: cannot open material library \"%s\"\n", (base + lib).c_str()); continue; }
            LineReader ml; ml.fp = mf;
            int id = -1;
            auto f3 = [&](float *dst) {     // Buffer::ReadFloat3: one value means grey
                dst[0] = dst[1] = dst[2] = 0;
                const int n = sscanf(ml.data + 2, "%f %f %f", &dst[0], &dst[1], &dst[2]);
                if (n == 1) dst[2] = dst[1] = dst[0];
            };
            auto str = [&](int start, int len) {
                std::string r;
                if (len > start) { int st = start; while (ml.data[st] != '\0' && ml.data[st] <= ' ') st++; r = ml.data + st; }
                return r;
            };
            while (int rb = ml.ReadLine()) {
                if (ml.IsCommand("newmtl")) {
                    id = -1;
                    const char *nm = rb > 7 ? ml.data + 7 : "";
                    for (size_t i = 0; i < mtlList.size(); i++) if (mtlList[i].name == nm) { id = (int)i; break; }
                    if (id >= 0) mtls[id].name = nm;
                } else if (id >= 0) {
                    ObjMtl &m = mtls[id];
                    if (ml.IsCommand("Kd")) f3(m.Kd);
                    else if (ml.IsCommand("Ks")) f3(m.Ks);
                    else if (ml.IsCommand("Tf")) f3(m.Tf);
                    else if (ml.IsCommand("Ns")) sscanf(ml.data + 2, "%f", &m.Ns);
                    else if (ml.IsCommand("Ni")) sscanf(ml.data + 2, "%f", &m.Ni);
                    else if (ml.IsCommand("illum") && rb > 5) sscanf(ml.data + 5, "%d", &m.illum);
                    else if (ml.IsCommand("map_Kd")) m.map_Kd = str(7, rb);
                    else if (ml.IsCommand("map_Ks")) m.map_Ks = str(7, rb);
                }
                if (feof(mf)) break;
            }
            fclose(mf);
        }
    }
    return true;
}

// cyTriMesh::ComputeNormals, FIN/include/cyTriMesh.h:248-261
void TriObj::ComputeNormals()
{
    vn.assign(v.size(), 0.0f);
    fn = f;
    auto P = [&](uint32_t i) { return Point3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); };
    for (size_t i = 0; i < f.size(); i += 3) {
        Point3 N = (P(f[i + 1]) - P(f[i])) ^ (P(f[i + 2]) - P(f[i]));
        for (int k = 0; k < 3; k++) {
            float *d = &vn[3 * f[i + k]];
            d[0] += N.x; d[1] += N.y; d[2] += N.z;
        }
    }
    for (size_t i = 0; i < vn.size(); i += 3) {
        Point3 n(vn[i], vn[i + 1], vn[i + 2]);
        n.Normalize();
        vn[i] = n.x; vn[i + 1] = n.y; vn[i + 2] = n.z;
    }
}

void TriObj::BuildBVH(unsigned maxElementsPerNode)
{
    BuildMeanSplitBVH(v.data(), f.data(), NF(), maxElementsPerNode, nodes, elements);
}

// TriObj::Load, FIN/include/objects.h:137-145
bool TriObj::Load(const char *filename, bool loadMtl, std::string *err)
{
    nodes.clear(); elements.clear();
    if (!LoadFromFileObj(filename, loadMtl, err)) return false;
    if (!HasNormals()) ComputeNormals();
    BuildBVH(4);
    return true;
}

// ---- cyBVH build (mean split) ------------------------------------------------------------------
// FIN/include/cyBVH.h: Build :122-142, SplitTempNode :242-278, MeanSplit :295-328,
// ConvertTempData :281-291.  Node ids: root 1, the two children of a node adjacent, the left
// subtree's nodes numbered before the right subtree's.
namespace {
struct BuildCtx {
    const float *v; const uint32_t *f; unsigned maxPer;
    std::vector<uint32_t> *elements;
    struct Tmp { float box[6]; unsigned count, offset; int c1, c2; };
    std::vector<Tmp> tmp;

    void Bounds(uint32_t e, float b[6]) const
    {
        const uint32_t *fi = f + 3 * (size_t)e;
        const float *p = v + 3 * (size_t)fi[0];
        b[0] = b[3] = p[0]; b[1] = b[4] = p[1]; b[2] = b[5] = p[2];
        for (int j = 1; j < 3; j++) {
            p = v + 3 * (size_t)fi[j];
            for (int k = 0; k < 3; k++) { if (b[k] > p[k]) b[k] = p[k]; if (b[k + 3] < p[k]) b[k + 3] = p[k]; }
        }
    }
    float Center(uint32_t e, unsigned dim) const
    {
        const uint32_t *fi = f + 3 * (size_t)e;
        return (v[3 * (size_t)fi[0] + dim] + v[3 * (size_t)fi[1] + dim] + v[3 * (size_t)fi[2] + dim]) / 3.0f;
    }
    static void Grow(float b[6], const float o[6])
    {
        for (int i = 0; i < 3; i++) { if (b[i] > o[i]) b[i] = o[i]; if (b[i + 3] < o[i + 3]) b[i + 3] = o[i + 3]; }
    }
    static void Empty(float b[6]) { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }

    unsigned MeanSplit(unsigned count, uint32_t *ne, const float *box) const
    {
        if (count <= maxPer) return 0;
        const float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
        unsigned sd[3];
        sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        sd[1] = (sd[0] + 1) % 3;
        sd[2] = (sd[0] + 2) % 3;
        if (d[sd[1]] < d[sd[2]]) std::swap(sd[1], sd[2]);
        for (int s = 0; s < 3; s++) {
            const unsigned dim = sd[s];
            const float splitPos = 0.5f * (box[dim] + box[dim + 3]);
            unsigned i = 0, j = count;
            while (i < j) {
                if (Center(ne[i], dim) <= splitPos) i++;
                else { j--; std::swap(ne[i], ne[j]); }
            }
            if (i < count && i > 0) return i;
        }
        return 0;
    }
    void Split(int t)
    {
        uint32_t *ne = elements->data() + tmp[t].offset;
        unsigned c1 = MeanSplit(tmp[t].count, ne, tmp[t].box);
        if (c1 == 0 || c1 >= tmp[t].count) {
            if (tmp[t].count > 8) c1 = tmp[t].count / 2;     // CY_BVH_MAX_ELEMENT_COUNT
            else return;
        }
        Tmp a, b;
        Empty(a.box); Empty(b.box);
        float eb[6];
        for (unsigned i = 0; i < c1; i++) { Bounds(ne[i], eb); Grow(a.box, eb); }
        for (unsigned i = c1; i < tmp[t].count; i++) { Bounds(ne[i], eb); Grow(b.box, eb); }
        a.count = c1; a.offset = tmp[t].offset; a.c1 = a.c2 = -1;
        b.count = tmp[t].count - c1; b.offset = tmp[t].offset + c1; b.c1 = b.c2 = -1;
        const int ia = (int)tmp.size(); tmp.push_back(a);
        const int ib = (int)tmp.size(); tmp.push_back(b);
        tmp[t].c1 = ia; tmp[t].c2 = ib;
        Split(ia);
        Split(ib);
    }
    unsigned Convert(std::vector<rt_bvh_node> &nodes, unsigned id, int t, unsigned childIndex) const
    {
        memcpy(nodes[id].box, tmp[t].box, sizeof(float) * 6);
        if (tmp[t].c1 < 0) {
            nodes[id].data = (tmp[t].offset & 0x0FFFFFFFu) | ((tmp[t].count - 1) << 28) | 0x80000000u;
            return childIndex;
        }
        nodes[id].data = childIndex & 0x7FFFFFFFu;
        const unsigned next = Convert(nodes, childIndex, tmp[t].c1, childIndex + 2);
        return Convert(nodes, childIndex + 1, tmp[t].c2, next);
    }
};
}  // namespace

void BuildMeanSplitBVH(const float *v, const uint32_t *f, unsigned nf, unsigned maxPerLeaf,
                       std::vector<rt_bvh_node> &nodes, std::vector<uint32_t> &elements)
{
    nodes.clear(); elements.clear();
    if (nf == 0) return;
    BuildCtx c;
    c.v = v; c.f = f; c.maxPer = std::min(maxPerLeaf, 8u); c.elements = &elements;
    elements.resize(nf);
    for (unsigned i = 0; i < nf; i++) elements[i] = i;
    BuildCtx::Tmp root;
    BuildCtx::Empty(root.box);
    float eb[6];
    for (unsigned i = 0; i < nf; i++) { c.Bounds(i, eb); BuildCtx::Grow(root.box, eb); }
    root.count = nf; root.offset = 0; root.c1 = root.c2 = -1;
    c.tmp.reserve(2 * (size_t)nf);
    c.tmp.push_back(root);
    c.Split(0);
    nodes.assign(c.tmp.size() + 1, rt_bvh_node{});
    c.Convert(nodes, 1, 0, 2);
}

// ---- photon map balancing ------------------------------------------------------------------------
// PhotonMap::PrepareForIrradianceEstimation + BalanceSegment, FIN/include/cyPhotonMap.h:196-284:
// left-balanced kd-tree in heap order (children of i at 2i, 2i+1), split axis = widest extent
// of the segment's box, quick-select partition that permutes the input array in place.
namespace {
struct Balancer {
    rt_photon *ph, *bal;
    void Segment(float bmin[3], float bmax[3], uint32_t index, uint32_t start, uint32_t end)
    {
        uint32_t median = 1;
        while (4 * median <= end - start + 1) median += median;
        if (3 * median <= end - start + 1) { median += median; median += start - 1; }
        else median = end - median + 1;
        int axis = 2;
        const float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
        if (dx > dy) { if (dx > dz) axis = 0; }
        else if (dy > dz) axis = 1;
        uint32_t left = start, right = end;
        while (right > left) {
            const float pivot = ph[right].position[axis];
            uint32_t i = left - 1, j = right;
            while (ph[++i].position[axis] < pivot) {}
            while (ph[--j].position[axis] > pivot && j > left) {}
            while (i < j) {
                std::swap(ph[i], ph[j]);
                while (ph[++i].position[axis] < pivot) {}
                while (ph[--j].position[axis] > pivot && j > left) {}
            }
            std::swap(ph[i], ph[right]);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
        bal[index] = ph[median];
        bal[index].plane_and_dirz = (uint8_t)((bal[index].plane_and_dirz & 0x8) | (axis & 0x3));
        const float split = bal[index].position[axis];
        if (median > start) {
            if (start < median - 1) {
                float tmax[3] = {bmax[0], bmax[1], bmax[2]};
                tmax[axis] = split;
                Segment(bmin, tmax, 2 * index, start, median - 1);
            } else bal[2 * index] = ph[start];
        }
        if (median < end) {
            if (median + 1 < end) {
                float tmin[3] = {bmin[0], bmin[1], bmin[2]};
                tmin[axis] = split;
                Segment(tmin, bmax, 2 * index + 1, median + 1, end);
            } else bal[2 * index + 1] = ph[end];
        }
    }
};
}  // namespace

void BalancePhotons(rt_photon *in, uint32_t n, rt_photon *out)
{
    // the reference's box loop starts at photons[0], the unused slot (cyPhotonMap.h:201-210)
    float bmin[3] = {in[0].position[0], in[0].position[1], in[0].position[2]};
    float bmax[3] = {bmin[0], bmin[1], bmin[2]};
    for (uint32_t i = 1; i <= n; i++)
        for (int a = 0; a < 3; a++) {
            if (bmin[a] > in[i].position[a]) bmin[a] = in[i].position[a];
            if (bmax[a] < in[i].position[a]) bmax[a] = in[i].position[a];
        }
    memset(out, 0, sizeof(rt_photon) * ((size_t)n + 1));
    if (n == 0) return;
    Balancer b{in, out};
    b.Segment(bmin, bmax, 1, 1, n);
}

// Which photons of an UNBALANCED array LocatePhotons can never reach once it is balanced: it descends only while
// index < halfStoredPhotons = n/2 - 1 (cyPhotonMap.h:217,371), so heap slots [2*half, n] -- the last three or four -- are
// never visited.  Found without balancing everything: the content of a segment when BalanceSegment partitions it depends
// only on the partitions of its ancestors, so the same quick-select is run along the root paths of those slots alone
// (about 2n element visits instead of n log n), on (position, index) records with the reference's own swap sequence.
namespace {
struct PathBalancer {
    struct Rec { float pos[3]; uint32_t idx; };
    Rec *ph; uint32_t first, last;          // target heap slots [first, last]
    std::vector<uint32_t> *out;
    bool Holds(uint32_t index) const
    {
        for (uint32_t t = first; t <= last; t++) { uint32_t a = t; while (a > index) a >>= 1; if (a == index) return true; }
        return false;
    }
    void Put(uint32_t index, const Rec &r) { if (index >= first && index <= last) out->push_back(r.idx); }
    void Segment(float bmin[3], float bmax[3], uint32_t index, uint32_t start, uint32_t end)
    {
        uint32_t median = 1;
        while (4 * median <= end - start + 1) median += median;
        if (3 * median <= end - start + 1) { median += median; median += start - 1; }
        else median = end - median + 1;
        int axis = 2;
        const float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
        if (dx > dy) { if (dx > dz) axis = 0; }
        else if (dy > dz) axis = 1;
        uint32_t left = start, right = end;
        while (right > left) {
            const float pivot = ph[right].pos[axis];
            uint32_t i = left - 1, j = right;
            while (ph[++i].pos[axis] < pivot) {}
            while (ph[--j].pos[axis] > pivot && j > left) {}
            while (i < j) {
                std::swap(ph[i], ph[j]);
                while (ph[++i].pos[axis] < pivot) {}
                while (ph[--j].pos[axis] > pivot && j > left) {}
            }
            std::swap(ph[i], ph[right]);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
        Put(index, ph[median]);
        const float split = ph[median].pos[axis];
        if (median > start) {
            if (start < median - 1) {
                if (Holds(2 * index)) {
                    float tmax[3] = {bmax[0], bmax[1], bmax[2]};
                    tmax[axis] = split;
                    Segment(bmin, tmax, 2 * index, start, median - 1);
                }
            } else Put(2 * index, ph[start]);
        }
        if (median < end) {
            if (median + 1 < end) {
                if (Holds(2 * index + 1)) {
                    float tmin[3] = {bmin[0], bmin[1], bmin[2]};
                    tmin[axis] = split;
                    Segment(tmin, bmax, 2 * index + 1, median + 1, end);
                }
            } else Put(2 * index + 1, ph[end]);
        }
    }
};
}  // namespace

uint32_t ReachablePhotonSlots(uint32_t n)
{
    const long long half = (long long)(n / 2) - 1;
    long long reach = 2 * half - 1;
    if (reach < 1) reach = 1;
    if (reach > (long long)n) reach = n;
    return n ? (uint32_t)reach : 0u;
}

void UnreachablePhotonRecs(PhotonPosRec *recs, uint32_t n, std::vector<uint32_t> &raw_indices)
{
    static_assert(sizeof(PhotonPosRec) == sizeof(PathBalancer::Rec) && sizeof(PhotonPosRec) == 16, "same record");
    raw_indices.clear();
    const uint32_t reach = ReachablePhotonSlots(n);
    if (n == 0 || reach >= n) return;
    float bmin[3] = {recs[0].pos[0], recs[0].pos[1], recs[0].pos[2]};             // the reference's box loop starts at the unused slot
    float bmax[3] = {bmin[0], bmin[1], bmin[2]};
    for (uint32_t i = 1; i <= n; i++)
        for (int a = 0; a < 3; a++) {
            if (bmin[a] > recs[i].pos[a]) bmin[a] = recs[i].pos[a];
            if (bmax[a] < recs[i].pos[a]) bmax[a] = recs[i].pos[a];
        }
    PathBalancer b{reinterpret_cast<PathBalancer::Rec *>(recs), reach + 1, n, &raw_indices};
    b.Segment(bmin, bmax, 1, 1, n);
    std::sort(raw_indices.begin(), raw_indices.end());
}

void UnreachablePhotons(const rt_photon *in, uint32_t n, std::vector<uint32_t> &raw_indices)
{
    raw_indices.clear();
    const uint32_t reach = ReachablePhotonSlots(n);
    if (n == 0 || reach >= n) return;
    std::vector<PathBalancer::Rec> recs((size_t)n + 1);
    float bmin[3] = {in[0].position[0], in[0].position[1], in[0].position[2]};     // the reference's box loop starts at the unused slot
    float bmax[3] = {bmin[0], bmin[1], bmin[2]};
    for (uint32_t i = 0; i <= n; i++) {
        memcpy(recs[i].pos, in[i].position, 12);
        recs[i].idx = i;
        if (i) for (int a = 0; a < 3; a++) {
            if (bmin[a] > in[i].position[a]) bmin[a] = in[i].position[a];
            if (bmax[a] < in[i].position[a]) bmax[a] = in[i].position[a];
        }
    }
    PathBalancer b{recs.data(), reach + 1, n, &raw_indices};
    b.Segment(bmin, bmax, 1, 1, n);
    std::sort(raw_indices.begin(), raw_indices.end());
}

// ---- Scene ------------------------------------------------------------------------------------------
Material *Scene::FindMaterial(const std::string &n)
{
    for (auto &m : materials) if (m && m->name == n) return m.get();
    return nullptr;
}
TriObj *Scene::FindObj(const std::string &n)
{
    for (auto &o : objList) if (o.first == n) return o.second.get();
    return nullptr;
}
Texture *Scene::FindTexture(const std::string &n)
{
    for (auto &t : textureList) if (t && t->name == n) return t.get();
    return nullptr;
}
void Scene::Clear()
{
    rootNode.Init(); materials.clear(); lights.clear(); objList.clear(); textureList.clear();
    environment = TexturedColor(); background = TexturedColor(); camera.Init();
}

// ---- lowering: Node tree -> rt_node[] in TraceNode's visiting order --------------------------------
namespace {
struct Lowerer {
    const Scene &sc; SceneData &out; std::string *err;
    std::vector<const TriObj *> meshes;
    std::vector<const Material *> mats;
    bool ok = true;
    int MeshIndex(const TriObj *t)
    {
        for (size_t i = 0; i < meshes.size(); i++) if (meshes[i] == t) return (int)i;
        meshes.push_back(t);
        return (int)meshes.size() - 1;
    }
    int MaterialIndex(const Material *m)
    {
        if (!m) return -1;
        for (size_t i = 0; i < mats.size(); i++) if (mats[i] == m) return (int)i;
        mats.push_back(m);
        return (int)mats.size() - 1;
    }
    void Visit(const Node &n, int parent)
    {
        rt_node r;
        memcpy(r.tm, n.GetTransform().data, sizeof r.tm);
        memcpy(r.itm, n.GetInverseTransform().data, sizeof r.itm);
        r.pos[0] = n.GetPosition().x; r.pos[1] = n.GetPosition().y; r.pos[2] = n.GetPosition().z;
        r.parent = parent; r.obj_type = RT_OBJ_NONE; r.mesh = -1;
        r.material = MaterialIndex(n.GetMaterial());
        if (const Object *o = n.GetNodeObj()) {
            r.obj_type = o->Kind();
            if (r.obj_type == RT_OBJ_MESH) {
                const TriObj *t = dynamic_cast<const TriObj *>(o);
                if (!t) { ok = false; if (err) *err = "node '" + n.name + "': mesh object is not a TriObj"; return; }
                r.mesh = MeshIndex(t);
            } else if (r.obj_type != RT_OBJ_SPHERE && r.obj_type != RT_OBJ_PLANE) {
                ok = false;
                if (err) *err = "node '" + n.name + "': object kind has no device lowering (no CPU path exists)";
                return;
            }
            if (r.material < 0) { ok = false; if (err) *err = "node '" + n.name + "' has an object but no material"; return; }
        }
        const int me = (int)out.nodes.size();
        out.nodes.push_back(r);
        for (int i = 0; i < n.GetNumChild() && ok; i++) Visit(*n.GetChild(i), me);
    }
};
}  // namespace

bool Lower(const Scene &scene, SceneData &out, std::string *err)
{
    out = SceneData();
    Lowerer L{scene, out, err};
    L.Visit(scene.rootNode, -1);
    if (!L.ok) return false;
    for (const TriObj *t : L.meshes) {
        MeshData m;
        m.v = t->v; m.vn = t->vn; m.f = t->f; m.fn = t->fn; m.vt = t->vt; m.ft = t->ft; m.nodes = t->nodes; m.elements = t->elements;
        out.meshes.push_back(std::move(m));
    }
    for (const Material *m : L.mats) {
        rt_blinn b;
        memset(&b, 0, sizeof b);
        if (!m->Lower(b)) { if (err) *err = "material '" + m->name + "' has no device lowering"; return false; }
        out.materials.push_back(b);
    }
    // textures in TextureList order; maps refer to them by index
    std::vector<const Texture *> texs;
    for (auto &t : scene.textureList) {
        rt_texture r;
        if (!t->Lower(r, out.texels)) { if (err) *err = "texture '" + t->name + "' has no device lowering"; return false; }
        out.textures.push_back(r);
        texs.push_back(t.get());
    }
    auto lower_map = [&](const TextureMap *m) {
        rt_texmap r;
        memset(&r, 0, sizeof r);
        r.texture = RT_MAP_NONE;
        if (!m) return r;
        r.texture = RT_MAP_EMPTY;
        for (size_t i = 0; i < texs.size(); i++) if (texs[i] == m->GetTexture()) r.texture = (int32_t)i;
        memcpy(r.tm, m->GetTransform().data, 36); memcpy(r.itm, m->GetInverseTransform().data, 36);
        r.pos[0] = m->GetPosition().x; r.pos[1] = m->GetPosition().y; r.pos[2] = m->GetPosition().z;
        return r;
    };
    bool any_map = false;
    for (const Material *m : L.mats) if (m->DiffuseMap() || m->SpecularMap()) any_map = true;
    if (any_map)
        for (const Material *m : L.mats) { out.material_maps.push_back(lower_map(m->DiffuseMap())); out.material_maps.push_back(lower_map(m->SpecularMap())); }
    out.env_map = lower_map(scene.environment.GetTexture());
    out.bg_map = lower_map(scene.background.GetTexture());
    for (auto &l : scene.lights) { rt_light r; l->Lower(r); out.lights.push_back(r); }
    const Camera &c = scene.camera;
    rt_camera &rc = out.camera;
    rc.pos[0] = c.pos.x; rc.pos[1] = c.pos.y; rc.pos[2] = c.pos.z;
    rc.dir[0] = c.dir.x; rc.dir[1] = c.dir.y; rc.dir[2] = c.dir.z;
    rc.up[0] = c.up.x; rc.up[1] = c.up.y; rc.up[2] = c.up.z;
    rc.fov = c.fov; rc.focaldist = c.focaldist; rc.dof = c.dof; rc.width = c.imgWidth; rc.height = c.imgHeight;
    out.has_camera = true;
    put3(out.env, scene.environment.GetColor());
    put3(out.bg, scene.background.GetColor());
    return true;
}

// ---- PNG (RenderImage::SavePNG, FIN/include/scene.h:645-655): 8-bit grey / RGB ------------------------
namespace {
uint32_t crc_table[256];
void crc_init() { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; } }
uint32_t crc(const uint8_t *b, size_t n, uint32_t c = 0xFFFFFFFFu) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ b[i]) & 0xFF] ^ (c >> 8); return c; }
void be32(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
void chunk(std::vector<uint8_t> &o, const char *tag, const std::vector<uint8_t> &d)
{
    be32(o, (uint32_t)d.size());
    std::vector<uint8_t> t(tag, tag + 4);
    t.insert(t.end(), d.begin(), d.end());
    o.insert(o.end(), t.begin(), t.end());
    be32(o, crc(t.data(), t.size()) ^ 0xFFFFFFFFu);
}
}  // namespace

bool WritePNG(const char *filename, const uint8_t *data, int w, int h, int comps)
{
    if (comps != 1 && comps != 3) return false;
    crc_init();
    std::vector<uint8_t> raw;
    const size_t row = (size_t)w * comps;
    raw.reserve((row + 1) * h);
    for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), data + y * row, data + (y + 1) * row); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(comps == 3 ? 2 : 0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE *fp = fopen(filename, "wb");
    if (!fp) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), fp) == out.size();
    fclose(fp);
    return ok;
}

}  // namespace rt
