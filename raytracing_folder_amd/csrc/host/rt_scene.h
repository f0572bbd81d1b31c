// rt_scene.h -- host-side mirror of the reference's scene model / plugin surface.
//
// Same class and method names, argument meaning and ownership rules as the reference
// (FIN/include/scene.h:224-536, objects.h, lights.h, materials.h; FIN = /root/reference/
// RayTracingFinal/RayTracingFinal), so that host code written against the reference's
// Node / Object / Light / Material classes keeps working.  What changes is what happens
// below: instead of virtual IntersectRay/Shade calls per ray, a scene is LOWERED to the POD
// records of include/rt_mi355x.h (rt::Lower) and rendered by the HIP kernels.  Built-in types
// lower to device records; a subclass the lowering does not know makes Lower() fail loudly
// (there is no CPU path).
#ifndef RT_HOST_SCENE_H
#define RT_HOST_SCENE_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt_mi355x.h"

namespace rt {

// ---- float3 / 3x3 algebra with the reference's evaluation order -------------------------
// (cyPoint.h:259-350, cyMatrix.h:279-640: column-major, sums left to right)
struct Point3 {
    float x, y, z;
    Point3() : x(0), y(0), z(0) {}
    Point3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
    Point3 operator+(const Point3 &p) const { return Point3(x + p.x, y + p.y, z + p.z); }
    Point3 operator-(const Point3 &p) const { return Point3(x - p.x, y - p.y, z - p.z); }
    Point3 operator*(float v) const { return Point3(x * v, y * v, z * v); }
    Point3 operator/(float v) const { return Point3(x / v, y / v, z / v); }
    Point3 operator-() const { return Point3(-x, -y, -z); }
    Point3 &operator+=(const Point3 &p) { x += p.x; y += p.y; z += p.z; return *this; }
    float Dot(const Point3 &p) const { return x * p.x + y * p.y + z * p.z; }
    Point3 Cross(const Point3 &p) const { return Point3(y * p.z - z * p.y, z * p.x - x * p.z, x * p.y - y * p.x); }
    Point3 operator^(const Point3 &p) const { return Cross(p); }
    float LengthSquared() const { return Dot(*this); }
    float Length() const { return sqrtf(LengthSquared()); }
    void Normalize() { *this = *this / Length(); }
    Point3 GetNormalized() const { return *this / Length(); }
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

struct Color {
    float r, g, b;
    Color() : r(0), g(0), b(0) {}
    Color(float R, float G, float B) : r(R), g(G), b(B) {}
    explicit Color(float v) : r(v), g(v), b(v) {}
    Color operator*(float v) const { return Color(r * v, g * v, b * v); }
    Color &operator*=(float v) { r *= v; g *= v; b *= v; return *this; }
    float Gray() const { return (r + g + b) / 3.0f; }
};

struct Matrix3 {
    float data[9];   // | 0 3 6 | 1 4 7 | 2 5 8 |
    void SetIdentity() { Zero(); data[0] = data[4] = data[8] = 1; }
    void Zero() { for (float &d : data) d = 0; }
    Matrix3 operator*(const Matrix3 &right) const {
        Matrix3 r;
        for (int i = 0; i < 9; i += 3)
            for (int j = 0; j < 3; ++j)
                r.data[i + j] = data[j] * right.data[i] + data[3 + j] * right.data[i + 1] + data[6 + j] * right.data[i + 2];
        return r;
    }
    Point3 operator*(const Point3 &p) const {
        return Point3(p.x * data[0] + p.y * data[3] + p.z * data[6],
                      p.x * data[1] + p.y * data[4] + p.z * data[7],
                      p.x * data[2] + p.y * data[5] + p.z * data[8]);
    }
    void SetRotation(const Point3 &axis, float angle);
    void GetInverse(Matrix3 &inverse) const;
};

// ---- Transformation (FIN/include/scene.h:224-262) ----------------------------------------
class Transformation {
    Matrix3 tm;
    Point3 pos;
    Matrix3 itm;
public:
    Transformation() { tm.SetIdentity(); itm.SetIdentity(); }
    const Matrix3 &GetTransform() const { return tm; }
    const Point3 &GetPosition() const { return pos; }
    const Matrix3 &GetInverseTransform() const { return itm; }
    void Translate(Point3 p) { pos += p; }
    void Rotate(Point3 axis, float degree) { Matrix3 m; m.SetRotation(axis, degree * (float)M_PI / 180.0f); Transform(m); }
    void Scale(float sx, float sy, float sz) { Matrix3 m; m.Zero(); m.data[0] = sx; m.data[4] = sy; m.data[8] = sz; Transform(m); }
    void Transform(const Matrix3 &m) { tm = m * tm; pos = m * pos; tm.GetInverse(itm); }
    void InitTransform() { pos = Point3(); tm.SetIdentity(); itm.SetIdentity(); }
};

// ---- textures (FIN/include/scene.h:323-434, FIN/include/texture.h) ---------------------------------
class Texture {
public:
    std::string name;
    virtual ~Texture() {}
    virtual bool Lower(rt_texture &out, std::vector<uint8_t> &texels) const = 0;
};
class TextureFile : public Texture {          // PNG / PPM, bilinear + tiling on the device
public:
    int width = 0, height = 0;
    std::vector<uint8_t> data;                // RGB8
    bool Load(std::string *err = nullptr);    // reads `name`
    bool Lower(rt_texture &out, std::vector<uint8_t> &texels) const override;
};
class TextureChecker : public Texture {
    Color color1{0, 0, 0}, color2{1, 1, 1};
public:
    void SetColor1(const Color &c) { color1 = c; }
    void SetColor2(const Color &c) { color2 = c; }
    bool Lower(rt_texture &out, std::vector<uint8_t> &texels) const override;
};
// a texture reference plus a transformation of the uvw coordinate
class TextureMap : public Transformation {
    Texture *texture;
public:
    explicit TextureMap(Texture *tex = nullptr) : texture(tex) {}
    void SetTexture(Texture *tex) { texture = tex; }
    const Texture *GetTexture() const { return texture; }
};
// a colour that may carry a TextureMap; the sampled value is colour * texture
class TexturedColor {
    Color color;
    std::unique_ptr<TextureMap> map;
public:
    TexturedColor() {}
    TexturedColor(float r, float g, float b) : color(r, g, b) {}
    void SetColor(const Color &c) { color = c; }
    void SetTexture(TextureMap *m) { map.reset(m); }
    Color GetColor() const { return color; }
    const TextureMap *GetTexture() const { return map.get(); }
};

// ---- Object plug-ins (FIN/include/scene.h:269-275, objects.h) ----------------------------
class Material;
class Object {
public:
    virtual ~Object() {}
    // what the lowering needs to know instead of a per-ray IntersectRay callback
    virtual int Kind() const = 0;   // RT_OBJ_*
};
class Sphere : public Object { public: int Kind() const override { return RT_OBJ_SPHERE; } };
class Plane : public Object { public: int Kind() const override { return RT_OBJ_PLANE; } };

// cyTriMesh arrays + cyBVH arrays of one OBJ (FIN/include/objects.h:124-145)
// cyTriMesh::Mtl (FIN/include/cyTriMesh.h:74-102): the fields of a .mtl entry this path consumes
struct ObjMtl {
    std::string name;
    float Kd[3] = {1, 1, 1}, Ks[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};
    float Ns = 0, Ni = 1;
    int illum = 2;
    std::string map_Kd, map_Ks;          // empty = not given
};

class TriObj : public Object {
public:
    std::vector<float> v, vn, vt;        // xyz triples (vt: uvw)
    std::vector<uint32_t> f, fn, ft;     // 3 indices per face
    std::vector<ObjMtl> mtls;            // in order of first `usemtl` (loadMtl only)
    std::vector<uint32_t> mcfc;          // material cumulative face count (cyTriMesh.h:113)
    std::vector<rt_bvh_node> nodes;      // reference layout, node 0 unused
    std::vector<uint32_t> elements;
    int Kind() const override { return RT_OBJ_MESH; }
    unsigned NV() const { return (unsigned)(v.size() / 3); }
    unsigned NF() const { return (unsigned)(f.size() / 3); }
    unsigned NVN() const { return (unsigned)(vn.size() / 3); }
    unsigned NVT() const { return (unsigned)(vt.size() / 3); }
    unsigned NM() const { return (unsigned)mtls.size(); }
    bool HasNormals() const { return NVN() > 0; }
    bool HasTextureVertices() const { return NVT() > 0; }
    // TriObj::Load: LoadFromFileObj + ComputeNormals if needed + bvh.SetMesh(this,4)
    bool Load(const char *filename, bool loadMtl, std::string *err = nullptr);
    bool LoadFromFileObj(const char *filename, bool loadMtl, std::string *err = nullptr);
    void ComputeNormals();
    void BuildBVH(unsigned maxElementsPerNode = 4);
};

// ---- lights (FIN/include/lights.h) --------------------------------------------------------
class Light {
public:
    std::string name;
    virtual ~Light() {}
    virtual bool IsAmbient() const { return false; }
    virtual void Lower(rt_light &out) const = 0;
};
class AmbientLight : public Light {
    Color intensity;
public:
    bool IsAmbient() const override { return true; }
    void SetIntensity(Color c) { intensity = c; }
    void Lower(rt_light &o) const override;
};
class DirectLight : public Light {
    Color intensity; Point3 direction{0, 0, 1};
public:
    void SetIntensity(Color c) { intensity = c; }
    void SetDirection(Point3 d) { direction = d.GetNormalized(); }
    void Lower(rt_light &o) const override;
};
class PointLight : public Light {
    Color intensity; Point3 position; float size = 0;
public:
    void SetIntensity(Color c) { intensity = c; }
    void SetPosition(Point3 p) { position = p; }
    void SetSize(float s) { size = s; }
    void Lower(rt_light &o) const override;
};

// ---- materials (FIN/include/materials.h:68-98) -------------------------------------------
class Material {
public:
    std::string name;
    virtual ~Material() {}
    virtual bool Lower(rt_blinn &out) const = 0;
    virtual const TextureMap *DiffuseMap() const { return nullptr; }
    virtual const TextureMap *SpecularMap() const { return nullptr; }
};
class MtlBlinn : public Material {
    TexturedColor diffuse{0.5f, 0.5f, 0.5f}, specular{0.7f, 0.7f, 0.7f};
    Color reflection, refraction, emission, absorption;
    float glossiness = 20.0f, ior = 1, reflectionGlossiness = 0, refractionGlossiness = 0;
public:
    void SetDiffuse(Color c) { diffuse.SetColor(c); }
    void SetSpecular(Color c) { specular.SetColor(c); }
    void SetDiffuseTexture(TextureMap *m) { diffuse.SetTexture(m); }
    void SetSpecularTexture(TextureMap *m) { specular.SetTexture(m); }
    const TextureMap *DiffuseMap() const override { return diffuse.GetTexture(); }
    const TextureMap *SpecularMap() const override { return specular.GetTexture(); }
    void SetGlossiness(float g) { glossiness = g; }
    void SetEmission(Color c) { emission = c; }
    void SetReflection(Color c) { reflection = c; }
    void SetRefraction(Color c) { refraction = c; }
    void SetAbsorption(Color c) { absorption = c; }
    void SetRefractionIndex(float i) { ior = i; }
    void SetReflectionGlossiness(float g) { reflectionGlossiness = g; }
    void SetRefractionGlossiness(float g) { refractionGlossiness = g; }
    bool IsPhotonSurface() const { return diffuse.GetColor().Gray() > 0; }
    bool Lower(rt_blinn &o) const override;
};

// MultiMtl (FIN/include/materials.h:385-404): Shade/RandomPhotonBounce/IsPhotonSurface pick
// mtls[hInfo.mtlID], and nothing on the render path ever sets HitInfo::mtlID after Init() zeroes it
// (scene.h:163) -- so a MultiMtl renders as its FIRST sub-material on every face, which is what the
// lowering hands the device.  (With no sub-material Shade returns white; the loader never makes one.)
class MultiMtl : public Material {
    std::vector<std::unique_ptr<MtlBlinn>> mtls;
public:
    void AppendMaterial(MtlBlinn *m) { mtls.emplace_back(m); }
    int NumMaterials() const { return (int)mtls.size(); }
    const MtlBlinn *Sub(int i) const { return mtls[i].get(); }
    bool Lower(rt_blinn &o) const override;
    const TextureMap *DiffuseMap() const override { return mtls.empty() ? nullptr : mtls[0]->DiffuseMap(); }
    const TextureMap *SpecularMap() const override { return mtls.empty() ? nullptr : mtls[0]->SpecularMap(); }
};

// ---- Node (FIN/include/scene.h:438-514): owns its children, not its Object/Material -------
class Node : public Transformation {
    std::vector<Node *> child;
    Object *obj = nullptr;
    Material *mtl = nullptr;
public:
    std::string name, mtlName;
    ~Node() { DeleteAllChildNodes(); }
    int GetNumChild() const { return (int)child.size(); }
    const Node *GetChild(int i) const { return child[i]; }
    Node *GetChild(int i) { return child[i]; }
    void AppendChild(Node *n) { child.push_back(n); }
    void DeleteAllChildNodes() { for (Node *c : child) delete c; child.clear(); }
    const Object *GetNodeObj() const { return obj; }
    void SetNodeObj(Object *o) { obj = o; }
    const Material *GetMaterial() const { return mtl; }
    void SetMaterial(Material *m) { mtl = m; }
    void Init() { DeleteAllChildNodes(); obj = nullptr; mtl = nullptr; name.clear(); InitTransform(); }
};

// ---- Camera (FIN/include/scene.h:518-536) --------------------------------------------------
struct Camera {
    Point3 pos, dir{0, 0, -1}, up{0, 1, 0};
    float fov = 40, focaldist = 1, dof = 0;
    int imgWidth = 200, imgHeight = 150;
    void Init() { *this = Camera(); }
};

// ---- the reference's global singletons, gathered in one object -----------------------------
// (rootNode, camera, materials, lights, objList, environment, background: FIN/main.cpp:36-49)
struct Scene {
    Node rootNode;
    Camera camera;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Light>> lights;
    std::vector<std::pair<std::string, std::unique_ptr<TriObj>>> objList;   // ObjFileList
    Sphere theSphere;
    Plane thePlane;
    TexturedColor environment, background;
    std::vector<std::unique_ptr<Texture>> textureList;                       // TextureList
    Texture *FindTexture(const std::string &n);
    Material *FindMaterial(const std::string &n);
    TriObj *FindObj(const std::string &n);
    void Clear();
};

// LoadScene (FIN/xmlload.cpp:65-132): returns 1 on success, 0 on failure (err gets the reason)
int LoadScene(Scene &scene, const char *filename, std::string *err = nullptr);

// ---- lowered form: exactly what the C ABI setters take -------------------------------------
struct MeshData {
    std::vector<float> v, vn, vt;        // vt/ft empty: the mesh has no texture vertices
    std::vector<uint32_t> f, fn, ft;
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> elements;
};
struct SceneData {
    std::vector<rt_node> nodes;
    std::vector<MeshData> meshes;
    std::vector<rt_blinn> materials;
    std::vector<rt_light> lights;
    std::vector<rt_photon> photons;    // balanced, [0] unused; empty = no photon map
    std::vector<rt_photon> caustic_photons;   // the second map (P13's causticmap), same format
    rt_camera camera{};
    float env[3] = {0, 0, 0}, bg[3] = {0, 0, 0};
    bool has_camera = false;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<rt_texmap> material_maps;      // 2 per material (diffuse, specular); empty = none anywhere
    rt_texmap env_map, bg_map;                 // texture == RT_MAP_NONE when absent
    SceneData() { memset(&env_map, 0, sizeof env_map); memset(&bg_map, 0, sizeof bg_map); env_map.texture = bg_map.texture = RT_MAP_NONE; }
};
bool Lower(const Scene &scene, SceneData &out, std::string *err);

// cyBVH::Build with MeanSplit (FIN/include/cyBVH.h:122-142,242-328)
void BuildMeanSplitBVH(const float *v, const uint32_t *f, unsigned nf, unsigned maxPerLeaf,
                       std::vector<rt_bvh_node> &nodes, std::vector<uint32_t> &elements);
// PhotonMap::PrepareForIrradianceEstimation (FIN/include/cyPhotonMap.h:196-284)
void BalancePhotons(rt_photon *in, uint32_t n, rt_photon *out);
// heap slots 1..ReachablePhotonSlots(n) are the ones LocatePhotons can visit (cyPhotonMap.h:217,371)
uint32_t ReachablePhotonSlots(uint32_t n);
// indices (1-based, ascending) into an UNBALANCED array of the photons that balancing would put beyond them
void UnreachablePhotons(const rt_photon *in, uint32_t n, std::vector<uint32_t> &raw_indices);
// the same on (position, raw index) records [0, n] ([0] = the unused all-zero slot), which the replay permutes in place
struct PhotonPosRec { float pos[3]; uint32_t idx; };
void UnreachablePhotonRecs(PhotonPosRec *recs, uint32_t n, std::vector<uint32_t> &raw_indices);

// RenderImage::ComputeZBufferImage / ComputeSampleCountImage (FIN/include/scene.h:591-637) on plain arrays
void ZBufferImage(const float *zbuffer, size_t size, uint8_t *zbufferImg);
int SampleCountImage(const uint8_t *sampleCount, size_t size, uint8_t *sampleCountImg);
// PNG (non-interlaced, any legal bit depth) and binary PPM reader -> RGB8, as TextureFile::Load needs
bool ReadImageRGB(const char *filename, int &w, int &h, std::vector<uint8_t> &rgb, std::string *err);
// minimal PNG writer (8-bit grey or RGB, stored deflate blocks) for RenderImage::SavePNG
bool WritePNG(const char *filename, const uint8_t *data, int width, int height, int comps);

}  // namespace rt
#endif
