// rt_binding.cpp -- the REFERENCE-SIDE binding of librt_mi355x: the file a maintainer of
// Roia2529/RayTracing-folder drops next to RayTracingFinal/main.cpp to replace its BeginRender / StopRender /
// saveImage (FIN/main.cpp:984-1012) by calls into the C ABI of include/rt_mi355x.h, keeping viewport.cpp,
// xmlload.cpp and the global singletons untouched.  Compiled against the reference's OWN headers: syntax-checked
// for both snapshots (`make -C oracle binding-check`) and LINKED AND RUN with the reference's main.cpp, its
// LoadScene and its saveImage by oracle/ref_binding_harness.cpp (`make -C oracle refbinding`; the GPU tests run
// it on the box: reference scene graph -> this file -> HIP kernels -> the reference's RenderImage -> its PNG writer).
//
// The reference keeps the fields of MtlBlinn, the lights, TextureFile/TextureChecker, TextureMap and
// TexturedColor private and offers no getters for most of them.  A maintainer would add accessors (or a
// `friend` line); this file, living outside the reference's tree, reads them by opening the classes up for its
// own translation unit only (the two defines below change no layout).
// the standard headers come FIRST: scene.h defines min/max as macros (scene.h:48-54), which breaks every
// libstdc++ header included after it (main.cpp gets away with it through its include order)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <typeinfo>
#include <vector>

#define private public
#define protected public
#include "scene.h"
#include "objects.h"
#include "lights.h"
#include "materials.h"
#include "texture.h"
#include "cyPhotonMap.h"
#undef private
#undef protected

#include "rt_mi355x.h"

extern Node rootNode; extern Camera camera; extern RenderImage renderImage;
extern MaterialList materials; extern LightList lights; extern TextureList textureList;
extern TexturedColor environment, background; extern cy::PhotonMap photonmap;
extern Sphere theSphere; extern Plane thePlane;

static rt_scene *g_scene; static std::vector<rt_job *> g_jobs;

// walk the Node tree in TraceNode's order (FIN/main.cpp:108-130): parent before children
static void LowerNodes(const Node &n, int parent, std::vector<rt_node> &out, std::vector<const TriObj *> &meshes)
{
    rt_node r;
    memset(&r, 0, sizeof r);
    memcpy(r.tm, n.GetTransform().data, 36);
    memcpy(r.itm, n.GetInverseTransform().data, 36);
    memcpy(r.pos, &n.GetPosition().x, 12);
    r.parent = parent; r.mesh = -1; r.obj_type = RT_OBJ_NONE; r.material = -1;
    for (size_t i = 0; i < materials.size(); i++) if (materials[i] == n.GetMaterial()) r.material = (int32_t)i;
    const Object *o = n.GetNodeObj();
    if (o == &theSphere) r.obj_type = RT_OBJ_SPHERE;
    else if (o == &thePlane) r.obj_type = RT_OBJ_PLANE;
    else if (const TriObj *t = dynamic_cast<const TriObj *>(o)) {
        r.obj_type = RT_OBJ_MESH;
        r.mesh = (int32_t)(std::find(meshes.begin(), meshes.end(), t) - meshes.begin());
        if (r.mesh == (int32_t)meshes.size()) meshes.push_back(t);
    }
    const int me = (int)out.size();
    out.push_back(r);
    for (int i = 0; i < n.GetNumChild(); i++) LowerNodes(*n.GetChild(i), me, out, meshes);
}

static void put(float dst[3], const Color &c) { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; }
static void put(float dst[3], const Point3 &p) { dst[0] = p.x; dst[1] = p.y; dst[2] = p.z; }

// MtlBlinn (materials.h:68-100) -> the 88-byte rt_blinn block.  A MultiMtl lowers to its FIRST sub-material:
// HitInfo::mtlID is never set on the render path (scene.h:163, materials.h:393).
static std::vector<rt_blinn> LowerMaterials(const MaterialList &list, std::vector<const MtlBlinn *> &blinns)
{
    std::vector<rt_blinn> out;
    for (size_t i = 0; i < list.size(); i++) {
        const Material *m = list[i];
        if (const MultiMtl *mm = dynamic_cast<const MultiMtl *>(m)) m = mm->mtls.empty() ? nullptr : mm->mtls[0];
        const MtlBlinn *b = dynamic_cast<const MtlBlinn *>(m);
        rt_blinn r;
        memset(&r, 0, sizeof r);
        if (b) {
            put(r.diffuse, b->diffuse.GetColor()); put(r.specular, b->specular.GetColor());
            put(r.reflection, b->reflection.GetColor()); put(r.refraction, b->refraction.GetColor());
            put(r.emission, b->emission.GetColor()); put(r.absorption, b->absorption);
            r.glossiness = b->glossiness; r.ior = b->ior;
            r.reflection_glossiness = b->reflectionGlossiness; r.refraction_glossiness = b->refractionGlossiness;
        }
        blinns.push_back(b);
        out.push_back(r);
    }
    return out;
}

// AmbientLight / DirectLight / PointLight (lights.h:28-175) -> rt_light
static std::vector<rt_light> LowerLights(const LightList &list)
{
    std::vector<rt_light> out;
    for (size_t i = 0; i < list.size(); i++) {
        rt_light r;
        memset(&r, 0, sizeof r);
        if (const AmbientLight *a = dynamic_cast<const AmbientLight *>(list[i])) { r.type = RT_LIGHT_AMBIENT; put(r.intensity, a->intensity); }
        else if (const DirectLight *d = dynamic_cast<const DirectLight *>(list[i])) { r.type = RT_LIGHT_DIRECT; put(r.intensity, d->intensity); put(r.direction, d->direction); }
        else if (const PointLight *p = dynamic_cast<const PointLight *>(list[i])) { r.type = RT_LIGHT_POINT; put(r.intensity, p->intensity); put(r.position, p->position); r.size = p->size; }
        out.push_back(r);
    }
    return out;
}

// ItemFileList<Texture> (scene.h:198-220) keeps its entries in a private ItemList<FileInfo>
static size_t TextureCount(const TextureList &l) { return l.list.size(); }
static const Texture *TextureAt(const TextureList &l, size_t i) { return l.list[i] ? l.list[i]->item : nullptr; }

// textureList (TextureFile: RGB8 texels into one shared byte array; TextureChecker: two colours) -> rt_texture[]
static void LowerTextures(const TextureList &list, std::vector<rt_texture> &tex, std::vector<uint8_t> &texels)
{
    for (size_t i = 0; i < TextureCount(list); i++) {
        rt_texture t;
        memset(&t, 0, sizeof t);
        if (const TextureFile *f = dynamic_cast<const TextureFile *>(TextureAt(list, i))) {
            t.type = RT_TEX_FILE; t.width = f->width; t.height = f->height; t.texel_offset = (uint32_t)texels.size();
            for (size_t k = 0; k < f->data.size(); k++) { texels.push_back(f->data[k].r); texels.push_back(f->data[k].g); texels.push_back(f->data[k].b); }
        } else if (const TextureChecker *c = dynamic_cast<const TextureChecker *>(TextureAt(list, i))) {
            t.type = RT_TEX_CHECKER; put(t.color1, c->color1); put(t.color2, c->color2);
        }
        tex.push_back(t);
    }
}

// one TexturedColor's TextureMap (scene.h:376-398) -> rt_texmap {texture index, tm, itm, pos}
static rt_texmap LowerMap(const TexturedColor &tc, const TextureList &list)
{
    rt_texmap m;
    memset(&m, 0, sizeof m);
    m.texture = RT_MAP_NONE;
    m.tm[0] = m.tm[4] = m.tm[8] = m.itm[0] = m.itm[4] = m.itm[8] = 1.0f;
    const TextureMap *map = tc.GetTexture();
    if (!map) return m;
    m.texture = RT_MAP_EMPTY;                        // a map without a texture samples black (scene.h:383)
    for (size_t i = 0; i < TextureCount(list); i++) if (TextureAt(list, i) == map->texture) m.texture = (int32_t)i;
    memcpy(m.tm, map->GetTransform().data, 36);
    memcpy(m.itm, map->GetInverseTransform().data, 36);
    memcpy(m.pos, &map->GetPosition().x, 12);
    return m;
}

void BeginRender()                                   // must return immediately (viewport.cpp:36)
{
    if (!g_scene) rt_scene_create(&g_scene);
    std::vector<rt_node> nodes; std::vector<const TriObj *> meshes;
    LowerNodes(rootNode, -1, nodes, meshes);
    rt_scene_set_nodes(g_scene, nodes.data(), (int32_t)nodes.size());
    for (size_t m = 0; m < meshes.size(); m++) {
        const TriObj &t = *meshes[m];                // V/F/VN/FN are contiguous float / unsigned triples
        // cyBVH keeps its arrays private: the identical tree comes out of the public helper
        std::vector<rt_bvh_node> bn(2 * t.NF() + 2); std::vector<uint32_t> el(t.NF()); int32_t nn = 0;
        rt_bvh_build(&t.V(0).x, (int32_t)t.NV(), t.F(0).v, (int32_t)t.NF(), 4, bn.data(), &nn, el.data());
        rt_scene_set_mesh(g_scene, (int32_t)m, &t.V(0).x, (int32_t)t.NV(), t.F(0).v, (int32_t)t.NF(),
                          &t.VN(0).x, (int32_t)t.NVN(), t.FN(0).v, bn.data(), nn, el.data());
        if (t.HasTextureVertices())                  // read by the PROJ13-family triangle only (objects.h:203)
            rt_scene_set_mesh_texcoords(g_scene, (int32_t)m, &t.VT(0).x, (int32_t)t.NVT(), t.FT(0).v);
    }
    std::vector<const MtlBlinn *> blinns;
    std::vector<rt_blinn> mats = LowerMaterials(materials, blinns);
    std::vector<rt_light> ls = LowerLights(lights);
    rt_scene_set_materials(g_scene, mats.data(), (int32_t)mats.size());
    rt_scene_set_lights(g_scene, ls.data(), (int32_t)ls.size());
    const Color e = environment.GetColor(), b = background.GetColor();
    rt_scene_set_environment(g_scene, &e.r, &b.r);
    std::vector<rt_texture> tex; std::vector<uint8_t> texels;
    LowerTextures(textureList, tex, texels);
    rt_scene_set_textures(g_scene, tex.data(), (int32_t)tex.size(), texels.data(), (uint64_t)texels.size());
    std::vector<rt_texmap> maps;                     // [2i] diffuse, [2i+1] specular (the two Shade samples, FIN/main.cpp:531-532)
    for (size_t i = 0; i < blinns.size(); i++) {
        const TexturedColor none;
        maps.push_back(LowerMap(blinns[i] ? blinns[i]->diffuse : none, textureList));
        maps.push_back(LowerMap(blinns[i] ? blinns[i]->specular : none, textureList));
    }
    rt_scene_set_material_maps(g_scene, maps.data(), (int32_t)blinns.size());
    const rt_texmap env_map = LowerMap(environment, textureList), bg_map = LowerMap(background, textureList);
    rt_scene_set_environment_maps(g_scene, &env_map, &bg_map);

    // generatePhotonMap() (FIN/main.cpp:350-402) is what the replaced BeginRender called first (:989).  Either the host
    // program still runs the reference's own pass before this function -- then its balanced vector IS the wire format
    // (sizeof(cy::PhotonMap::Photon) == sizeof(rt_photon) == 24, photons[0] unused on both sides) and is handed over --
    // or the map is empty and rt_render_begin runs the photon pass itself, on the GPU, on the job's thread
    // (rt_params.photon_count / photon_bounce = MAX_NUM_OF_PHOTON / PHOTON_BOUNCE; rt_scene_set_photon_dump names the
    // .dat that :397-400 wrote to a hard-coded path).
    static_assert(sizeof(cy::PhotonMap::Photon) == sizeof(rt_photon), "photon record layout");
    if (photonmap.photons.size() > 1) rt_scene_set_photons(g_scene, (const rt_photon *)&photonmap.photons[0], (uint32_t)photonmap.photons.size() - 1);
    else rt_scene_set_photons(g_scene, nullptr, 0);
    // RayTracingProj13's second map: rt_caustic_pass + rt_photon_balance + rt_scene_set_caustic_photons, and
    // p.shade_model = RT_SHADE_P13, p.caustic_k = 400, p.caustic_radius = 0.5 (prj13.html).

    rt_camera cam; rt_params p; rt_params_default(&p);    // the FIN #defines (main.cpp:19-32) as run-time values
    // p.shade_model picks the snapshot whose Shade / primitive semantics are wanted: RT_SHADE_FIN (default),
    // RT_SHADE_P13, RT_SHADE_P12 (live GI), RT_SHADE_P6, RT_SHADE_P3; p.seed keys the counter RNG that stands
    // in for rand() (photon pass, soft shadows, glossy, DoF, GI)
    memcpy(cam.pos, &camera.pos.x, 12); memcpy(cam.dir, &camera.dir.x, 12); memcpy(cam.up, &camera.up.x, 12);
    cam.fov = camera.fov; cam.focaldist = camera.focaldist; cam.dof = camera.dof;
    cam.width = camera.imgWidth; cam.height = camera.imgHeight;
    // one job per gfx950 device of the node, interleaved 32 x 8 tiles (the reference: 2 x hardware_concurrency threads on
    // one shared pixel counter, :987-997), all into the reference's own RenderImage buffers; the first job to start runs
    // the photon pass, the others wait for it
    const int N = rt_device_count() > 0 ? rt_device_count() : 1;
    for (int r = 0; r < N; r++) {
        const rt_tile_range mine = {32, 8, r, N};
        rt_job *job = nullptr;
        if (rt_render_begin(g_scene, &cam, &p, &mine, /*device*/ r, &renderImage.GetPixels()->r, renderImage.GetZBuffer(),
                            renderImage.GetSampleCount(), &job) != RT_OK) {
            fprintf(stderr, "BeginRender: %s\n", rt_last_error());    // no GPU => loud failure, there is no CPU path
            break;
        }
        g_jobs.push_back(job);
    }
}

void StopRender() { for (rt_job *j : g_jobs) rt_render_stop(j); }

// viewport.cpp polls renderImage.IsRenderDone() (viewport.cpp:390-409); its idle callback feeds the counter:
//     renderImage.ResetNumRenderedPixels(); renderImage.IncrementNumRenderPixel(RenderProgress());
bool RenderRunning() { return !g_jobs.empty(); }   // false after a BeginRender that could not start (no GPU)
int RenderProgress() { int n = 0; for (rt_job *j : g_jobs) n += rt_render_progress(j); return n; }

void saveImage()                                     // the body of FIN/main.cpp:1000-1007 behind a wait
{
    for (rt_job *j : g_jobs) { if (rt_render_wait(j) != RT_OK) fprintf(stderr, "render: %s\n", rt_last_error()); rt_job_destroy(j); }
    g_jobs.clear();
    renderImage.ComputeZBufferImage();
    renderImage.SaveImage("prj13box.png");
    renderImage.ComputeSampleCountImage();
    renderImage.SaveSampleCountImage("prj13box_sc.png");
}
