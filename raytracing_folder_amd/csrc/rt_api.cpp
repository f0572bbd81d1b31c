// rt_api.cpp -- implementation of the C ABI in include/rt_mi355x.h: scene store, lowering to the
// device layout of rt_dev.h, per-chunk wavefront orchestration on a HIP stream, async jobs.
// There is no CPU render path in this library: without a gfx950 device the render calls fail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "host/rt_scene.h"
#include "rt_dev.h"

// launch wrappers defined in rt_kernels.hip
void rtk_launch_primary(hipStream_t, const DevScene &, const DevWork &, const rt_params &, const DevRayQueue &, uint32_t *,
                        const DevCamera &, const DevTiles &, uint32_t, uint32_t, int, int, int, int, const float *, int);
void rtk_launch_bounce(hipStream_t, const DevScene &, const DevWork &, const rt_params &, const DevRayQueue &,
                       const DevRayQueue &, uint32_t *, int, int);
void rtk_launch_trace(hipStream_t, const DevScene &, int, const float *, long long, uint8_t *, float *, float *, float *, int32_t *, uint8_t *);
bool rtk_launch_wavefront_queue(hipStream_t, const DevScene &, const DevWork &, const rt_params &, const DevRayQueue &, const uint32_t *,
                                const DevRayQueue &, uint32_t *, const DevCamera &, const DevTiles &, uint32_t, int, int);
void rtk_launch_gather(hipStream_t, const DevPhotonMap &, const float4 *, const float4 *, const float4 *, const uint32_t *,
                       uint32_t, int, float, float *, float *, float *, int, unsigned long long *, int, uint32_t *, float *);
bool rtk_wavefront_usable(const DevScene &, const rt_params &);
void rtk_launch_resolve(hipStream_t, const DevScene &, const DevWork &, const DevCamera &, const DevTiles &, uint32_t, uint32_t, int, int,
                        float, float, int, const float *, uint8_t *, float *, uint8_t *, void *, int);
void rtk_launch_unpack_tiles(hipStream_t, const void *, int, int, int, int, int, int, uint8_t *, float *, uint8_t *);

// k_gather is a persistent grid that pulls query batches from a counter: enough workgroups to fill
// every CU at the kernel's occupancy (256 CUs x 5 resident workgroups of 4 waves)
static int gather_blocks()
{
    static int n = 0;
    if (n == 0) {
        const char *e = getenv("RT_GATHER_BLOCKS");     // tuning experiments only
        const long v = e ? atol(e) : 0;
        n = (v >= 64 && v <= 16384) ? (int)v : 256 * 5;
    }
    return n;
}
static const size_t STATS_BYTES = (size_t)ST_COUNT * RT_STAT_STRIDE * 8;    // the statistics block: counters RT_STAT_STRIDE apart (rt_dev.h)
#define GATHER_BLOCKS gather_blocks()

void rtk_launch_photon_trace(hipStream_t, const DevScene &, unsigned long long, uint32_t, uint32_t, int, float *, uint32_t *, int);
// rt_photon_build.hip: the photon set-up on the GPU
struct PhotonGridOut { float min[3]; float cell; int dim[3]; };
size_t rtk_photon_compact_scratch(uint32_t n_attempts);
void rtk_photon_compact(hipStream_t, const float *recs, const uint32_t *count, uint32_t n_attempts, int mode, unsigned long long max_count,
                        void *state_dev, rt_photon *out, uint32_t out_cap, void *scratch, size_t scratch_bytes);
void rtk_photon_scale(hipStream_t, rt_photon *ph, uint32_t n, float scale);
void rtk_photon_copy_skipping(hipStream_t, const rt_photon *in, uint32_t n_in, const uint32_t *skip, uint32_t n_skip, rt_photon *out);
size_t rtk_photon_structure_scratch(uint32_t n, uint32_t n_sub);
hipError_t rtk_photon_structure(hipStream_t, const rt_photon *ph, uint32_t n, uint32_t n_sub, float4 *pa, float4 *pb, float4 *box4,
                                uint32_t *grid, PhotonGridOut *grid_out, void *scratch, size_t scratch_bytes);
size_t rtk_photon_unreachable_scratch(uint32_t n);
hipError_t rtk_photon_unreachable(hipStream_t, const rt_photon *ph0, uint32_t n, uint32_t first, uint32_t last, void *scratch, size_t scratch_bytes, uint32_t result[16]);
void rtk_photon_pack_positions(hipStream_t, const rt_photon *ph0, uint32_t n, void *recs16);
void rtk_photon_cell_start(hipStream_t, const float4 *tbox, uint32_t n_leaves, const float grid_min[3], float cell, const int dim[3], float radius, uint32_t *start);

// ---- errors ---------------------------------------------------------------------------------------
static thread_local std::string g_err;
static rt_status fail(rt_status st, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *rt_last_error(void) { return g_err.c_str(); }
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION; }

static bool device_is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; i++) if (device_is_gfx950(i)) ok++;
    return ok;
}

extern "C" void rt_params_default(rt_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->min_sample = 4; p->max_sample = 8; p->threshold = 1e-3f; p->bounce = 4;      // FIN/main.cpp:19-26
    p->hemisphere_sample = 30; p->knn_k = 400; p->knn_radius = 1.0f;                // :26, :699
    p->shade_model = RT_SHADE_FIN; p->shadow_samples = 4; p->seed = 20171203u; p->gamma = 2.2;
    p->photon_count = 1000000; p->photon_bounce = 8;                                // MAX_NUM_OF_PHOTON, PHOTON_BOUNCE (:27, :29)
}

// ---- device buffers --------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    rt_status ensure(size_t n)
    {
        if (n <= bytes && p) return RT_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (n == 0) n = 16;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return RT_OK;
    }
    rt_status upload(const void *src, size_t n)
    {
        rt_status st = ensure(n);
        if (st) return st;
        if (n) HIP_TRY(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
        return RT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct DevMeshBufs { DevBuf nodes, tris, nrm, tex; };

// Per-chunk working set.  A frame's chunks alternate between RT_STREAMS of these, each on its own HIP
// stream, so that the short launches of one chunk (deep bounce levels with few rays, whose duration is
// the latency of one ray's path) overlap the wide launches of the other.
#define RT_STREAMS 4                    /* slots compiled in; render_streams() says how many are used */
struct Workspace {
    DevBuf sample_rgb, sample_z, sample_hit, rq[2][5], pq[3], cq[3], counts, pixel_list;
    DevBuf bvh_spill;                   // traversal-stack entries beyond the kernels' LDS stacks (only for scenes whose BVHs can need them)
    size_t samples = 0; uint32_t rq_cap = 0, pq_cap = 0;
    hipStream_t stream = nullptr;       // slot 0 runs on the caller's / the device's main stream instead
    void release()
    {
        for (DevBuf *b : {&sample_rgb, &sample_z, &sample_hit, &counts, &pixel_list, &bvh_spill}) b->release();
        for (int i = 0; i < 2; i++) for (int k = 0; k < 5; k++) rq[i][k].release();
        for (int k = 0; k < 3; k++) { pq[k].release(); cq[k].release(); }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

struct DeviceState {
    int device = -1;
    bool scene_valid = false, photons_valid = false, caustic_valid = false;
    DevBuf nodes, objects, objects_back, meshes, materials, lights, node_material, textures, texels, material_maps;
    std::vector<DevMeshBufs> mesh_bufs;
    DevBuf pa, pb, box4, grid;                  // the gather structure of the photon map (rt_photon_build.hip)
    DevBuf cpa, cpb, cbox4, cgrid;              // ... of the caustic map
    DevBuf raw_photons;                         // 24-byte photons of the last photon pass on this device (1-based)
    DevBuf bvh_spill;                           // DevScene::bvh_spill of the single-stage entry points (rt_trace_rays, the photon pass); renders use their working set's
    // per density-grid cell, the k-th squared distance of the last query k_gather answered there: predicts the next one's (a
    // hint that only steers which of two exact paths a query takes); zeroed whenever the structure is rebuilt
    DevBuf cell_rk2, ccell_rk2;
    int rk2_k = 0, rk2_k_c = 0; float rk2_radius = 0, rk2_radius_c = 0;     // the gather parameters the two hint tables were filled under (0: empty)
    DevBuf cell_start, ccell_start;             // DevPhotonMap::cell_start of the two maps (built for the gather radius in use)
    DevScene scene{};
    Workspace ws[RT_STREAMS];
    DevBuf stats;
    std::atomic<int> busy{0};           // a host call is using the working sets of this device
    // an asynchronous render (sync == 0) returns while the GPU still uses the working sets, counters and
    // stats: `last_done` is recorded at its join and every later call orders its own stream behind it
    hipEvent_t last_done = nullptr;
    bool last_pending = false;
    bool last_pipelined = false;        // the render in flight ran every slot on the library's own streams (render_tiles_once)
    uint64_t frame_seq = 0;             // rotates the slots from one pipelined frame to the next
    bool async_overflow = false;        // an asynchronous render dropped rays and nobody has collected that verdict yet (rt_render_check does)
    // How full the ray / photon-query queues of the last renders got, per sample of a chunk (device-side peaks,
    // rt_stats.peak_*): the next render of the same kind sizes its queues from that instead of the 2^bounce worst case.
    struct QueueHistory { bool valid = false; int model = -1, bounce = -1, fan = -1; bool photons = false, caustic = false;
                          double rays_per_sample = 0, queries_per_sample = 0; } qhist;
    bool qhist_used = false;            // the render in flight (or last finished) was sized from qhist / the first-frame guess
    // scratch for the single-stage entry points
    DevBuf t_in, t_out[6];
    hipStream_t stream = nullptr;
    void release()
    {
        for (DevBuf *b : {&nodes, &objects, &objects_back, &meshes, &materials, &lights, &node_material, &textures, &texels, &material_maps, &pa, &pb, &box4, &grid, &cpa, &cpb, &cbox4, &cgrid,
                          &raw_photons, &bvh_spill, &cell_rk2, &ccell_rk2, &cell_start, &ccell_start, &stats, &t_in}) b->release();
        for (auto &m : mesh_bufs) { m.nodes.release(); m.tris.release(); m.nrm.release(); m.tex.release(); }
        for (Workspace &w : ws) w.release();
        for (int k = 0; k < 6; k++) t_out[k].release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        if (last_done) (void)hipEventDestroy(last_done);
        last_done = nullptr; last_pending = false;
    }
};

// One host call at a time may use a device's working sets, scratch buffers, counters and stats: the render
// entry points AND the single-stage ones (rt_trace_rays, rt_shade_rays, rt_estimate_irradiance, rt_photon_pass)
// claim the device for their duration and fail with RT_ERR_STATE while another call (or a live job) holds it.
struct DeviceClaim {
    DeviceState *D; bool ok;
    explicit DeviceClaim(DeviceState *d) : D(d), ok(d->busy.exchange(1) == 0) {}
    ~DeviceClaim() { if (ok) D->busy.store(0); }
    DeviceClaim(const DeviceClaim &) = delete;
    DeviceClaim &operator=(const DeviceClaim &) = delete;
};

struct rt_scene {
    rt::SceneData data;
    // A photon map made by rt_scene_generate_photons is kept as generatePhotonMap leaves it BEFORE balancing (1-based,
    // [0] zero) together with the indices of the few photons LocatePhotons could not reach after balancing
    // (rt::UnreachablePhotons); data.photons (the balanced form) is produced from it only when somebody asks for it.
    // Non-empty photons_raw takes precedence over data.photons; rt_scene_set_photons clears it.
    std::vector<rt_photon> photons_raw;
    std::vector<uint32_t> photons_skip;
    // A map made by the photon pass (rt_scene_generate_photons, rt_render_begin) is DERIVED from the scene: whatever it was
    // traced through (nodes, meshes, materials, lights, another XML) changing drops it, and rt_render_begin makes a new one when
    // there is none or when the render asks for another count / bounce limit / seed -- the reference runs generatePhotonMap() on
    // every BeginRender (FIN/main.cpp:984-990).  A map handed over with rt_scene_set_photons is the caller's: it stays.
    struct Generated { bool valid = false; uint32_t count = 0; int bounce = 0; uint32_t seed = 0; } gen;
    // ... and it STAYS on the device that generated it (DeviceState::raw_photons of gen_dev, gen_n photons) until somebody needs it
    // on the host: the .dat dump, rt_scene_get_photons, another device, the host's replay of BalanceSegment when a median is not
    // unique (raw_to_host below).  gen_n != 0 with an empty photons_raw = "on gen_dev only".
    DeviceState *gen_dev = nullptr; uint32_t gen_n = 0;
    std::string photon_dump;            // where rt_render_begin's photon pass writes its .dat ("" = nowhere)
    std::mutex gen_mu;                  // jobs started together on several devices: ONE of them runs the photon pass, the others wait for it
    std::mutex mu;
    std::vector<DeviceState *> devs;
    std::atomic<int> live_jobs{0};
    uint32_t photon_count() const { return gen_n ? gen_n : (!photons_raw.empty() ? (uint32_t)photons_raw.size() - 1 : (data.photons.empty() ? 0u : (uint32_t)data.photons.size() - 1)); }
    void drop_generated() { photons_raw.clear(); photons_skip.clear(); gen.valid = false; gen_dev = nullptr; gen_n = 0; }
    void invalidate(bool scene, bool photons, bool caustic = false)
    {
        for (DeviceState *d : devs) { if (scene) d->scene_valid = false; if (photons) d->photons_valid = false; if (caustic) d->caustic_valid = false; }
    }
    // geometry, materials or lights changed (caller holds mu): re-upload, and a generated photon map is no longer this scene's
    void geometry_changed()
    {
        if (gen.valid) { drop_generated(); data.photons.clear(); invalidate(true, true); }
        else invalidate(true, false);
    }
};

struct rt_job {
    rt_scene *scene = nullptr;
    std::thread worker;
    std::atomic<int> progress{0};
    std::atomic<bool> stop{false};
    std::atomic<bool> done{false};
    rt_status status = RT_OK;
    std::string error;
    rt_stats stats{};
    // caller-owned host image (rt_render_begin): finished bands are copied back chunk by chunk, so a
    // viewer that polls rt_render_progress can show the frame as it fills (viewport.cpp:367 reads
    // renderImage.GetPixels() while the workers run)
    uint8_t *host_rgb = nullptr, *host_count = nullptr; float *host_z = nullptr;
    rt_setup_ms setup{};                // the photon pass this job ran first (all zero when it did not)
};

// ---- scene store ---------------------------------------------------------------------------------------
extern "C" rt_status rt_scene_create(rt_scene **out)
{
    if (!out) return fail(RT_ERR_ARG, "rt_scene_create: out is NULL");
    *out = new rt_scene;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    // a job that is still rendering this scene holds a pointer to it: wait for it (jobs always terminate;
    // destroying the job first, or rt_render_stop, makes this immediate)
    while (s->live_jobs.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    for (DeviceState *d : s->devs) {
        if (hipSetDevice(d->device) == hipSuccess) d->release();
        delete d;
    }
    delete s;
}

static rt_status check_idle(rt_scene *s, const char *who)
{
    if (!s) return fail(RT_ERR_ARG, "%s: scene is NULL", who);
    if (s->live_jobs.load() > 0) return fail(RT_ERR_STATE, "%s: a render job is still live on this scene", who);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_nodes(rt_scene *s, const rt_node *nodes, int32_t n)
{
    rt_status st = check_idle(s, "rt_scene_set_nodes");
    if (st) return st;
    if (n < 0 || (n > 0 && !nodes)) return fail(RT_ERR_ARG, "rt_scene_set_nodes: bad array");
    for (int32_t i = 0; i < n; i++) {
        if (i == 0 ? nodes[i].parent != -1 : (nodes[i].parent < 0 || nodes[i].parent >= i))
            return fail(RT_ERR_ARG, "rt_scene_set_nodes: node %d has parent %d (parents must precede children, root first)", i, nodes[i].parent);
        if (nodes[i].obj_type < RT_OBJ_NONE || nodes[i].obj_type > RT_OBJ_MESH)
            return fail(RT_ERR_ARG, "rt_scene_set_nodes: node %d has unknown object type %d", i, nodes[i].obj_type);
    }
    std::lock_guard<std::mutex> lk(s->mu);
    s->data.nodes.assign(nodes, nodes + n);
    s->geometry_changed();
    return RT_OK;
}

extern "C" rt_status rt_scene_set_mesh(rt_scene *s, int32_t mesh, const float *v, int32_t nv, const uint32_t *f, int32_t nf,
                                       const float *vn, int32_t nvn, const uint32_t *fn, const rt_bvh_node *nodes,
                                       int32_t nnodes, const uint32_t *elements)
{
    rt_status st = check_idle(s, "rt_scene_set_mesh");
    if (st) return st;
    if (mesh < 0 || mesh > 65535) return fail(RT_ERR_ARG, "rt_scene_set_mesh: mesh index %d out of range", mesh);
    if (nv <= 0 || nf <= 0 || nvn <= 0 || nnodes < 2 || !v || !f || !vn || !fn || !nodes || !elements)
        return fail(RT_ERR_ARG, "rt_scene_set_mesh: every array is required (nv=%d nf=%d nvn=%d nnodes=%d)", nv, nf, nvn, nnodes);
    for (int64_t i = 0; i < 3LL * nf; i++) {
        if (f[i] >= (uint32_t)nv) return fail(RT_ERR_ARG, "rt_scene_set_mesh: face index %u >= nv", f[i]);
        if (fn[i] >= (uint32_t)nvn) return fail(RT_ERR_ARG, "rt_scene_set_mesh: normal index %u >= nvn", fn[i]);
    }
    for (int32_t i = 0; i < nf; i++) if (elements[i] >= (uint32_t)nf) return fail(RT_ERR_ARG, "rt_scene_set_mesh: element %u >= nf", elements[i]);
    std::lock_guard<std::mutex> lk(s->mu);
    if ((size_t)mesh >= s->data.meshes.size()) s->data.meshes.resize((size_t)mesh + 1);
    rt::MeshData &m = s->data.meshes[mesh];
    m.v.assign(v, v + 3 * (size_t)nv); m.f.assign(f, f + 3 * (size_t)nf);
    m.vn.assign(vn, vn + 3 * (size_t)nvn); m.fn.assign(fn, fn + 3 * (size_t)nf);
    m.nodes.assign(nodes, nodes + nnodes); m.elements.assign(elements, elements + nf);
    m.vt.clear(); m.ft.clear();
    s->geometry_changed();
    return RT_OK;
}

extern "C" rt_status rt_scene_set_mesh_texcoords(rt_scene *s, int32_t mesh, const float *vt, int32_t nvt, const uint32_t *ft)
{
    rt_status st = check_idle(s, "rt_scene_set_mesh_texcoords");
    if (st) return st;
    std::lock_guard<std::mutex> lk(s->mu);
    if (mesh < 0 || (size_t)mesh >= s->data.meshes.size() || s->data.meshes[mesh].f.empty())
        return fail(RT_ERR_ARG, "rt_scene_set_mesh_texcoords: mesh %d has not been set", mesh);
    rt::MeshData &m = s->data.meshes[mesh];
    if (nvt < 0 || (nvt > 0 && (!vt || !ft))) return fail(RT_ERR_ARG, "rt_scene_set_mesh_texcoords: bad arrays");
    for (size_t i = 0; nvt > 0 && i < m.f.size(); i++)
        if (ft[i] >= (uint32_t)nvt) return fail(RT_ERR_ARG, "rt_scene_set_mesh_texcoords: texture index %u >= nvt", ft[i]);
    if (nvt == 0) { m.vt.clear(); m.ft.clear(); }
    else { m.vt.assign(vt, vt + 3 * (size_t)nvt); m.ft.assign(ft, ft + m.f.size()); }
    s->invalidate(true, false);
    return RT_OK;
}

extern "C" rt_status rt_scene_get_mesh_texcoords(const rt_scene *s, int32_t mesh, int32_t *nvt, float *vt, uint32_t *ft)
{
    if (!s || mesh < 0 || (size_t)mesh >= s->data.meshes.size()) return fail(RT_ERR_ARG, "rt_scene_get_mesh_texcoords: bad mesh index");
    const rt::MeshData &m = s->data.meshes[mesh];
    if (nvt) *nvt = (int32_t)(m.vt.size() / 3);
    if (vt) memcpy(vt, m.vt.data(), m.vt.size() * 4);
    if (ft) memcpy(ft, m.ft.data(), m.ft.size() * 4);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_materials(rt_scene *s, const rt_blinn *m, int32_t n)
{
    rt_status st = check_idle(s, "rt_scene_set_materials");
    if (st) return st;
    if (n < 0 || (n > 0 && !m)) return fail(RT_ERR_ARG, "rt_scene_set_materials: bad array");
    std::lock_guard<std::mutex> lk(s->mu);
    s->data.materials.assign(m, m + n);
    s->geometry_changed();
    return RT_OK;
}

extern "C" rt_status rt_scene_set_lights(rt_scene *s, const rt_light *l, int32_t n)
{
    rt_status st = check_idle(s, "rt_scene_set_lights");
    if (st) return st;
    if (n < 0 || (n > 0 && !l)) return fail(RT_ERR_ARG, "rt_scene_set_lights: bad array");
    for (int32_t i = 0; i < n; i++)
        if (l[i].type < RT_LIGHT_AMBIENT || l[i].type > RT_LIGHT_POINT) return fail(RT_ERR_ARG, "rt_scene_set_lights: light %d has unknown type", i);
    std::lock_guard<std::mutex> lk(s->mu);
    s->data.lights.assign(l, l + n);
    s->geometry_changed();
    return RT_OK;
}

extern "C" rt_status rt_scene_set_environment(rt_scene *s, const float env[3], const float bg[3])
{
    rt_status st = check_idle(s, "rt_scene_set_environment");
    if (st) return st;
    std::lock_guard<std::mutex> lk(s->mu);
    if (env) memcpy(s->data.env, env, 12);
    if (bg) memcpy(s->data.bg, bg, 12);
    s->invalidate(true, false);
    return RT_OK;
}

extern "C" rt_status rt_scene_get_environment(const rt_scene *s, float env[3], float bg[3])
{
    if (!s) return fail(RT_ERR_ARG, "rt_scene_get_environment: scene is NULL");
    if (env) memcpy(env, s->data.env, 12);
    if (bg) memcpy(bg, s->data.bg, 12);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_textures(rt_scene *s, const rt_texture *tex, int32_t n, const uint8_t *texels, uint64_t n_bytes)
{
    rt_status st = check_idle(s, "rt_scene_set_textures");
    if (st) return st;
    if (n < 0 || (n > 0 && !tex) || (n_bytes > 0 && !texels)) return fail(RT_ERR_ARG, "rt_scene_set_textures: bad array");
    for (int32_t i = 0; i < n; i++) {
        if (tex[i].type != RT_TEX_FILE && tex[i].type != RT_TEX_CHECKER) return fail(RT_ERR_ARG, "rt_scene_set_textures: texture %d has unknown type", i);
        if (tex[i].type == RT_TEX_FILE && (tex[i].width < 0 || tex[i].height < 0 || (uint64_t)tex[i].texel_offset + 3ull * tex[i].width * tex[i].height > n_bytes))
            return fail(RT_ERR_ARG, "rt_scene_set_textures: texture %d exceeds the texel array", i);
    }
    std::lock_guard<std::mutex> lk(s->mu);
    s->data.textures.assign(tex, tex + n);
    s->data.texels.assign(texels, texels + n_bytes);
    s->invalidate(true, false);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_material_maps(rt_scene *s, const rt_texmap *maps, int32_t n_materials)
{
    rt_status st = check_idle(s, "rt_scene_set_material_maps");
    if (st) return st;
    if (n_materials < 0 || (n_materials > 0 && !maps)) return fail(RT_ERR_ARG, "rt_scene_set_material_maps: bad array");
    std::lock_guard<std::mutex> lk(s->mu);
    s->data.material_maps.assign(maps, maps + 2 * (size_t)n_materials);
    s->invalidate(true, false);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_environment_maps(rt_scene *s, const rt_texmap *environment, const rt_texmap *background)
{
    rt_status st = check_idle(s, "rt_scene_set_environment_maps");
    if (st) return st;
    std::lock_guard<std::mutex> lk(s->mu);
    rt_texmap none;
    memset(&none, 0, sizeof none);
    none.texture = RT_MAP_NONE;
    s->data.env_map = environment ? *environment : none;
    s->data.bg_map = background ? *background : none;
    s->invalidate(true, false);
    return RT_OK;
}

extern "C" rt_status rt_scene_get_textures(const rt_scene *s, rt_texture *tex, int32_t cap, uint8_t *texels, uint64_t texel_cap,
                                           int32_t *n_tex, uint64_t *n_bytes)
{
    if (!s) return fail(RT_ERR_ARG, "rt_scene_get_textures: scene is NULL");
    if (n_tex) *n_tex = (int32_t)s->data.textures.size();
    if (n_bytes) *n_bytes = s->data.texels.size();
    if (tex) { if ((size_t)cap < s->data.textures.size()) return fail(RT_ERR_ARG, "rt_scene_get_textures: capacity"); if (!s->data.textures.empty()) memcpy(tex, s->data.textures.data(), s->data.textures.size() * sizeof(rt_texture)); }
    if (texels) { if (texel_cap < s->data.texels.size()) return fail(RT_ERR_ARG, "rt_scene_get_textures: texel capacity"); if (!s->data.texels.empty()) memcpy(texels, s->data.texels.data(), s->data.texels.size()); }
    return RT_OK;
}

extern "C" rt_status rt_scene_get_maps(const rt_scene *s, rt_texmap *material_maps, int32_t cap, rt_texmap *environment, rt_texmap *background)
{
    if (!s) return fail(RT_ERR_ARG, "rt_scene_get_maps: scene is NULL");
    if (material_maps) {
        if ((size_t)cap < s->data.material_maps.size()) return fail(RT_ERR_ARG, "rt_scene_get_maps: capacity %d < %zu", cap, s->data.material_maps.size());
        if (!s->data.material_maps.empty()) memcpy(material_maps, s->data.material_maps.data(), s->data.material_maps.size() * sizeof(rt_texmap));
    }
    if (environment) *environment = s->data.env_map;
    if (background) *background = s->data.bg_map;
    return RT_OK;
}

extern "C" rt_status rt_image_read_rgb(const char *path, int32_t *w, int32_t *h, uint8_t *rgb, uint64_t cap)
{
    if (!path || !w || !h) return fail(RT_ERR_ARG, "rt_image_read_rgb: NULL argument");
    int iw = 0, ih = 0;
    std::vector<uint8_t> d;
    std::string err;
    if (!rt::ReadImageRGB(path, iw, ih, d, &err)) return fail(RT_ERR_IO, "rt_image_read_rgb(%s): %s", path, err.c_str());
    *w = iw; *h = ih;
    if (rgb) {
        if (cap < d.size()) return fail(RT_ERR_ARG, "rt_image_read_rgb: buffer of %llu bytes, need %zu", (unsigned long long)cap, d.size());
        memcpy(rgb, d.data(), d.size());
    }
    return RT_OK;
}

extern "C" rt_status rt_image_write_png(const char *path, const uint8_t *data, int32_t w, int32_t h, int32_t comps)
{
    if (!path || !data || w <= 0 || h <= 0) return fail(RT_ERR_ARG, "rt_image_write_png: bad argument");
    if (!rt::WritePNG(path, data, w, h, comps)) return fail(RT_ERR_IO, "rt_image_write_png(%s): cannot write (comps must be 1 or 3)", path);
    return RT_OK;
}

extern "C" rt_status rt_image_zbuffer(const float *zbuffer, int32_t w, int32_t h, uint8_t *zbuffer_img)
{
    if (!zbuffer || !zbuffer_img || w <= 0 || h <= 0) return fail(RT_ERR_ARG, "rt_image_zbuffer: bad argument");
    rt::ZBufferImage(zbuffer, (size_t)w * (size_t)h, zbuffer_img);
    return RT_OK;
}

extern "C" rt_status rt_image_sample_count(const uint8_t *sample_count, int32_t w, int32_t h, uint8_t *sample_count_img, int32_t *smax)
{
    if (!sample_count || !sample_count_img || w <= 0 || h <= 0) return fail(RT_ERR_ARG, "rt_image_sample_count: bad argument");
    const int m = rt::SampleCountImage(sample_count, (size_t)w * (size_t)h, sample_count_img);
    if (smax) *smax = m;
    return RT_OK;
}

extern "C" rt_status rt_scene_set_photons(rt_scene *s, const rt_photon *photons, uint32_t n_stored)
{
    rt_status st = check_idle(s, "rt_scene_set_photons");
    if (st) return st;
    if (n_stored > 0 && !photons) return fail(RT_ERR_ARG, "rt_scene_set_photons: photons is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    s->drop_generated();
    if (n_stored == 0) s->data.photons.clear();
    else s->data.photons.assign(photons, photons + (size_t)n_stored + 1);
    s->invalidate(false, true);
    return RT_OK;
}

extern "C" rt_status rt_scene_set_caustic_photons(rt_scene *s, const rt_photon *photons, uint32_t n_stored)
{
    rt_status st = check_idle(s, "rt_scene_set_caustic_photons");
    if (st) return st;
    if (n_stored > 0 && !photons) return fail(RT_ERR_ARG, "rt_scene_set_caustic_photons: photons is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    if (n_stored == 0) s->data.caustic_photons.clear();
    else s->data.caustic_photons.assign(photons, photons + (size_t)n_stored + 1);
    s->invalidate(false, false, true);
    return RT_OK;
}

extern "C" rt_status rt_scene_load_xml(rt_scene *s, const char *path)
{
    rt_status st = check_idle(s, "rt_scene_load_xml");
    if (st) return st;
    if (!path) return fail(RT_ERR_ARG, "rt_scene_load_xml: path is NULL");
    rt::Scene graph;
    std::string err;
    if (!rt::LoadScene(graph, path, &err)) return fail(RT_ERR_IO, "rt_scene_load_xml(%s): %s", path, err.c_str());
    rt::SceneData d;
    if (!rt::Lower(graph, d, &err)) return fail(RT_ERR_ARG, "rt_scene_load_xml(%s): %s", path, err.c_str());
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->gen.valid) d.photons = s->data.photons;      // a map the caller set stays; a generated one belonged to the old scene
    s->data = std::move(d);
    s->geometry_changed();
    return RT_OK;
}

extern "C" rt_status rt_scene_get_camera(const rt_scene *s, rt_camera *out)
{
    if (!s || !out) return fail(RT_ERR_ARG, "rt_scene_get_camera: NULL argument");
    if (!s->data.has_camera) return fail(RT_ERR_STATE, "rt_scene_get_camera: the scene has no camera (not loaded from XML)");
    *out = s->data.camera;
    return RT_OK;
}

extern "C" rt_status rt_scene_counts(const rt_scene *s, int32_t *n_nodes, int32_t *n_meshes, int32_t *n_materials,
                                     int32_t *n_lights, uint32_t *n_photons)
{
    if (!s) return fail(RT_ERR_ARG, "rt_scene_counts: scene is NULL");
    if (n_nodes) *n_nodes = (int32_t)s->data.nodes.size();
    if (n_meshes) *n_meshes = (int32_t)s->data.meshes.size();
    if (n_materials) *n_materials = (int32_t)s->data.materials.size();
    if (n_lights) *n_lights = (int32_t)s->data.lights.size();
    if (n_photons) *n_photons = s->photon_count();
    return RT_OK;
}

template <class T> static rt_status copy_out(const std::vector<T> &v, T *out, int32_t cap, const char *who)
{
    if (!out) return fail(RT_ERR_ARG, "%s: out is NULL", who);
    if ((size_t)cap < v.size()) return fail(RT_ERR_ARG, "%s: capacity %d < %zu", who, cap, v.size());
    if (!v.empty()) memcpy(out, v.data(), sizeof(T) * v.size());
    return RT_OK;
}
extern "C" rt_status rt_scene_get_nodes(const rt_scene *s, rt_node *out, int32_t cap) { return s ? copy_out(s->data.nodes, out, cap, "rt_scene_get_nodes") : fail(RT_ERR_ARG, "NULL scene"); }
extern "C" rt_status rt_scene_get_materials(const rt_scene *s, rt_blinn *out, int32_t cap) { return s ? copy_out(s->data.materials, out, cap, "rt_scene_get_materials") : fail(RT_ERR_ARG, "NULL scene"); }
extern "C" rt_status rt_scene_get_lights(const rt_scene *s, rt_light *out, int32_t cap) { return s ? copy_out(s->data.lights, out, cap, "rt_scene_get_lights") : fail(RT_ERR_ARG, "NULL scene"); }

extern "C" rt_status rt_scene_mesh_counts(const rt_scene *s, int32_t mesh, int32_t *nv, int32_t *nf, int32_t *nvn, int32_t *nnodes)
{
    if (!s || mesh < 0 || (size_t)mesh >= s->data.meshes.size()) return fail(RT_ERR_ARG, "rt_scene_mesh_counts: bad mesh index");
    const rt::MeshData &m = s->data.meshes[mesh];
    if (nv) *nv = (int32_t)(m.v.size() / 3);
    if (nf) *nf = (int32_t)(m.f.size() / 3);
    if (nvn) *nvn = (int32_t)(m.vn.size() / 3);
    if (nnodes) *nnodes = (int32_t)m.nodes.size();
    return RT_OK;
}

extern "C" rt_status rt_scene_get_mesh(const rt_scene *s, int32_t mesh, float *v, uint32_t *f, float *vn, uint32_t *fn,
                                       rt_bvh_node *nodes, uint32_t *elements)
{
    if (!s || mesh < 0 || (size_t)mesh >= s->data.meshes.size()) return fail(RT_ERR_ARG, "rt_scene_get_mesh: bad mesh index");
    const rt::MeshData &m = s->data.meshes[mesh];
    if (v) memcpy(v, m.v.data(), m.v.size() * 4);
    if (f) memcpy(f, m.f.data(), m.f.size() * 4);
    if (vn) memcpy(vn, m.vn.data(), m.vn.size() * 4);
    if (fn) memcpy(fn, m.fn.data(), m.fn.size() * 4);
    if (nodes) memcpy(nodes, m.nodes.data(), m.nodes.size() * sizeof(rt_bvh_node));
    if (elements) memcpy(elements, m.elements.data(), m.elements.size() * 4);
    return RT_OK;
}

extern "C" rt_status rt_bvh_build(const float *v, int32_t nv, const uint32_t *f, int32_t nf, int32_t max_per_leaf,
                                  rt_bvh_node *nodes_out, int32_t *nnodes, uint32_t *elements_out)
{
    if (!v || !f || nf <= 0 || nv <= 0 || !nodes_out || !nnodes || !elements_out) return fail(RT_ERR_ARG, "rt_bvh_build: NULL/empty argument");
    for (int64_t i = 0; i < 3LL * nf; i++) if (f[i] >= (uint32_t)nv) return fail(RT_ERR_ARG, "rt_bvh_build: face index out of range");
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> el;
    rt::BuildMeanSplitBVH(v, f, (unsigned)nf, (unsigned)max_per_leaf, nodes, el);
    memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rt_bvh_node));
    memcpy(elements_out, el.data(), el.size() * 4);
    *nnodes = (int32_t)nodes.size();
    return RT_OK;
}

extern "C" rt_status rt_photons_write_dat(const char *path, const rt_photon *photons, uint32_t n)
{
    if (!path || (n > 0 && !photons)) return fail(RT_ERR_ARG, "rt_photons_write_dat: NULL argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return fail(RT_ERR_IO, "rt_photons_write_dat: cannot create \"%s\"", path);
    const size_t w = n ? fwrite(photons + 1, sizeof(rt_photon), n, fp) : 0;
    const bool ok = (fclose(fp) == 0) && w == n;
    return ok ? RT_OK : fail(RT_ERR_IO, "rt_photons_write_dat: short write to \"%s\"", path);
}

extern "C" rt_status rt_photons_read_dat(const char *path, rt_photon *out, uint32_t cap, uint32_t *n)
{
    if (!path || !n) return fail(RT_ERR_ARG, "rt_photons_read_dat: NULL argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(RT_ERR_IO, "rt_photons_read_dat: cannot open \"%s\"", path);
    fseek(fp, 0, SEEK_END);
    const long bytes = ftell(fp);
    rewind(fp);
    const uint64_t count = bytes > 0 ? (uint64_t)bytes / sizeof(rt_photon) : 0;
    if (count > 0xFFFFFFF0ull) { fclose(fp); return fail(RT_ERR_ARG, "rt_photons_read_dat: \"%s\" holds too many photons", path); }
    *n = (uint32_t)count;
    rt_status st = RT_OK;
    if (out) {
        if (cap < count + 1) st = fail(RT_ERR_ARG, "rt_photons_read_dat: buffer of %u records, %llu needed", cap, (unsigned long long)count + 1);
        else {
            memset(out, 0, sizeof(rt_photon));
            if (count && fread(out + 1, sizeof(rt_photon), count, fp) != count) st = fail(RT_ERR_IO, "rt_photons_read_dat: short read from \"%s\"", path);
        }
    }
    fclose(fp);
    return st;
}

extern "C" rt_status rt_photon_balance(rt_photon *in, uint32_t n, rt_photon *out)
{
    if (!in || !out) return fail(RT_ERR_ARG, "rt_photon_balance: NULL argument");
    rt::BalancePhotons(in, n, out);
    return RT_OK;
}

extern "C" rt_status rt_photon_unreachable(const rt_photon *in, uint32_t n, uint32_t *raw_indices, uint32_t cap, uint32_t *count)
{
    if (!in || !count) return fail(RT_ERR_ARG, "rt_photon_unreachable: NULL argument");
    std::vector<uint32_t> idx;
    rt::UnreachablePhotons(in, n, idx);
    *count = (uint32_t)idx.size();
    if (raw_indices) {
        if (cap < idx.size()) return fail(RT_ERR_ARG, "rt_photon_unreachable: room for %u indices, %zu needed", cap, idx.size());
        if (!idx.empty()) memcpy(raw_indices, idx.data(), idx.size() * 4);
    }
    return RT_OK;
}

extern "C" rt_status rt_photon_unreachable_device(int device, const rt_photon *in, uint32_t n, uint32_t *raw_indices, uint32_t cap, uint32_t *count, int32_t *exact)
{
    if (!in || !count || !exact) return fail(RT_ERR_ARG, "rt_photon_unreachable_device: NULL argument");
    *count = 0; *exact = 1;
    if (!device_is_gfx950(device)) return fail(RT_ERR_NO_DEVICE, "rt_photon_unreachable_device: device %d is not gfx950 (no CPU path: rt_photon_unreachable is the host's)", device);
    const uint32_t reach = rt::ReachablePhotonSlots(n);
    if (n == 0 || reach >= n) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    struct TempBuf : DevBuf { ~TempBuf() { release(); } } ph, scratch;
    rt_status st;
    if ((st = ph.upload(in, ((size_t)n + 1) * sizeof(rt_photon)))) return st;
    HIP_TRY(hipMemset(ph.p, 0, sizeof(rt_photon)));                  // slot 0 is the unused all-zero photon
    const size_t sbytes = rtk_photon_unreachable_scratch(n);
    if ((st = scratch.ensure(sbytes))) return st;
    uint32_t res[16];
    const hipError_t e = rtk_photon_unreachable(nullptr, (const rt_photon *)ph.p, n, reach + 1, n, scratch.p, sbytes, res);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_photon_unreachable_device: %s", hipGetErrorString(e));
    if (res[1] != 0 || res[0] != n - reach) { *exact = 0; return RT_OK; }
    std::vector<uint32_t> idx(res + 2, res + 2 + res[0]);
    std::sort(idx.begin(), idx.end());
    *count = (uint32_t)idx.size();
    if (raw_indices) {
        if (cap < idx.size()) return fail(RT_ERR_ARG, "rt_photon_unreachable_device: room for %u indices, %zu needed", cap, idx.size());
        memcpy(raw_indices, idx.data(), idx.size() * 4);
    }
    return RT_OK;
}

// ---- lowering to the device ---------------------------------------------------------------------------
static const size_t SPILL_BYTES = (size_t)RT_SPILL_BLOCKS * RT_BLOCK * RT_BVH_SPILL * 4;     // DevScene::bvh_spill / DevWork::bvh_spill
static DeviceState *device_state(rt_scene *s, int device)
{
    for (DeviceState *d : s->devs) if (d->device == device) return d;
    DeviceState *d = new DeviceState;
    d->device = device;
    s->devs.push_back(d);
    return d;
}

// The tree the kernels walk is built HERE from the triangles (binned surface-area heuristic, leaves of at most four, collapsed
// four-wide by largest child first): any bounding hierarchy over the same triangles gives the same closest hit -- the
// triangle tests are the reference's, operation by operation, and `t < z` keeps the nearest whatever the order (only two
// hits at EXACTLY equal t depend on it) -- so the reference's mean-split tree (rt_bvh_build, rt_scene_set_mesh: kept bit-identical
// for whoever reads it back) is not what limits traversal any more.  slot_face: the face of every triangle slot (leaf order).
static rt_status build_sah_bvh(const rt::MeshData &m, std::vector<DevBvhNode> &out, std::vector<uint32_t> &slot_face, uint32_t &root_ref, int &stack_need)
{
    const size_t nf = m.f.size() / 3;
    struct TB { float lo[3], hi[3], c[3]; };
    std::vector<TB> tb(nf);
    for (size_t f = 0; f < nf; f++) {
        TB &t = tb[f];
        for (int a = 0; a < 3; a++) { t.lo[a] = 3.0e38f; t.hi[a] = -3.0e38f; }
        for (int k = 0; k < 3; k++) {
            const float *q = &m.v[3 * (size_t)m.f[3 * f + k]];
            for (int a = 0; a < 3; a++) { t.lo[a] = std::min(t.lo[a], q[a]); t.hi[a] = std::max(t.hi[a], q[a]); }
        }
        for (int a = 0; a < 3; a++) t.c[a] = 0.5f * (t.lo[a] + t.hi[a]);
    }
    struct BN { float lo[3], hi[3]; int32_t left, right; uint32_t first, count; };      // binary tree: leaf when left < 0
    std::vector<BN> bn;
    std::vector<uint32_t> idx(nf);
    for (size_t i = 0; i < nf; i++) idx[i] = (uint32_t)i;
    auto area = [](const float *lo, const float *hi) { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return 2.0f * (dx * dy + dy * dz + dz * dx); };
    struct Rec {
        std::vector<TB> &tb; std::vector<BN> &bn; std::vector<uint32_t> &idx; decltype(area) &area;
        int32_t go(uint32_t b, uint32_t e, int depth)
        {
            BN n;
            for (int a = 0; a < 3; a++) { n.lo[a] = 3.0e38f; n.hi[a] = -3.0e38f; }
            float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (uint32_t i = b; i < e; i++) {
                const TB &t = tb[idx[i]];
                for (int a = 0; a < 3; a++) { n.lo[a] = std::min(n.lo[a], t.lo[a]); n.hi[a] = std::max(n.hi[a], t.hi[a]); clo[a] = std::min(clo[a], t.c[a]); chi[a] = std::max(chi[a], t.c[a]); }
            }
            n.left = n.right = -1; n.first = b; n.count = e - b;
            const int32_t me = (int32_t)bn.size();
            bn.push_back(n);
            // leaves of at most four triangles (measured on MI355X, tracer ms Cornell / 102 k triangles / C3: 2: 15.45 / 21.77 / 202.4, 3: 15.46 / 21.79 /
            // 201.1, 4: 15.51 / 22.22 / 196.5, 6: 15.62 / 22.63 / 202.6, 8: 15.87 / 23.65 / 213.6), its triangles by face id
            if (e - b <= 4) { std::sort(idx.begin() + b, idx.begin() + e); return me; }
            uint32_t mid = 0;
            if (depth < 28) {
                // binned SAH over the centroids, 16 bins per axis
                const int NB = 16;
                float best = 3.0e38f; int best_axis = -1, best_bin = 0;
                for (int a = 0; a < 3; a++) {
                    const float ext = chi[a] - clo[a];
                    if (!(ext > 0)) continue;
                    struct Bin { float lo[3], hi[3]; uint32_t n; } bins[NB];
                    for (int k = 0; k < NB; k++) { bins[k].n = 0; for (int c = 0; c < 3; c++) { bins[k].lo[c] = 3.0e38f; bins[k].hi[c] = -3.0e38f; } }
                    const float scale = (float)NB / ext;
                    for (uint32_t i = b; i < e; i++) {
                        const TB &t = tb[idx[i]];
                        int k = (int)((t.c[a] - clo[a]) * scale);
                        k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                        bins[k].n++;
                        for (int c = 0; c < 3; c++) { bins[k].lo[c] = std::min(bins[k].lo[c], t.lo[c]); bins[k].hi[c] = std::max(bins[k].hi[c], t.hi[c]); }
                    }
                    float rlo[NB][3], rhi[NB][3]; uint32_t rn[NB];
                    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}; uint32_t cnt = 0;
                    for (int k = NB - 1; k >= 1; k--) {
                        cnt += bins[k].n;
                        for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], bins[k].lo[c]); hi[c] = std::max(hi[c], bins[k].hi[c]); }
                        rn[k] = cnt; for (int c = 0; c < 3; c++) { rlo[k][c] = lo[c]; rhi[k][c] = hi[c]; }
                    }
                    for (int c = 0; c < 3; c++) { lo[c] = 3.0e38f; hi[c] = -3.0e38f; }
                    cnt = 0;
                    for (int k = 0; k + 1 < NB; k++) {               // split between bin k and k + 1
                        cnt += bins[k].n;
                        for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], bins[k].lo[c]); hi[c] = std::max(hi[c], bins[k].hi[c]); }
                        if (cnt == 0 || rn[k + 1] == 0) continue;
                        const float cost = area(lo, hi) * (float)cnt + area(rlo[k + 1], rhi[k + 1]) * (float)rn[k + 1];
                        if (cost < best) { best = cost; best_axis = a; best_bin = k; }
                    }
                }
                if (best_axis >= 0) {
                    const int a = best_axis;
                    const float scale = 16.0f / (chi[a] - clo[a]);
                    auto it = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t f) {
                        int k = (int)((tb[f].c[a] - clo[a]) * scale);
                        k = k < 0 ? 0 : (k >= 16 ? 15 : k);
                        return k <= best_bin;
                    });
                    mid = (uint32_t)(it - idx.begin());
                }
            }
            if (mid <= b || mid >= e) {
                // no usable plane (all centroids in one bin, or the tree got deep): halves by the widest centroid axis
                int a = 0;
                if (chi[1] - clo[1] > chi[a] - clo[a]) a = 1;
                if (chi[2] - clo[2] > chi[a] - clo[a]) a = 2;
                mid = b + (e - b) / 2;
                std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e, [&](uint32_t x, uint32_t y) { return tb[x].c[a] < tb[y].c[a] || (tb[x].c[a] == tb[y].c[a] && x < y); });
            }
            const int32_t l = go(b, mid, depth + 1);
            const int32_t r = go(mid, e, depth + 1);
            bn[me].left = l; bn[me].right = r;
            return me;
        }
    } rec{tb, bn, idx, area};
    const int32_t root = rec.go(0, (uint32_t)nf, 0);
    auto leafref = [](uint32_t off, uint32_t cnt) { return 0x80000000u | ((cnt - 1) << 28) | (off & 0x0FFFFFFFu); };
    slot_face = idx;
    if (nf >= 0x10000000ull) return fail(RT_ERR_LIMIT, "mesh too large");
    int max_levels = 0;
    // four-wide: a device node holds up to four descendants of the binary node it stands for -- starting from its two children, the
    // internal one with the largest surface is replaced by its own two children until there are four (or only leaves)
    struct Col {
        std::vector<BN> &bn; std::vector<DevBvhNode> &out; decltype(area) &area; decltype(leafref) &leafref; int &max_levels;
        uint32_t go(int32_t id, int level)
        {
            if (level > max_levels) max_levels = level;
            const uint32_t me = (uint32_t)out.size();
            out.push_back(DevBvhNode{});
            int32_t ch[4]; int n = 0;
            ch[n++] = bn[id].left; ch[n++] = bn[id].right;
            while (n < 4) {
                int pick = -1; float big = -1.0f;
                for (int i = 0; i < n; i++) if (bn[ch[i]].left >= 0) { const float s = area(bn[ch[i]].lo, bn[ch[i]].hi); if (s > big) { big = s; pick = i; } }
                if (pick < 0) break;
                const int32_t c = ch[pick];
                for (int i = n; i > pick + 1; i--) ch[i] = ch[i - 1];
                ch[pick] = bn[c].left; ch[pick + 1] = bn[c].right;
                n++;
            }
            DevBvhNode d;
            memset(&d, 0, sizeof d);
            const float nan = std::nanf("");
            for (int i = 0; i < 4; i++) {
                for (int a = 0; a < 3; a++) { d.lo[a][i] = i < n ? bn[ch[i]].lo[a] : nan; d.hi[a][i] = i < n ? bn[ch[i]].hi[a] : nan; }
                if (i >= n) d.c[i] = leafref(0, 1);
                else if (bn[ch[i]].left < 0) d.c[i] = leafref(bn[ch[i]].first, bn[ch[i]].count);
                else d.c[i] = go(ch[i], level + 1);
            }
            out[me] = d;
            return me;
        }
    } col{bn, out, area, leafref, max_levels};
    if (bn[root].left < 0) { root_ref = leafref(bn[root].first, bn[root].count); stack_need = 0; }
    else { root_ref = col.go(root, 1); stack_need = 3 * max_levels; }
    return RT_OK;
}

static rt_status upload_scene(rt_scene *s, DeviceState *D)
{
    const rt::SceneData &sd = s->data;
    if (sd.nodes.empty()) return fail(RT_ERR_STATE, "scene has no nodes");
    if (sd.nodes.size() > 65535) return fail(RT_ERR_LIMIT, "too many scene nodes (%zu)", sd.nodes.size());
    if (sd.materials.size() > 65535) return fail(RT_ERR_LIMIT, "too many materials (%zu): ray records carry 16-bit material indices", sd.materials.size());
    const int nn = (int)sd.nodes.size();
    std::vector<DevNodeXf> xf(nn);
    std::vector<int32_t> node_mat(nn, 0);
    std::vector<DevObject> objs;
    std::vector<DevObjectBack> backs;
    auto xf_of = [](const rt_node &n) { DevNodeXf x; memcpy(x.itm, n.itm, 36); memcpy(x.pos, n.pos, 12); memcpy(x.tm, n.tm, 36); x.pad[0] = x.pad[1] = x.pad[2] = 0; return x; };
    for (int i = 0; i < nn; i++) {
        const rt_node &n = sd.nodes[i];
        xf[i] = xf_of(n);
        if (n.obj_type == RT_OBJ_NONE) continue;
        if (n.material < 0 || (size_t)n.material >= sd.materials.size())
            return fail(RT_ERR_ARG, "node %d carries an object but its material index %d is invalid", i, n.material);
        node_mat[i] = n.material;
        DevObject o{};
        o.type = n.obj_type; o.mesh = -1; o.material = n.material; o.node = i;
        if (n.obj_type == RT_OBJ_MESH) {
            if (n.mesh < 0 || (size_t)n.mesh >= sd.meshes.size() || sd.meshes[n.mesh].f.empty())
                return fail(RT_ERR_ARG, "node %d references mesh %d which was never set", i, n.mesh);
            o.mesh = n.mesh;
        }
        int chain[64], len = 0;
        for (int a = i; a >= 0; a = sd.nodes[a].parent) { if (len >= 64) break; chain[len++] = a; }
        if (len > RT_MAX_DEPTH) return fail(RT_ERR_LIMIT, "node %d is nested %d deep; the device supports %d levels", i, len, RT_MAX_DEPTH);
        o.chain_len = len;
        for (int c = 0; c < len; c++) o.chain[c] = chain[len - 1 - c];      // root .. self
        memcpy(o.own_itm, n.itm, 36); memcpy(o.own_pos, n.pos, 12);         // the node's own ToNodeCoords transform, beside the rest
        {   // bounds in the ROOT node's coordinates (where every ray is taken first): the 8 corners of the
            // local extent through tm*p + pos of self .. the root's child (double), inflated by 1e-3 of
            // the extent (the exact test runs in float in local space)
            double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};                 // unit sphere; unit square at z = 0
            if (n.obj_type == RT_OBJ_PLANE) lo[2] = hi[2] = 0;
            if (n.obj_type == RT_OBJ_MESH) {
                const float *b = sd.meshes[n.mesh].nodes[1].box;
                for (int a = 0; a < 3; a++) { lo[a] = b[a]; hi[a] = b[a + 3]; }
            }
            double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            for (int corner = 0; corner < 8; corner++) {
                double q[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
                for (int c = 0; c + 1 < len; c++) {                          // chain[] here is still self .. root
                    const rt_node &X = sd.nodes[chain[c]];
                    const double r[3] = {q[0] * X.tm[0] + q[1] * X.tm[3] + q[2] * X.tm[6] + X.pos[0],
                                         q[0] * X.tm[1] + q[1] * X.tm[4] + q[2] * X.tm[7] + X.pos[1],
                                         q[0] * X.tm[2] + q[1] * X.tm[5] + q[2] * X.tm[8] + X.pos[2]};
                    q[0] = r[0]; q[1] = r[1]; q[2] = r[2];
                }
                for (int a = 0; a < 3; a++) { if (!std::isfinite(q[a])) finite = false; wlo[a] = std::min(wlo[a], q[a]); whi[a] = std::max(whi[a], q[a]); }
            }
            const double ext = std::max(std::max(whi[0] - wlo[0], whi[1] - wlo[1]), whi[2] - wlo[2]);
            const double pad = 1e-3 * ext + 1e-5;
            for (int a = 0; a < 3; a++) {
                // a degenerate transform (non-finite corner) disables the cull for this object
                o.wlo[a] = finite ? (float)(wlo[a] - pad) : -3.0e38f;
                o.whi[a] = finite ? (float)(whi[a] + pad) : 3.0e38f;
            }
        }
        objs.push_back(o);
        DevObjectBack b;
        memset(&b, 0, sizeof b);
        b.chain_len = len; b.node = i;
        for (int k = 0; k < RT_BACK_LEVELS && k < len - 1; k++) b.lvl[k] = xf_of(sd.nodes[chain[k]]);     // chain[] is self .. root
        backs.push_back(b);
    }
    if (objs.size() > RT_MAX_OBJECTS) return fail(RT_ERR_LIMIT, "too many objects (%zu)", objs.size());

    rt_status st;
    if ((st = D->nodes.upload(xf.data(), xf.size() * sizeof(DevNodeXf)))) return st;
    if ((st = D->objects.upload(objs.data(), objs.size() * sizeof(DevObject)))) return st;
    if ((st = D->objects_back.upload(backs.data(), backs.size() * sizeof(DevObjectBack)))) return st;
    if ((st = D->node_material.upload(node_mat.data(), node_mat.size() * 4))) return st;
    if ((st = D->materials.upload(sd.materials.data(), sd.materials.size() * sizeof(rt_blinn)))) return st;
    if ((st = D->lights.upload(sd.lights.data(), sd.lights.size() * sizeof(rt_light)))) return st;

    for (auto &mb : D->mesh_bufs) { mb.nodes.release(); mb.tris.release(); mb.nrm.release(); mb.tex.release(); }
    D->mesh_bufs.assign(sd.meshes.size(), DevMeshBufs());
    std::vector<DevMesh> dm(sd.meshes.size());
    int max_depth = 0;
    for (size_t mi = 0; mi < sd.meshes.size(); mi++) {
        const rt::MeshData &m = sd.meshes[mi];
        memset(&dm[mi], 0, sizeof(DevMesh));
        if (m.f.empty()) continue;
        std::vector<DevBvhNode> bn;
        uint32_t root_ref = 0;
        int depth = 0;
        std::vector<uint32_t> slot_face;
        if ((st = build_sah_bvh(m, bn, slot_face, root_ref, depth))) return st;
        // (the smallest LDS stack of any tracing kernel is RT_BVH_LDS entries; RT_BVH_SPILL more per thread wait in HBM: spill buffers below)
        if (depth > RT_BVH_LDS + RT_BVH_SPILL) return fail(RT_ERR_LIMIT, "mesh %zu: the BVH can need %d traversal-stack entries, the device provides %d", mi, depth, RT_BVH_LDS + RT_BVH_SPILL);
        max_depth = std::max(max_depth, depth);
        const size_t nf = m.f.size() / 3;
        std::vector<DevTri> tris(nf);
        std::vector<uint32_t> tri_face(nf);
        std::vector<float> nrm(9 * nf);
        for (size_t sidx = 0; sidx < nf; sidx++) {
            const uint32_t face = slot_face[sidx];
            tri_face[sidx] = face;
            rt::Point3 P[3];
            for (int k = 0; k < 3; k++) { const float *q = &m.v[3 * (size_t)m.f[3 * (size_t)face + k]]; P[k] = rt::Point3(q[0], q[1], q[2]); }
            rt::Point3 N = (P[1] - P[0]).Cross(P[2] - P[0]);      // FIN/include/objects.h:234-235
            N.Normalize();
            DevTri &T = tris[sidx];
            T.A[0] = P[0].x; T.A[1] = P[0].y; T.A[2] = P[0].z; T.B[0] = P[1].x; T.B[1] = P[1].y; T.B[2] = P[1].z;
            T.C[0] = P[2].x; T.C[1] = P[2].y; T.C[2] = P[2].z; T.N[0] = N.x; T.N[1] = N.y; T.N[2] = N.z;
        }
        for (size_t sidx = 0; sidx < nf; sidx++)                 // per triangle SLOT (leaf order), like tris
            for (int k = 0; k < 3; k++) memcpy(&nrm[9 * sidx + 3 * k], &m.vn[3 * (size_t)m.fn[3 * (size_t)tri_face[sidx] + k]], 12);
        DevMeshBufs &mb = D->mesh_bufs[mi];
        if ((st = mb.nodes.upload(bn.data(), bn.size() * sizeof(DevBvhNode)))) return st;
        if ((st = mb.tris.upload(tris.data(), tris.size() * sizeof(DevTri)))) return st;
        if ((st = mb.nrm.upload(nrm.data(), nrm.size() * 4))) return st;
        dm[mi].tex = nullptr;
        if (!m.vt.empty() && m.ft.size() == m.f.size()) {      // vt[ft0], vt[ft1], vt[ft2] per triangle slot, like nrm
            std::vector<float> tex(9 * nf);
            for (size_t sidx = 0; sidx < nf; sidx++)
                for (int k = 0; k < 3; k++) memcpy(&tex[9 * sidx + 3 * k], &m.vt[3 * (size_t)m.ft[3 * (size_t)tri_face[sidx] + k]], 12);
            if ((st = mb.tex.upload(tex.data(), tex.size() * 4))) return st;
            dm[mi].tex = (const float *)mb.tex.p;
        }
        dm[mi].nodes = (const DevBvhNode *)mb.nodes.p; dm[mi].tris = (const DevTri *)mb.tris.p;
        dm[mi].nrm = (const float *)mb.nrm.p;
        memcpy(dm[mi].root_box, m.nodes[1].box, 24);
        dm[mi].root_ref = root_ref; dm[mi].n_tris = (uint32_t)nf;
    }
    if ((st = D->meshes.upload(dm.data(), dm.size() * sizeof(DevMesh)))) return st;

    DevScene &S = D->scene;
    S.nodes = (const DevNodeXf *)D->nodes.p; S.objects = (const DevObject *)D->objects.p; S.objects_back = (const DevObjectBack *)D->objects_back.p;
    S.meshes = (const DevMesh *)D->meshes.p; S.materials = (const rt_blinn *)D->materials.p;
    S.lights = (const rt_light *)D->lights.p; S.node_material = (const int32_t *)D->node_material.p;
    S.n_nodes = nn; S.n_objects = (int)objs.size(); S.n_meshes = (int)sd.meshes.size();
    S.n_materials = (int)sd.materials.size(); S.n_lights = (int)sd.lights.size();
    memcpy(S.env, sd.env, 12); memcpy(S.bg, sd.bg, 12);
    // textures and maps
    for (const rt_texture &t : sd.textures)
        if (t.type == RT_TEX_FILE && (t.width < 0 || t.height < 0 || (uint64_t)t.texel_offset + 3ull * t.width * t.height > sd.texels.size()))
            return fail(RT_ERR_ARG, "texture texels out of range");
    if (!sd.material_maps.empty() && sd.material_maps.size() != 2 * sd.materials.size())
        return fail(RT_ERR_ARG, "material maps: need 2 per material (%zu given for %zu materials)", sd.material_maps.size(), sd.materials.size());
    auto check_map = [&](const rt_texmap &m) { return m.texture == RT_MAP_NONE || m.texture == RT_MAP_EMPTY || (m.texture >= 0 && (size_t)m.texture < sd.textures.size()); };
    for (const rt_texmap &m : sd.material_maps) if (!check_map(m)) return fail(RT_ERR_ARG, "material map refers to texture %d", m.texture);
    if (!check_map(sd.env_map) || !check_map(sd.bg_map)) return fail(RT_ERR_ARG, "environment/background map refers to a missing texture");
    if ((st = D->textures.upload(sd.textures.data(), sd.textures.size() * sizeof(rt_texture)))) return st;
    if ((st = D->texels.upload(sd.texels.data(), sd.texels.size()))) return st;
    if ((st = D->material_maps.upload(sd.material_maps.data(), sd.material_maps.size() * sizeof(rt_texmap)))) return st;
    S.textures = (const rt_texture *)D->textures.p; S.texels = (const uint8_t *)D->texels.p; S.n_textures = (int)sd.textures.size();
    S.material_maps = sd.material_maps.empty() ? nullptr : (const rt_texmap *)D->material_maps.p;
    S.env_map = sd.env_map; S.bg_map = sd.bg_map;
    S.use_uvw = sd.material_maps.empty() ? 0 : 1;
    S.max_bvh_depth = max_depth;
    S.bvh_spill = nullptr;
    if (max_depth > RT_BVH_LDS) {
        if ((st = D->bvh_spill.ensure(SPILL_BYTES))) return st;
        S.bvh_spill = (uint32_t *)D->bvh_spill.p;
    }
    S.stochastic = 0;
    for (const rt_light &l : sd.lights) if (l.type == RT_LIGHT_POINT && l.size != 0) S.stochastic = 1;
    for (const rt_blinn &m : sd.materials) if (m.reflection_glossiness != 0 || m.refraction_glossiness != 0) S.stochastic = 1;
    D->scene_valid = true;
    return RT_OK;
}

// Gather structure of a photon map (rt_dev.h, built by rt_photon_build.hip on the GPU): the photons LocatePhotons can reach
// -- it descends only while index < halfStoredPhotons = n/2 - 1, cyPhotonMap.h:217,371, so heap slots >= 2*half are never
// visited: for a balanced array that is a prefix, for an unbalanced one everything but the `skip` indices -- re-sorted by
// recursive median splits into 2^D sub-leaves of <= RT_SUB_PHOTONS photons with their tight boxes; RT_LEAF_SUBS consecutive
// sub-leaves are one leaf (128 slots) of the tree the queries walk (boxes of all heap nodes in `box4`: the tree over the
// leaves is its head, the sub-leaf boxes its last level).  `src_dev` != NULL: the photons already sit on this device
// (n_src records, 0-based); else they are uploaded from `src_host`.  skip: ascending 0-based positions to leave out.
static rt_status build_photon_structure(DeviceState *D, bool caustic, const rt_photon *src_dev, const rt_photon *src_host, uint32_t n_src,
                                        const uint32_t *skip, uint32_t n_skip, double *ms_upload, double *ms_build)
{
    using clk = std::chrono::steady_clock;
    DevPhotonMap &pm = caustic ? D->scene.cm : D->scene.pm;
    bool &valid = caustic ? D->caustic_valid : D->photons_valid;
    // nothing is published before the build and its final synchronisation have succeeded: a failure leaves the device WITHOUT a
    // valid map, so the next render tries again (and fails as loudly) instead of rendering without global illumination
    memset(&pm, 0, sizeof pm);
    valid = false;
    struct TempBuf : DevBuf { ~TempBuf() { release(); } };      // released on every path out of this function
    DevBuf &b_pa = caustic ? D->cpa : D->pa, &b_pb = caustic ? D->cpb : D->pb, &b_box = caustic ? D->cbox4 : D->box4, &b_grid = caustic ? D->cgrid : D->grid;
    if (n_src <= n_skip) { valid = true; return RT_OK; }
    if (n_skip > 8) return fail(RT_ERR_LIMIT, "photon structure: %u photons to leave out, the device copy takes at most 8 (rt::UnreachablePhotons yields at most 4)", n_skip);
    const uint32_t n = n_src - n_skip;
    uint32_t n_leaves = 1;
    while ((size_t)n_leaves * RT_LEAF_SUBS * RT_SUB_PHOTONS < n) n_leaves <<= 1;
    // leaf ids travel as 16-bit values through the kernel's LDS lists; slots are addressed by 32-bit byte offsets
    if (n_leaves > 65536) return fail(RT_ERR_LIMIT, "photon map too large for the gather structure (%u photons, at most 8 Mi)", n);
    static_assert((65536ull * RT_LEAF_SUBS + 1) * RT_SUB_PHOTONS * sizeof(float4) <= 0xFFFFFFFFull, "photon slots are addressed by 32-bit byte offsets");
    const uint32_t n_sub = n_leaves * RT_LEAF_SUBS;
    rt_status st;
    hipStream_t stream = D->stream;
    const auto t0 = clk::now();
    TempBuf staged, compacted, scratch;
    const rt_photon *ph = src_dev;
    if (!ph) {
        if ((st = staged.ensure((size_t)n_src * sizeof(rt_photon)))) return st;
        HIP_TRY(hipMemcpyAsync(staged.p, src_host, (size_t)n_src * sizeof(rt_photon), hipMemcpyHostToDevice, stream));
        ph = (const rt_photon *)staged.p;
    }
    if (n_skip) {
        if ((st = compacted.ensure((size_t)n * sizeof(rt_photon)))) return st;
        rtk_photon_copy_skipping(stream, ph, n_src, skip, n_skip, (rt_photon *)compacted.p);
        ph = (const rt_photon *)compacted.p;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    const auto t1 = clk::now();
    // one more sub-leaf than the tree has, every slot empty: the kernel pads its sub-leaf lists with it
    const size_t slots = ((size_t)n_sub + 1) * RT_SUB_PHOTONS;
    const size_t sbytes = rtk_photon_structure_scratch(n, n_sub);
    if ((st = b_pa.ensure(slots * sizeof(float4))) || (st = b_pb.ensure(slots * sizeof(float4))) || (st = b_box.ensure((size_t)4 * n_sub * sizeof(float4))) ||
        (st = b_grid.ensure((size_t)64 * 64 * 64 * 4)) || (st = scratch.ensure(sbytes))) return st;
    PhotonGridOut g;
    const hipError_t e = rtk_photon_structure(stream, ph, n, n_sub, (float4 *)b_pa.p, (float4 *)b_pb.p, (float4 *)b_box.p, (uint32_t *)b_grid.p, &g, scratch.p, sbytes);
    hipError_t e2 = hipStreamSynchronize(stream);
    if (e != hipSuccess || e2 != hipSuccess) return fail(RT_ERR_DEVICE, "photon structure build failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    DevPhotonMap built;
    memset(&built, 0, sizeof built);
    built.pa = (const float4 *)b_pa.p; built.pb = (const float4 *)b_pb.p;
    built.tbox = (const float4 *)b_box.p; built.sbox = (const float4 *)b_box.p + 2 * (size_t)n_sub;
    built.n_leaves = n_leaves; built.n_photons = n;
    built.grid = (const uint32_t *)b_grid.p;
    for (int a = 0; a < 3; a++) { built.grid_min[a] = g.min[a]; built.grid_dim[a] = g.dim[a]; }
    built.cell = g.cell; built.inv_cell = 1.0f / g.cell;
    {
        DevBuf &b_rk = caustic ? D->ccell_rk2 : D->cell_rk2;
        if ((st = b_rk.ensure((size_t)64 * 64 * 64 * 4))) return st;
        HIP_TRY(hipMemsetAsync(b_rk.p, 0, (size_t)64 * 64 * 64 * 4, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    pm = built;
    valid = true;
    (caustic ? D->rk2_k_c : D->rk2_k) = 0;          // the hint table is empty: no gather parameters recorded yet
    const auto t2 = clk::now();
    if (ms_upload) *ms_upload += std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (ms_build) *ms_build += std::chrono::duration<double, std::milli>(t2 - t1).count();
    return RT_OK;
}

// the generated photons on the host (caller holds s->mu): copied from the device that made them, once
static rt_status raw_to_host(rt_scene *s)
{
    if (!s->gen_n || !s->photons_raw.empty()) return RT_OK;
    if (!s->gen_dev || !s->gen_dev->raw_photons.p) return fail(RT_ERR_STATE, "the generated photon map is gone from its device");
    int cur = 0;
    (void)hipGetDevice(&cur);
    HIP_TRY(hipSetDevice(s->gen_dev->device));
    std::vector<rt_photon> raw((size_t)s->gen_n + 1);
    HIP_TRY(hipMemcpy(raw.data(), s->gen_dev->raw_photons.p, raw.size() * sizeof(rt_photon), hipMemcpyDeviceToHost));
    HIP_TRY(hipSetDevice(cur));
    s->photons_raw.swap(raw);
    return RT_OK;
}

static rt_status upload_photons(rt_scene *s, DeviceState *D, bool caustic)
{
    if (!caustic && s->gen_n) {
        std::vector<uint32_t> skip0;
        for (uint32_t i : s->photons_skip) skip0.push_back(i - 1);              // 1-based raw index -> position in [1..n]
        if (D == s->gen_dev && D->raw_photons.p)                                 // still on this device
            return build_photon_structure(D, false, (const rt_photon *)D->raw_photons.p + 1, nullptr, s->gen_n, skip0.data(), (uint32_t)skip0.size(), nullptr, nullptr);
        rt_status st = raw_to_host(s);                                           // another device: through the host
        if (st) return st;
        HIP_TRY(hipSetDevice(D->device));
    }
    if (!caustic && !s->photons_raw.empty()) {
        // as generated (unbalanced): everything but the photons balancing would put out of LocatePhotons' reach
        std::vector<uint32_t> skip0;
        for (uint32_t i : s->photons_skip) skip0.push_back(i - 1);              // 1-based raw index -> position in [1..n]
        return build_photon_structure(D, false, nullptr, s->photons_raw.data() + 1, (uint32_t)s->photons_raw.size() - 1, skip0.data(), (uint32_t)skip0.size(), nullptr, nullptr);
    }
    const std::vector<rt_photon> &ph = caustic ? s->data.caustic_photons : s->data.photons;
    if (ph.size() < 2) { memset(caustic ? &D->scene.cm : &D->scene.pm, 0, sizeof(DevPhotonMap)); (caustic ? D->caustic_valid : D->photons_valid) = true; return RT_OK; }
    const uint32_t n = (uint32_t)ph.size() - 1;
    return build_photon_structure(D, caustic, nullptr, ph.data() + 1, rt::ReachablePhotonSlots(n), nullptr, 0, nullptr, nullptr);
}

static rt_status prepare_device(rt_scene *s, int device, DeviceState **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(RT_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU path)"); }
    if (device < 0 || device >= n) return fail(RT_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    if (!device_is_gfx950(device)) return fail(RT_ERR_NO_DEVICE, "device %d is not gfx950 (MI355X); kernels are built for gfx950 only", device);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceState *D = device_state(s, device);
    if (!D->stream) HIP_TRY(hipStreamCreateWithFlags(&D->stream, hipStreamNonBlocking));
    rt_status st;
    // an asynchronous render (sync == 0) may still be reading the buffers an upload is about to overwrite (scene tables and
    // photon structure are re-used in place when the new data fits): wait for it on the host first
    if ((!D->scene_valid || !D->photons_valid || !D->caustic_valid) && D->last_pending && D->last_done) HIP_TRY(hipEventSynchronize(D->last_done));
    if (!D->scene_valid) { const DevPhotonMap keep = D->scene.pm, keepc = D->scene.cm; if ((st = upload_scene(s, D))) return st; D->scene.pm = keep; D->scene.cm = keepc; }
    if (!D->photons_valid) if ((st = upload_photons(s, D, false))) return st;
    if (!D->caustic_valid) if ((st = upload_photons(s, D, true))) return st;
    *out = D;
    return RT_OK;
}

// ---- render orchestration ---------------------------------------------------------------------------------
// Samples per pipeline pass.  The kernels are persistent grids that take their work from device-side counters, so a
// bigger chunk means fewer launches, fewer drain phases and longer-lived LDS ray stacks: measured on MI355X (Cornell
// 1080p x 64 spp, two chunks in flight) 4 Mi: 77.4 ms, 8 Mi: 72.3, 32 Mi: 68.9, 64 Mi: 67.7 per frame.  A device-side
// render (rt_render_tiles_device: bench, multi-GPU) therefore takes 64 Mi samples at a time -- 1.1 GB of per-sample
// buffers; the queues are sized from measurement, not from 2^bounce -- while a job (rt_render_begin) keeps 8 Mi chunks:
// its caller watches the image fill band by band, and a band is a chunk.
static size_t chunk_samples_limit(bool job)
{
    const char *e = getenv("RT_CHUNK_SAMPLES");
    long long v = e ? atoll(e) : 0;
    if (v < 4096) v = job ? (8LL << 20) : (64LL << 20);
    return (size_t)v;
}

// Chunks of a frame in flight at once (each on its own stream with its own working set); RT_STREAMS=n in the
// environment (1..RT_STREAMS) overrides the default (tuning / debugging).
static int render_streams(uint64_t n_chunks)
{
    const char *e = getenv("RT_STREAMS");
    // measured on MI355X, round 2.  Many small chunks (Cornell frame in 16 chunks of 8 Mi samples): 1: 74.9 ms, 2: 71.8,
    // 3: 73.5, 4: 71.7 -- a second chunk fills the first one's launch gaps and drain phases, three slots leave the
    // sixteenth chunk alone at the end, two need the least memory.  One or two whole-frame chunks (64 Mi samples): the
    // kernels are persistent grids that fill the GPU on their own and the frame is the sum of them -- a second chunk
    // in flight only makes them share it: 1: 50.5 ms, 2: 51.2.  Round 4, with kernels a third faster: the drain phase of a
    // persistent launch (its last workgroups, its overflow pass) is now worth filling with the other chunk -- Cornell frame
    // 36.15 -> 35.64 ms, the 102 k-triangle frame 21.81 -> 20.66; four chunks of 32 Mi samples stay slower (37.2).
    const int v = e ? atoi(e) : (n_chunks <= 1 ? 1 : 2);
    return v < 1 ? 1 : (v > RT_STREAMS ? RT_STREAMS : v);
}

// ray_factor / query_factor: queue entries per sample to provide (<= 0: the worst case, every hit spawning `fan` rays level
// after level); an overflow is detected on the device and reported, never silent
static rt_status ensure_workspace(DeviceState *D, int slot, size_t samples, int bounce, size_t list_pixels, int fan = 2, bool caustic = false,
                                  double ray_factor = 0, double query_factor = 0)
{
    rt_status st;
    Workspace &w = D->ws[slot];
    if (bounce < 0) bounce = 0;
    if (bounce > 12) return fail(RT_ERR_LIMIT, "bounce limit %d > 12", bounce);
    // worst case: every hit spawns `fan` rays, level after level (overflow is detected and reported)
    double grow = 1.0;
    for (int b = 0; b < bounce; b++) grow *= fan;
    unsigned long long rq_cap = (unsigned long long)std::min(4.0e9, (double)samples * grow);
    unsigned long long pq_cap = (unsigned long long)std::min(4.0e9, (double)samples * grow * 2.0);    // hits on levels 1..bounce
    if (ray_factor > 0) rq_cap = std::min<unsigned long long>(rq_cap, (unsigned long long)((double)samples * ray_factor) + 65536ull);
    if (query_factor > 0) pq_cap = std::min<unsigned long long>(pq_cap, (unsigned long long)((double)samples * query_factor) + 65536ull);
    const unsigned long long lim = 1ull << 28;
    if (rq_cap > lim) rq_cap = lim;
    if (pq_cap > lim) pq_cap = lim;
    if (const char *e = getenv("RT_QUEUE_CAP")) {          // tests only: force a small queue to exercise the overflow report
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 64) { rq_cap = std::min(rq_cap, v); pq_cap = std::min(pq_cap, v); }
    }
    if (rq_cap < 64) rq_cap = 64;
    if (pq_cap < 64) pq_cap = 64;
    if ((st = w.sample_rgb.ensure(samples * 12))) return st;
    if ((st = w.sample_z.ensure(samples * 4))) return st;
    if ((st = w.sample_hit.ensure(samples))) return st;
    for (int i = 0; i < 2; i++) for (int k = 0; k < 5; k++) if ((st = w.rq[i][k].ensure((size_t)rq_cap * 16))) return st;
    for (int k = 0; k < 3; k++) if ((st = w.pq[k].ensure((size_t)pq_cap * 16))) return st;
    if (caustic) for (int k = 0; k < 3; k++) if ((st = w.cq[k].ensure((size_t)pq_cap * 16))) return st;
    if ((st = w.counts.ensure(CNT_TOTAL * 4))) return st;
    if (D->scene.max_bvh_depth > RT_BVH_LDS && (st = w.bvh_spill.ensure(SPILL_BYTES))) return st;
    if ((st = w.pixel_list.ensure(std::max<size_t>(list_pixels, 1) * 4))) return st;
    if (!D->stats.p) { if ((st = D->stats.ensure(STATS_BYTES))) return st; }
    if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));      // (slot 0 uses its own only in a pipelined frame)
    w.samples = samples; w.rq_cap = (uint32_t)rq_cap; w.pq_cap = (uint32_t)pq_cap;
    return RT_OK;
}

static DevWork make_work(DeviceState *D, int slot)
{
    const Workspace &w = D->ws[slot];
    DevWork W;
    W.sample_rgb = (float *)w.sample_rgb.p; W.sample_z = (float *)w.sample_z.p; W.sample_hit = (uint8_t *)w.sample_hit.p;
    for (int i = 0; i < 2; i++) {
        W.rq[i].a = (float4 *)w.rq[i][0].p; W.rq[i].b = (float4 *)w.rq[i][1].p; W.rq[i].c = (float4 *)w.rq[i][2].p;
        W.rq[i].d = (uint4 *)w.rq[i][3].p; W.rq[i].e = (float4 *)w.rq[i][4].p; W.rq[i].cap = w.rq_cap;
    }
    W.pq.qa = (float4 *)w.pq[0].p; W.pq.qb = (float4 *)w.pq[1].p; W.pq.qc = (float4 *)w.pq[2].p; W.pq.cap = w.pq_cap;
    W.cq.qa = (float4 *)w.cq[0].p; W.cq.qb = (float4 *)w.cq[1].p; W.cq.qc = (float4 *)w.cq[2].p; W.cq.cap = w.cq[0].p ? w.pq_cap : 0;
    W.counts = (uint32_t *)w.counts.p; W.pixel_list = (uint32_t *)w.pixel_list.p;
    W.bvh_spill = D->scene.max_bvh_depth > RT_BVH_LDS ? (uint32_t *)w.bvh_spill.p : nullptr;
    W.stats = (unsigned long long *)D->stats.p;
    return W;
}

// Work queued by an earlier asynchronous render still owns the working sets: order `st` behind it.
static rt_status order_after_pending(DeviceState *D, hipStream_t st)
{
    if (D->last_pending && D->last_done) HIP_TRY(hipStreamWaitEvent(st, D->last_done, 0));
    return RT_OK;
}
// Dropped rays / photon queries of the renders since the counter was last cleared (host-synchronous read).
// The statistics block keeps its counters RT_STAT_STRIDE apart (rt_dev.h): counters [first, first + n) as a dense array, and back to zero
static hipError_t stats_read(const void *dev, int first, int n, unsigned long long *out)
{
    return hipMemcpy2D(out, 8, (const unsigned long long *)dev + ST_AT(first), (size_t)RT_STAT_STRIDE * 8, 8, (size_t)n, hipMemcpyDeviceToHost);
}
static hipError_t stats_zero(void *dev, int first, int n, hipStream_t st)
{
    if (first == 0 && n == ST_COUNT) return hipMemsetAsync(dev, 0, STATS_BYTES, st);
    for (int i = first; i < first + n; i++)
        if (hipError_t e = hipMemsetAsync((unsigned long long *)dev + ST_AT(i), 0, 8, st)) return e;
    return hipSuccess;
}

static rt_status read_overflow(DeviceState *D, unsigned long long *drops)
{
    *drops = 0;
    if (!D->stats.p) return RT_OK;
    HIP_TRY(stats_read(D->stats.p, ST_QUEUE_OVERFLOW, 1, drops));
    return RT_OK;
}

// camera set-up of RenderPixel, FIN/main.cpp:205-224 (tan in double, everything else float)
static void camera_setup(const rt_camera &cam, DevCamera &dc)
{
    using rt::Point3;
    const float theta = cam.fov;
    const float l = cam.focaldist;
    const float h = (float)(2 * l * tan(theta / 2 * (M_PI / 180)));
    const float w = h * (float)cam.width / cam.height;
    Point3 b(-w / 2, h / 2, -l);
    const float u = w / cam.width;
    const float v = -h / cam.height;
    const float du = u / 2, dv = v / 2;
    b.x += du; b.y += dv;
    Point3 up(cam.up[0], cam.up[1], cam.up[2]);
    Point3 z_new = Point3(cam.dir[0], cam.dir[1], cam.dir[2]) * (float)-1;
    Point3 x_new = up ^ z_new;
    up.Normalize(); z_new.Normalize(); x_new.Normalize();
    memcpy(dc.pos, cam.pos, 12);
    dc.m[0] = x_new.x; dc.m[1] = x_new.y; dc.m[2] = x_new.z;
    dc.m[3] = up.x; dc.m[4] = up.y; dc.m[5] = up.z;
    dc.m[6] = z_new.x; dc.m[7] = z_new.y; dc.m[8] = z_new.z;
    dc.b[0] = b.x; dc.b[1] = b.y; dc.b[2] = b.z;
    dc.u = u; dc.v = v; dc.dof = cam.dof; dc.width = cam.width; dc.height = cam.height;
}

static rt_status validate_render(const rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *t)
{
    if (!cam || !p || !t) return fail(RT_ERR_ARG, "render: camera, params and tile range are required");
    if (cam->width <= 0 || cam->height <= 0 || (long long)cam->width * cam->height > (1LL << 28)) return fail(RT_ERR_ARG, "render: bad image size %dx%d", cam->width, cam->height);
    if ((unsigned long long)cam->width * cam->height * (unsigned long long)p->max_sample >= (1ull << 32))
        return fail(RT_ERR_LIMIT, "render: width*height*max_sample must stay below 2^32 (sample ids are 32-bit)");
    if (p->min_sample < 1 || p->max_sample < p->min_sample || p->max_sample > 4096) return fail(RT_ERR_ARG, "render: need 1 <= min_sample <= max_sample <= 4096");
    if (p->shade_model < RT_SHADE_FIN || p->shade_model > RT_SHADE_P3)
        return fail(RT_ERR_ARG, "render: unknown shade model %d", p->shade_model);
    if (p->shade_model == RT_SHADE_P6 && p->bounce > 7) return fail(RT_ERR_LIMIT, "render: RT_SHADE_P6 supports bounce <= 7");
    if (p->shade_model == RT_SHADE_P12 && (p->hemisphere_sample < 1 || p->hemisphere_sample > 256))
        return fail(RT_ERR_ARG, "render: hemisphere_sample must be 1..256 for RT_SHADE_P12");
    if (p->knn_k < 1 || p->knn_k > 65536 || !(p->knn_radius > 0)) return fail(RT_ERR_ARG, "render: bad photon gather parameters");
    if (!(p->gamma > 0)) return fail(RT_ERR_ARG, "render: gamma must be positive");
    if (p->caustic_k < 0 || p->caustic_k > 65536 || (p->caustic_k > 0 && !(p->caustic_radius > 0))) return fail(RT_ERR_ARG, "render: bad caustic gather parameters");
    if (t->tile_w <= 0 || t->tile_h <= 0 || t->stride <= 0 || t->first < 0) return fail(RT_ERR_ARG, "render: bad tile range");
    if (p->shadow_samples > 32) return fail(RT_ERR_LIMIT, "render: at most 32 shadow samples per light");
    if (p->photon_count < 0 || p->photon_count > (1 << 28) || (p->photon_count > 0 && (p->photon_bounce < 1 || p->photon_bounce > 8)))
        return fail(RT_ERR_ARG, "render: need 0 <= photon_count <= 2^28 and 1 <= photon_bounce <= 8");
    return RT_OK;
}

// DevPhotonMap::cell_start for the radius the gather is about to use: built on first use and again when a larger radius comes
// (a table built for a radius serves every smaller one), on the stream the gather will run on
static rt_status ensure_cell_start(DeviceState *D, bool caustic, int k, float radius, hipStream_t st, bool *did_work = nullptr)
{
    DevPhotonMap &pm = caustic ? D->scene.cm : D->scene.pm;
    {
        // the per-cell k-th distances (DeviceState::cell_rk2) are hints of ONE gather configuration: what queries with another k
        // or radius left there is dropped, so that it cannot steer this render's choice between the (exact) gather paths
        int &hk = caustic ? D->rk2_k_c : D->rk2_k; float &hr = caustic ? D->rk2_radius_c : D->rk2_radius;
        DevBuf &b_rk = caustic ? D->ccell_rk2 : D->cell_rk2;
        if (b_rk.p && hk != 0 && (hk != k || hr != radius)) { HIP_TRY(hipMemsetAsync(b_rk.p, 0, (size_t)64 * 64 * 64 * 4, st)); if (did_work) *did_work = true; }
        hk = k; hr = radius;
    }
    if (pm.n_leaves < 2 || (pm.cell_start && radius <= pm.start_radius)) return RT_OK;
    DevBuf &b = caustic ? D->ccell_start : D->cell_start;
    rt_status s = b.ensure((size_t)64 * 64 * 64 * 4);
    if (s) return s;
    rtk_photon_cell_start(st, pm.tbox, pm.n_leaves, pm.grid_min, pm.cell, pm.grid_dim, radius, (uint32_t *)b.p);
    HIP_TRY(hipGetLastError());
    if (did_work) *did_work = true;
    pm.cell_start = (const uint32_t *)b.p; pm.start_radius = radius;
    return RT_OK;
}

struct Timing { std::vector<hipEvent_t> ev; std::vector<int> cls; };

static rt_status run_pipeline(DeviceState *D, int slot, hipStream_t st, const DevWork &W, const rt_params &P, Timing *tm,
                              const DevCamera &dc, const DevTiles &dt, uint32_t q0, uint32_t npix, int j0, int ns,
                              int max_sample, int mode, const float *rays_dev)
{
    const int max_blocks = 256 * 5;
    static_assert(256 * 5 <= RT_SPILL_BLOCKS, "DevScene::bvh_spill is sized for RT_SPILL_BLOCKS workgroups");
    auto mark = [&](int cls) -> rt_status {
        if (!tm) return RT_OK;
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(e, st));
        tm->ev.push_back(e); tm->cls.push_back(cls);
        return RT_OK;
    };
    rt_status s;
    // queue counters for levels 0..15 and the photon queue are reset; the pixel list count survives
    HIP_TRY(hipMemsetAsync(W.counts, 0, CNT_RESET * 4, st));
    if ((s = mark(-1))) return s;
    rtk_launch_primary(st, D->scene, W, P, W.rq[1], W.counts + 1, dc, dt, q0, npix, j0, ns, max_sample, mode, rays_dev, max_blocks);
    if ((s = mark(0))) return s;
    // P6: a side ray is spawned when its refraction ray ARRIVES, one queue level later than a sibling
    // would be, so a path can take up to two levels per bounce
    const int max_level = P.shade_model == RT_SHADE_P6 ? 2 * P.bounce : P.bounce;
    // k_wavefront's overflow (level-1 queue) first goes through a second k_wavefront pass; what overflows again, and the
    // models without that kernel, take one launch per level
    int first_level = 1;
    if (max_level >= 2 && rtk_launch_wavefront_queue(st, D->scene, W, P, W.rq[1], W.counts + 1, W.rq[0], W.counts + 2, dc, dt, q0, max_sample, mode)) first_level = 2;
    // (further passes of the same kernel over what the second one could not keep were measured: 1 / 2 / 3 / 4 queue passes, tracer ms behind
    // the first pass, Cornell 0.34 / 0.36 / 0.40 / 0.44, C3 13.0 / 13.0 / 13.1 / 13.1 -- the second pass takes everything that matters)
    for (int level = first_level; level <= max_level && level < 15; level++)
        rtk_launch_bounce(st, D->scene, W, P, W.rq[level & 1], W.rq[(level + 1) & 1], W.counts + level + 1, level, max_blocks);
    if ((s = mark(1))) return s;
    rt_status cs;
    if ((cs = ensure_cell_start(D, false, P.knn_k, P.knn_radius, st))) return cs;
    if (D->scene.pm.n_leaves) {
        rtk_launch_gather(st, D->scene.pm, W.pq.qa, W.pq.qb, W.pq.qc, W.counts + CNT_PHOTONQ, W.pq.cap, P.knn_k, P.knn_radius,
                          W.sample_rgb, nullptr, nullptr, 0, W.stats, GATHER_BLOCKS, W.counts + CNT_GATHER_NEXT, (float *)D->cell_rk2.p);
        if ((s = mark(2))) return s;
    }
    if (D->scene.cm.n_leaves && P.caustic_k > 0 && W.cq.cap && (cs = ensure_cell_start(D, true, P.caustic_k, P.caustic_radius, st))) return cs;
    if (D->scene.cm.n_leaves && P.caustic_k > 0 && W.cq.cap) {
        // the P13-family models queued their caustic lookups separately: same kernel on the second map
        rtk_launch_gather(st, D->scene.cm, W.cq.qa, W.cq.qb, W.cq.qc, W.counts + CNT_CAUSTICQ, W.cq.cap, P.caustic_k, P.caustic_radius,
                          W.sample_rgb, nullptr, nullptr, 0, W.stats, GATHER_BLOCKS, W.counts + CNT_GATHER_NEXT2, (float *)D->ccell_rk2.p);
        if ((s = mark(2))) return s;
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

#define RT_ERR_OVERFLOW_RETRY (-1000)     /* internal: queues sized from history overflowed; render again with worst-case queues */
static rt_status render_tiles_once(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles, int device,
                                   hipStream_t user_stream, bool use_user_stream, uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev,
                                   bool sync, rt_stats *stats_out, rt_job *job, void *packed_dev, bool worst_case);

// Queue sizing policy: the first render of a kind (shading model, bounce limit, fan-out, maps in use) provides one ray and
// half a photon query per sample, later ones twice what the fullest chunk so far needed (at least 1 ray and 0.25 queries per sample);
// if that ever overflows, a synchronous render
// is repeated once with the worst-case size (2^bounce per sample) -- an asynchronous one cannot be repeated by the
// library: it starts from the worst case unless there is history, and an overflow is reported by rt_render_check.
static rt_status render_tiles(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles, int device,
                              hipStream_t user_stream, bool use_user_stream, uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev,
                              bool sync, rt_stats *stats_out, rt_job *job, void *packed_dev = nullptr)
{
    rt_status st = render_tiles_once(s, cam, p, tiles, device, user_stream, use_user_stream, rgb8_dev, z_dev, count_dev, sync, stats_out, job,
                                     packed_dev, false);
    int attempts = 1;
    if (st == RT_ERR_OVERFLOW_RETRY) {
        attempts = 2;
        st = render_tiles_once(s, cam, p, tiles, device, user_stream, use_user_stream, rgb8_dev, z_dev, count_dev, sync, stats_out, job,
                               packed_dev, true);
    }
    if (st == RT_OK && stats_out) stats_out->attempts = (uint64_t)attempts;
    if (st == RT_OK && job) job->stats.attempts = (uint64_t)attempts;
    return st;
}

static rt_status render_tiles_once(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles, int device,
                                   hipStream_t user_stream, bool use_user_stream, uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev,
                                   bool sync, rt_stats *stats_out, rt_job *job, void *packed_dev, bool worst_case)
{
    rt_status st = validate_render(s, cam, p, tiles);
    if (st) return st;
    if (!packed_dev && (!rgb8_dev || !z_dev || !count_dev)) return fail(RT_ERR_ARG, "render: output buffers are required");
    DeviceState *D = nullptr;
    if ((st = prepare_device(s, device, &D))) return st;
    // one host call at a time per (scene, device): the working sets are not shared between concurrent calls
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "render: another call on this scene is using device %d", device);
    hipStream_t stream = use_user_stream ? user_stream : D->stream;
    if ((st = order_after_pending(D, stream))) return st;
    if (!D->last_done) HIP_TRY(hipEventCreateWithFlags(&D->last_done, hipEventDisableTiming));
    if (D->last_pending && (sync || stats_out != nullptr || job != nullptr)) {
        // this call will clear and read the shared drop counter: the verdict of the asynchronous renders before it is
        // collected first and kept for rt_render_check ("queue overflow is always reported")
        HIP_TRY(hipEventSynchronize(D->last_done));
        unsigned long long drops = 0;
        if ((st = read_overflow(D, &drops))) return st;
        if (drops) { D->async_overflow = true; D->qhist.valid = false; HIP_TRY(hipMemset((unsigned long long *)D->stats.p + ST_AT(ST_QUEUE_OVERFLOW), 0, 8)); }
        D->last_pending = false;
    }

    DevCamera dc;
    camera_setup(*cam, dc);
    DevTiles dt;
    dt.tile_w = tiles->tile_w; dt.tile_h = tiles->tile_h; dt.first = tiles->first; dt.stride = tiles->stride;
    dt.tiles_x = (cam->width + dt.tile_w - 1) / dt.tile_w;
    const int tiles_y = (cam->height + dt.tile_h - 1) / dt.tile_h;
    dt.tiles_total = dt.tiles_x * tiles_y;
    dt.n_tiles = dt.first >= dt.tiles_total ? 0 : (dt.tiles_total - dt.first + dt.stride - 1) / dt.stride;
    const uint64_t tile_px = (uint64_t)dt.tile_w * dt.tile_h;
    const uint64_t total_px = tile_px * (uint64_t)dt.n_tiles;

    const size_t limit = chunk_samples_limit(job != nullptr);
    uint64_t ppc = std::max<uint64_t>(1, limit / (uint64_t)p->max_sample);
    ppc = std::max<uint64_t>(tile_px, ppc / tile_px * tile_px);
    ppc = std::min<uint64_t>(ppc, std::max<uint64_t>(total_px, 1));
    // P12: a diffuse hit spawns hemisphere rays on top of the reflection/refraction pair; in practice one
    // of the three classes dominates per material, so the queues are sized for a fan-out of 2 and an
    // overflow is reported as an error rather than silently dropped
    const uint64_t n_chunks = total_px ? (total_px + ppc - 1) / ppc : 0;
    // FRAME PIPELINING (opt-in: RT_FRAME_PIPELINE=1).  An asynchronous render behind another one that is still in flight runs all of its
    // slots on the library's own streams and lets its tracing and gathering start at once: they touch nothing but the slot's working
    // set, which the slot's stream already orders, and read scene tables no stream-ordered work of the caller can change.  Only
    // k_resolve, which writes the caller's buffers, waits for what the caller's stream holds before this call (e_fork) -- the frame
    // before it, an all-gather of its tiles, a copy of the image; frames of one chunk alternate between two slots.  What it is for:
    // the tile exchange of a multi-GPU step sits between two frames, and this puts the next frame's rays beside it.  Why it is not the
    // default: measured on ONE GPU (r4, profiles/r04_experiments.json) it gains nothing where there is no exchange to hide -- a rank's
    // share at N = 2 / 4 / 8: 18.73 / 10.14 / 5.39 ms against 18.73 / 10.14 / 5.33 -- and costs the two-chunk frames their lockstep
    // (both chunks tracing, then both gathering: Cornell 36.4 against 36.0 ms, 102 k triangles 21.75 against 20.64: a tracer and a gather
    // side by side take each other's LDS); with an exchange beside persistent grids that leave no LDS free it is unmeasured.
    bool pipelined = false;
    {
        const char *e = getenv("RT_FRAME_PIPELINE");
        pipelined = e && atoi(e) != 0 && !sync && stats_out == nullptr && job == nullptr && D->last_pending && total_px > 0;
    }
    const int streams_wanted = render_streams(pipelined ? std::max<uint64_t>(n_chunks, 2) : n_chunks);
    if (streams_wanted < 2) pipelined = false;              // RT_STREAMS=1: one chunk at a time, as asked
    int n_slots = pipelined ? streams_wanted : (int)std::min<uint64_t>(std::max<uint64_t>(n_chunks, 1), (uint64_t)streams_wanted);
    const int fan = p->shade_model == RT_SHADE_P12 && p->hemisphere_sample > 1 ? 1 + p->hemisphere_sample : 2;
    const bool use_photons = D->scene.pm.n_leaves != 0, use_caustic = p->caustic_k > 0 && D->scene.cm.n_leaves != 0;
    const DeviceState::QueueHistory &H = D->qhist;
    const bool hist_ok = H.valid && H.model == p->shade_model && H.bounce == p->bounce && H.fan == fan && H.photons == use_photons && H.caustic == use_caustic;
    double ray_factor = 0, query_factor = 0;           // 0 = worst case
    if (!worst_case && getenv("RT_QUEUE_WORST_CASE") == nullptr) {
        // floors: how many rays k_wavefront cannot keep in LDS depends on timing, and a view change can bring glass into a
        // frame that had none -- one ray (160 B of queue) and a quarter of a query (12 B) per sample: 1.4 GB per 8 Mi-sample working
        // set of a job, 11.5 GB per 64 Mi-sample one of a device-side render -- and cover both
        if (hist_ok) { ray_factor = std::max(2.0 * H.rays_per_sample, 1.0); query_factor = std::max(2.0 * H.queries_per_sample, 0.25); }
        else if (sync || job) {
            // first render of a kind: from the model's fan-out.  k_wavefront keeps the ray tree in LDS (the global queue only sees
            // what does not fit).  The per-level models put a whole level into the queue: about one hemisphere ray per path that
            // is still alive (P12: hemisphere_sample of them on the first level, RayTracingProj12 main.cpp:393-446) PLUS the
            // ungated reflection / refraction pairs of the P13-family Shade, which double level by level inside glass
            // (P13/main.cpp:633-751: measured 2.5 rays per sample on the fullest level of a Cornell chunk that holds the glass
            // sphere, bounce 8) -- four per sample on top of the hemisphere rays, 640 B per sample of queue memory
            // (the kernels' own predicate: a scene k_wavefront does not take -- a BVH beyond its traversal stack, RT_TRACER=levels -- goes
            // through the per-level kernels and gets their figure; P12 through k_wavefront keeps the per-level figure too: its overflow is scene-dependent)
            const bool wf = (p->shade_model == RT_SHADE_FIN || p->shade_model == RT_SHADE_P13) && rtk_wavefront_usable(D->scene, *p);
            ray_factor = wf ? 1.0 : (p->shade_model == RT_SHADE_P12 ? (double)std::max(p->hemisphere_sample, 1) + 3.0 : 4.0);
            query_factor = 0.5;
        }
    }
    D->qhist_used = ray_factor > 0 || query_factor > 0;
    DevWork Ws[RT_STREAMS];
    int n_ready = 0;
    for (int i = 0; i < n_slots; i++) {
        if (i > 0 && D->ws[i].samples < (size_t)ppc * p->max_sample) {
            // an extra working set is an optimisation: take it only while a comfortable share of the HBM stays
            // free (other ranks rehearsing on the same device, the caller's own tensors)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
            const size_t first = D->ws[0].sample_rgb.bytes + D->ws[0].sample_z.bytes + D->ws[0].sample_hit.bytes + 10 * D->ws[0].rq[0][0].bytes +
                                 3 * D->ws[0].pq[0].bytes;
            if (free_b < first + first / 4 + (total_b >> 3)) break;
        }
        if ((st = ensure_workspace(D, i, (size_t)ppc * p->max_sample, p->bounce, (size_t)ppc, fan, use_caustic, ray_factor, query_factor))) return st;
        Ws[i] = make_work(D, i);
        n_ready++;
    }
    n_slots = n_ready;
    // a job that owns only some of the tiles renders into packed records of its own (see finish_oldest)
    const bool job_packed = job != nullptr && tiles->stride != 1;
    DevBuf job_packed_buf;
    struct Release { DevBuf &b; ~Release() { b.release(); } } job_packed_release{job_packed_buf};
    if (job_packed) {
        if ((st = job_packed_buf.ensure(std::max<uint64_t>(total_px, 1) * 8))) return st;
        packed_dev = job_packed_buf.p;
    }
    const bool want_stats = stats_out != nullptr || job != nullptr;
    Timing tm[RT_STREAMS];
    hipEvent_t e_begin = nullptr, e_end = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> resolve_ev;
    struct InFlight { uint64_t q0; uint32_t npix; hipEvent_t done; };
    std::vector<InFlight> flight;
    // every timing / completion event of this attempt is destroyed when the function is left, on whichever path
    struct EventGuard {
        Timing *tm; hipEvent_t &b, &e; std::vector<std::pair<hipEvent_t, hipEvent_t>> &rv; std::vector<InFlight> &fl;
        ~EventGuard()
        {
            for (int sl = 0; sl < RT_STREAMS; sl++) for (hipEvent_t x : tm[sl].ev) (void)hipEventDestroy(x);
            if (b) (void)hipEventDestroy(b);
            if (e) (void)hipEventDestroy(e);
            for (auto &pr : rv) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
            for (InFlight &f : fl) (void)hipEventDestroy(f.done);
        }
    } event_guard{tm, e_begin, e_end, resolve_ev, flight};
    if (want_stats) {
        HIP_TRY(stats_zero(Ws[0].stats, 0, ST_COUNT, stream));
        HIP_TRY(hipEventCreate(&e_begin)); HIP_TRY(hipEventCreate(&e_end));
        HIP_TRY(hipEventRecord(e_begin, stream));
    } else if (!D->last_pending) {
        // the drop counter is read back after every synchronous render (below) and by rt_render_check after
        // asynchronous ones; consecutive asynchronous renders accumulate into it until it is checked
        HIP_TRY(stats_zero(Ws[0].stats, ST_QUEUE_OVERFLOW, 1, stream));
        HIP_TRY(stats_zero(Ws[0].stats, ST_PEAK_RAYS, 2, stream));
    }
    // the gathers' start tables are built on `stream` before the slots fork from it
    bool tables_touched = false;
    if ((st = ensure_cell_start(D, false, p->knn_k, p->knn_radius, stream, &tables_touched))) return st;
    if (use_caustic && (st = ensure_cell_start(D, true, p->caustic_k, p->caustic_radius, stream, &tables_touched))) return st;
    // slot 0 runs on `stream` itself; the other slots' streams start after everything already queued on
    // `stream` (fork) and `stream` waits for them at the end (join), so the call keeps stream-order semantics
    auto slot_stream = [&](int slot) { return (slot == 0 && !pipelined) ? stream : D->ws[slot].stream; };
    hipEvent_t e_fork = nullptr;
    struct ForkGuard { hipEvent_t &e; ~ForkGuard() { if (e) (void)hipEventDestroy(e); } } fork_guard{e_fork};
    bool wait_at_start[RT_STREAMS];                      // the slot's stream has not been put behind e_fork yet
    for (int i = 0; i < RT_STREAMS; i++) wait_at_start[i] = false;
    bool resolve_waits[RT_STREAMS];
    for (int i = 0; i < RT_STREAMS; i++) resolve_waits[i] = false;
    if (n_slots > 1 || pipelined) {
        HIP_TRY(hipEventCreateWithFlags(&e_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(e_fork, stream));
        for (int i = 0; i < n_slots; i++) {
            if (i == 0 && !pipelined) continue;          // slot 0 IS `stream`
            // pipelined: the gathers' tables were just rewritten on `stream`, or the frame before used the slots on other streams
            // (slot 0 on the caller's): start behind everything; otherwise only this slot's k_resolve launches wait
            if (!pipelined || tables_touched || !D->last_pipelined) wait_at_start[i] = true;
            else resolve_waits[i] = true;
        }
        for (int i = 0; i < n_slots; i++) if (wait_at_start[i]) HIP_TRY(hipStreamWaitEvent(slot_stream(i), e_fork, 0));
    }
    const float inv_gamma = (float)(1.0 / p->gamma);        // powf(x, 1.0/gamma): double quotient narrowed to float
    double ms_resolve = 0;
    // chunks in flight (job mode): finished oldest-first for progress and the band copy
    int attempt_progress = 0;
    auto finish_oldest = [&]() -> rt_status {
        const InFlight f = flight.front();
        flight.erase(flight.begin());
        HIP_TRY(hipEventSynchronize(f.done));
        (void)hipEventDestroy(f.done);
        // pixels of this chunk that lie inside the image
        int done = 0;
        for (uint64_t q = f.q0; q < f.q0 + f.npix; q += tile_px) {
            const int t = dt.first + (int)(q / tile_px) * dt.stride;
            const int tx = t % dt.tiles_x, ty = t / dt.tiles_x;
            const int w = std::min(dt.tile_w, cam->width - tx * dt.tile_w), h = std::min(dt.tile_h, cam->height - ty * dt.tile_h);
            if (w > 0 && h > 0) done += w * h;
        }
        if (job->host_rgb && job_packed) {
            // a strided tile range (one job per device on the same caller-owned image): rows are shared with other jobs' tiles,
            // so only this job's pixels may be written -- the chunk's packed 8-byte records come back in one copy and are
            // scattered on the host
            std::vector<uint2> rec(f.npix);
            HIP_TRY(hipMemcpy(rec.data(), (const uint2 *)packed_dev + f.q0, (size_t)f.npix * 8, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < f.npix; i++) {
                const uint64_t q = f.q0 + i;
                const int t = dt.first + (int)(q / tile_px) * dt.stride;
                const int w = (int)(q % tile_px);
                const int x = (t % dt.tiles_x) * dt.tile_w + w % dt.tile_w, y = (t / dt.tiles_x) * dt.tile_h + w / dt.tile_w;
                if (x >= cam->width || y >= cam->height) continue;
                const size_t o = (size_t)y * cam->width + x;
                const uint2 v = rec[i];
                job->host_rgb[3 * o] = (uint8_t)(v.x & 255u); job->host_rgb[3 * o + 1] = (uint8_t)((v.x >> 8) & 255u); job->host_rgb[3 * o + 2] = (uint8_t)((v.x >> 16) & 255u);
                const uint32_t zb = (v.x >> 24) | (v.y << 8);
                memcpy(&job->host_z[o], &zb, 4);
                job->host_count[o] = (uint8_t)(v.y >> 24);
            }
        } else if (job->host_rgb) {
            // rows spanned by this chunk's tiles (tile-major order: a contiguous band of tile rows; a row shared
            // with a chunk still in flight may arrive torn and is copied again when that chunk finishes)
            const uint64_t k0 = f.q0 / tile_px, k1 = (f.q0 + f.npix - 1) / tile_px;
            const int ty0 = (dt.first + (int)k0 * dt.stride) / dt.tiles_x, ty1 = (dt.first + (int)k1 * dt.stride) / dt.tiles_x;
            const size_t y0 = (size_t)ty0 * dt.tile_h, y1 = std::min<size_t>((size_t)cam->height, (size_t)(ty1 + 1) * dt.tile_h);
            if (y1 > y0) {
                const size_t o = y0 * cam->width, n = (y1 - y0) * cam->width;
                HIP_TRY(hipMemcpy(job->host_rgb + 3 * o, rgb8_dev + 3 * o, 3 * n, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(job->host_z + o, z_dev + o, 4 * n, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(job->host_count + o, count_dev + o, n, hipMemcpyDeviceToHost));
            }
        }
        // monotone also when the frame is rendered a second time with larger queues (RT_ERR_OVERFLOW_RETRY)
        attempt_progress += done;
        int seen = job->progress.load();
        while (attempt_progress > seen && !job->progress.compare_exchange_weak(seen, attempt_progress)) {}
        return RT_OK;
    };
    uint64_t chunk_index = 0;
    for (uint64_t q0 = 0; q0 < total_px; q0 += ppc, chunk_index++) {
        if (job && job->stop.load()) break;
        const int slot = (int)((chunk_index + (pipelined ? D->frame_seq : 0)) % (uint64_t)n_slots);
        const hipStream_t cs = slot_stream(slot);
        const DevWork &W = Ws[slot];
        Timing *tmp = want_stats ? &tm[slot] : nullptr;
        const uint32_t npix = (uint32_t)std::min<uint64_t>(ppc, total_px - q0);
        HIP_TRY(hipMemsetAsync(W.counts + CNT_PIXLIST, 0, 4, cs));
        if ((st = run_pipeline(D, slot, cs, W, *p, tmp, dc, dt, (uint32_t)q0, npix, 0, p->min_sample, p->max_sample, 0, nullptr))) return st;
        auto timed_resolve = [&](int phase) -> rt_status {
            hipEvent_t r0 = nullptr, r1 = nullptr;
            if (want_stats) { HIP_TRY(hipEventCreate(&r0)); HIP_TRY(hipEventCreate(&r1)); HIP_TRY(hipEventRecord(r0, cs)); }
            rtk_launch_resolve(cs, D->scene, W, dc, dt, (uint32_t)q0, npix, p->min_sample, p->max_sample, p->threshold, inv_gamma, phase,
                               D->scene.bg, rgb8_dev, z_dev, count_dev, packed_dev, 2048);
            if (want_stats) { HIP_TRY(hipEventRecord(r1, cs)); resolve_ev.emplace_back(r0, r1); }
            return RT_OK;
        };
        if (resolve_waits[slot]) { HIP_TRY(hipStreamWaitEvent(cs, e_fork, 0)); resolve_waits[slot] = false; }
        if ((st = timed_resolve(0))) return st;
        if (p->max_sample > p->min_sample) {
            if ((st = run_pipeline(D, slot, cs, W, *p, tmp, dc, dt, (uint32_t)q0, npix, p->min_sample,
                                   p->max_sample - p->min_sample, p->max_sample, 1, nullptr))) return st;
            if ((st = timed_resolve(1))) return st;
        }
        HIP_TRY(hipGetLastError());
        if (job) {
            InFlight f; f.q0 = q0; f.npix = npix;
            HIP_TRY(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(f.done, cs));
            flight.push_back(f);
            while ((int)flight.size() >= n_slots) if ((st = finish_oldest())) return st;
        }
    }
    while (job && !flight.empty()) if ((st = finish_oldest())) return st;
    if (n_slots > 1 || pipelined) {
        for (int i = pipelined ? 0 : 1; i < n_slots; i++) {
            hipEvent_t e_join;
            HIP_TRY(hipEventCreateWithFlags(&e_join, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(e_join, slot_stream(i)));
            HIP_TRY(hipStreamWaitEvent(stream, e_join, 0));
            (void)hipEventDestroy(e_join);
        }
    }
    if (want_stats) HIP_TRY(hipEventRecord(e_end, stream));
    if (sync || want_stats) {
        HIP_TRY(hipStreamSynchronize(stream));
        D->last_pending = false;
        D->last_pipelined = false;
        {
            // a dropped ray or photon query means a wrong image: never RT_OK, whether or not statistics were asked for
            unsigned long long tail[ST_COUNT - ST_QUEUE_OVERFLOW];
            HIP_TRY(stats_read(D->stats.p, ST_QUEUE_OVERFLOW, ST_COUNT - ST_QUEUE_OVERFLOW, tail));
            const unsigned long long drops = tail[0], peak_r = tail[ST_PEAK_RAYS - ST_QUEUE_OVERFLOW], peak_q = tail[ST_PEAK_QUERIES - ST_QUEUE_OVERFLOW];
            if (drops) {
                D->qhist.valid = false;
                if (D->qhist_used && !(job && job->stop.load())) return RT_ERR_OVERFLOW_RETRY;      // sized from history: once more, worst case
                return fail(RT_ERR_LIMIT, "render: a ray/photon queue overflowed (%llu drops); lower RT_CHUNK_SAMPLES or the bounce limit", drops);
            }
            const double per = (double)std::max<uint64_t>(1, ppc * (uint64_t)p->max_sample);
            DeviceState::QueueHistory &Hn = D->qhist;
            const double r = (double)peak_r / per, q = (double)peak_q / per;
            if (hist_ok) { Hn.rays_per_sample = std::max(Hn.rays_per_sample, r); Hn.queries_per_sample = std::max(Hn.queries_per_sample, q); }
            else { Hn.valid = true; Hn.model = p->shade_model; Hn.bounce = p->bounce; Hn.fan = fan; Hn.photons = use_photons; Hn.caustic = use_caustic;
                   Hn.rays_per_sample = r; Hn.queries_per_sample = q; }
        }
    } else {
        HIP_TRY(hipEventRecord(D->last_done, stream));
        D->last_pending = true;
        D->last_pipelined = pipelined;
        if (pipelined) D->frame_seq += n_chunks;
    }
    if (want_stats) {
        rt_stats R;
        memset(&R, 0, sizeof R);
        unsigned long long hs[ST_COUNT];
        HIP_TRY(stats_read(Ws[0].stats, 0, ST_COUNT, hs));
        R.rays_primary = hs[ST_RAYS_PRIMARY]; R.rays_shadow = hs[ST_RAYS_SHADOW]; R.rays_reflect = hs[ST_RAYS_REFLECT];
        R.rays_refract = hs[ST_RAYS_REFRACT]; R.instance_visits = hs[ST_INSTANCE_VISITS]; R.bvh_nodes_visited = hs[ST_BVH_NODES];
        R.tris_tested = hs[ST_TRIS]; R.photon_queries = hs[ST_PHOTON_QUERIES]; R.photons_visited = hs[ST_PHOTONS_VISITED];
        R.pixels = 0; R.samples = hs[ST_RAYS_PRIMARY];
        R.gather_rounds = hs[ST_GATHER_ROUNDS]; R.gather_slow = hs[ST_GATHER_SLOW]; R.gather_leaf_reads = hs[ST_GATHER_LEAF_READS];
        R.peak_rays = hs[ST_PEAK_RAYS]; R.peak_queries = hs[ST_PEAK_QUERIES];
        // per-stream intervals between consecutive marks: with two chunks in flight a kernel shares the GPU
        // with the other stream's kernels, so these are durations under overlap (the same thing rocprofv3 reports)
        for (int sl = 0; sl < n_slots; sl++)
            for (size_t i = 1; i < tm[sl].ev.size(); i++) {
                if (tm[sl].cls[i] < 0) continue;
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, tm[sl].ev[i - 1], tm[sl].ev[i]));
                if (tm[sl].cls[i] == 0) { R.ms_primary += ms; R.launches_primary++; }
                else if (tm[sl].cls[i] == 1) { R.ms_bounce += ms; R.launches_bounce += (uint64_t)std::max(0, std::min(p->shade_model == RT_SHADE_P6 ? 2 * p->bounce : p->bounce, 14)); }
                else { R.ms_gather += ms; R.launches_gather++; }
            }
        R.ms_trace = R.ms_primary + R.ms_bounce;
        R.launches_trace = R.launches_primary + R.launches_bounce;
        R.streams = (uint64_t)n_slots;
        for (auto &pr : resolve_ev) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            ms_resolve += ms; R.launches_resolve++;
        }
        R.ms_resolve = ms_resolve;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e_begin, e_end));
        R.ms_total = ms;
        for (uint64_t k = 0; k < (uint64_t)dt.n_tiles; k++) {
            const int t = dt.first + (int)k * dt.stride;
            const int tx = t % dt.tiles_x, ty = t / dt.tiles_x;
            const int w = std::min(dt.tile_w, cam->width - tx * dt.tile_w), h = std::min(dt.tile_h, cam->height - ty * dt.tile_h);
            if (w > 0 && h > 0) R.pixels += (uint64_t)w * h;
        }
        if (stats_out) *stats_out = R;
        if (job) job->stats = R;
    }
    return RT_OK;
}

extern "C" rt_status rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles,
                                            int device, void *hip_stream, uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev,
                                            int sync, rt_stats *stats_out)
{
    if (!s) return fail(RT_ERR_ARG, "rt_render_tiles_device: scene is NULL");
    return render_tiles(s, cam, p, tiles, device, (hipStream_t)hip_stream, hip_stream != nullptr, rgb8_dev, z_dev, count_dev,
                        sync != 0, stats_out, nullptr);
}

extern "C" rt_status rt_render_tiles_packed_device(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles,
                                                   int device, void *hip_stream, void *packed_dev, uint64_t packed_bytes,
                                                   int sync, rt_stats *stats_out)
{
    if (!s) return fail(RT_ERR_ARG, "rt_render_tiles_packed_device: scene is NULL");
    if (!packed_dev) return fail(RT_ERR_ARG, "rt_render_tiles_packed_device: the packed buffer is required");
    rt_status st = validate_render(s, cam, p, tiles);
    if (st) return st;
    uint64_t need = 0;
    if ((st = rt_tiles_packed_size(cam->width, cam->height, tiles, &need, nullptr))) return st;
    if (packed_bytes < need) return fail(RT_ERR_ARG, "rt_render_tiles_packed_device: buffer of %llu bytes, this call's tiles need %llu",
                                         (unsigned long long)packed_bytes, (unsigned long long)need);
    return render_tiles(s, cam, p, tiles, device, (hipStream_t)hip_stream, hip_stream != nullptr, nullptr, nullptr, nullptr,
                        sync != 0, stats_out, nullptr, packed_dev);
}

extern "C" rt_status rt_tiles_packed_size(int32_t width, int32_t height, const rt_tile_range *t, uint64_t *bytes, int32_t *n_tiles)
{
    if (!t || width <= 0 || height <= 0 || t->tile_w <= 0 || t->tile_h <= 0 || t->stride <= 0 || t->first < 0)
        return fail(RT_ERR_ARG, "rt_tiles_packed_size: bad argument");
    const int64_t tiles_x = (width + t->tile_w - 1) / t->tile_w, tiles_y = (height + t->tile_h - 1) / t->tile_h;
    const int64_t total = tiles_x * tiles_y;
    const int64_t n = t->first >= total ? 0 : (total - t->first + t->stride - 1) / t->stride;
    if (bytes) *bytes = (uint64_t)n * (uint64_t)t->tile_w * (uint64_t)t->tile_h * 8ull;
    if (n_tiles) *n_tiles = (int32_t)n;
    return RT_OK;
}

extern "C" rt_status rt_tiles_unpack_device(int device, void *hip_stream, const void *gathered_dev, int32_t world, int32_t tiles_per_rank,
                                            int32_t width, int32_t height, int32_t tile_w, int32_t tile_h,
                                            uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev)
{
    if (!gathered_dev || !rgb8_dev || !z_dev || !count_dev) return fail(RT_ERR_ARG, "rt_tiles_unpack_device: NULL buffer");
    if (world <= 0 || width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0) return fail(RT_ERR_ARG, "rt_tiles_unpack_device: bad geometry");
    const int64_t total = (int64_t)((width + tile_w - 1) / tile_w) * ((height + tile_h - 1) / tile_h);
    if ((int64_t)tiles_per_rank * world < total || tiles_per_rank < (total + world - 1) / world)
        return fail(RT_ERR_ARG, "rt_tiles_unpack_device: %d tiles per rank x %d ranks cannot hold %lld tiles", tiles_per_rank, world, (long long)total);
    if (!device_is_gfx950(device)) return fail(RT_ERR_NO_DEVICE, "rt_tiles_unpack_device: device %d is not gfx950 (no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    rtk_launch_unpack_tiles((hipStream_t)hip_stream, gathered_dev, world, tiles_per_rank, width, height, tile_w, tile_h, rgb8_dev, z_dev, count_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" rt_status rt_render_check(rt_scene *s, int device)
{
    if (!s) return fail(RT_ERR_ARG, "rt_render_check: scene is NULL");
    DeviceState *D = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        for (DeviceState *d : s->devs) if (d->device == device) D = d;
    }
    if (!D) return RT_OK;                                    // nothing was ever rendered there
    HIP_TRY(hipSetDevice(device));
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_render_check: another call on this scene is using device %d", device);
    unsigned long long drops = 0;
    if (D->last_pending && D->last_done) {
        HIP_TRY(hipEventSynchronize(D->last_done));
        D->last_pending = false;
        rt_status st = read_overflow(D, &drops);
        if (st) return st;
    }
    const bool earlier = D->async_overflow;
    D->async_overflow = false;
    if (drops || earlier) {
        D->qhist.valid = false;                             // the next render starts from the worst case again
        if (drops) HIP_TRY(hipMemset((unsigned long long *)D->stats.p + ST_AT(ST_QUEUE_OVERFLOW), 0, 8));
        return fail(RT_ERR_LIMIT, "rt_render_check: a ray/photon queue overflowed (%llu drops) in an asynchronous render", drops);
    }
    return RT_OK;
}

extern "C" rt_status rt_render_counters(rt_scene *s, int device, int reset, rt_stats *out)
{
    if (!s || !out) return fail(RT_ERR_ARG, "rt_render_counters: NULL argument");
    memset(out, 0, sizeof *out);
    DeviceState *D = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        for (DeviceState *d : s->devs) if (d->device == device) D = d;
    }
    if (!D || !D->stats.p) return RT_OK;                     // nothing was ever rendered there
    HIP_TRY(hipSetDevice(device));
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_render_counters: another call on this scene is using device %d", device);
    // (the verdict of the asynchronous renders stays with rt_render_check: last_pending is left as it is)
    if (D->last_pending && D->last_done) HIP_TRY(hipEventSynchronize(D->last_done));
    unsigned long long hs[ST_COUNT];
    HIP_TRY(stats_read(D->stats.p, 0, ST_COUNT, hs));
    out->rays_primary = hs[ST_RAYS_PRIMARY]; out->rays_shadow = hs[ST_RAYS_SHADOW]; out->rays_reflect = hs[ST_RAYS_REFLECT];
    out->rays_refract = hs[ST_RAYS_REFRACT]; out->instance_visits = hs[ST_INSTANCE_VISITS]; out->bvh_nodes_visited = hs[ST_BVH_NODES];
    out->tris_tested = hs[ST_TRIS]; out->photon_queries = hs[ST_PHOTON_QUERIES]; out->photons_visited = hs[ST_PHOTONS_VISITED];
    out->samples = hs[ST_RAYS_PRIMARY];
    out->gather_rounds = hs[ST_GATHER_ROUNDS]; out->gather_slow = hs[ST_GATHER_SLOW]; out->gather_leaf_reads = hs[ST_GATHER_LEAF_READS];
    out->peak_rays = hs[ST_PEAK_RAYS]; out->peak_queries = hs[ST_PEAK_QUERIES];
    if (reset) {
        // the drop counter and the queue peaks belong to the overflow verdict and the queue sizing: not touched
        static_assert(ST_QUEUE_OVERFLOW == ST_PHOTONS_VISITED + 1 && ST_GATHER_ROUNDS == ST_QUEUE_OVERFLOW + 1 && ST_PEAK_RAYS == ST_GATHER_LEAF_READS + 1, "counter layout");
        HIP_TRY(hipMemset(D->stats.p, 0, ST_AT(ST_QUEUE_OVERFLOW) * 8));
        HIP_TRY(hipMemset((unsigned long long *)D->stats.p + ST_AT(ST_GATHER_ROUNDS), 0, ST_AT(ST_PEAK_RAYS - ST_GATHER_ROUNDS) * 8));
    }
    return RT_OK;
}

static rt_status generate_photons(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed, const char *dat_path,
                                  rt_setup_ms *ms_out, bool own_job);

extern "C" rt_status rt_render_begin(rt_scene *s, const rt_camera *cam, const rt_params *p, const rt_tile_range *tiles, int device,
                                     uint8_t *rgb8, float *z, uint8_t *count, rt_job **out)
{
    if (!s || !out) return fail(RT_ERR_ARG, "rt_render_begin: scene/out is NULL");
    rt_status st = validate_render(s, cam, p, tiles);
    if (st) return st;
    if (!rgb8 || !z || !count) return fail(RT_ERR_ARG, "rt_render_begin: output buffers are required");
    DeviceState *D = nullptr;
    if ((st = prepare_device(s, device, &D))) return st;      // fail early (and loudly) when there is no GPU
    rt_job *job = new rt_job;
    job->scene = s;
    job->host_rgb = rgb8; job->host_z = z; job->host_count = count;
    s->live_jobs.fetch_add(1);
    const rt_camera camv = *cam; const rt_params pv = *p; const rt_tile_range tv = *tiles;
    job->worker = std::thread([=]() {
        rt_status r = RT_OK;
        const size_t npx = (size_t)camv.width * camv.height;
        uint8_t *d_rgb = nullptr, *d_cnt = nullptr; float *d_z = nullptr;
        auto body = [&]() -> rt_status {
            HIP_TRY(hipSetDevice(device));
            // BeginRender calls generatePhotonMap() before it spawns its workers (FIN/main.cpp:984-998, :350-402); here the
            // photon pass runs on the job's thread, so the call itself still returns at once and progress stays 0 meanwhile
            if (pv.shade_model == RT_SHADE_FIN && pv.photon_count > 0) {
                std::lock_guard<std::mutex> gen(s->gen_mu);         // one job per device may have been started: the first one generates
                bool have_map, have_source = false;
                std::string dump;
                {
                    std::lock_guard<std::mutex> lk(s->mu);
                    // a generated map serves only the parameters it was generated with (see rt_scene::gen)
                    const bool stale = s->gen.valid && (s->gen.count != (uint32_t)pv.photon_count || s->gen.bounce != pv.photon_bounce || s->gen.seed != pv.seed);
                    have_map = s->photon_count() != 0 && !stale;
                    dump = s->photon_dump;
                }
                for (const rt_light &l : s->data.lights) if (l.type == RT_LIGHT_POINT) have_source = true;
                if (!have_map && have_source && !job->stop.load()) {
                    const rt_status g = generate_photons(s, device, (uint32_t)pv.photon_count, pv.photon_bounce, pv.seed, dump.empty() ? nullptr : dump.c_str(),
                                                         &job->setup, true);
                    if (g) return g;
                }
            }
            HIP_TRY(hipMalloc((void **)&d_rgb, npx * 3)); HIP_TRY(hipMalloc((void **)&d_z, npx * 4)); HIP_TRY(hipMalloc((void **)&d_cnt, npx));
            HIP_TRY(hipMemcpy(d_rgb, rgb8, npx * 3, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_z, z, npx * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_cnt, count, npx, hipMemcpyHostToDevice));
            // render_tiles copies every finished band of rows back into the caller's buffers
            return render_tiles(s, &camv, &pv, &tv, device, nullptr, false, d_rgb, d_z, d_cnt, true, nullptr, job);
        };
        r = body();
        if (d_rgb) (void)hipFree(d_rgb);
        if (d_z) (void)hipFree(d_z);
        if (d_cnt) (void)hipFree(d_cnt);
        job->status = r;
        if (r) job->error = g_err;
        job->done.store(true);
        s->live_jobs.fetch_sub(1);
    });
    *out = job;
    return RT_OK;
}

extern "C" int rt_render_progress(rt_job *j) { return j ? j->progress.load() : 0; }
extern "C" rt_status rt_render_stop(rt_job *j) { if (!j) return fail(RT_ERR_ARG, "rt_render_stop: job is NULL"); j->stop.store(true); return RT_OK; }
extern "C" rt_status rt_render_wait(rt_job *j)
{
    if (!j) return fail(RT_ERR_ARG, "rt_render_wait: job is NULL");
    if (j->worker.joinable()) j->worker.join();
    if (j->status) g_err = j->error;
    return j->status;
}
extern "C" rt_status rt_job_stats(rt_job *j, rt_stats *out)
{
    if (!j || !out) return fail(RT_ERR_ARG, "rt_job_stats: NULL argument");
    if (!j->done.load()) return fail(RT_ERR_STATE, "rt_job_stats: job still running");
    *out = j->stats;
    return RT_OK;
}
extern "C" rt_status rt_job_setup_ms(rt_job *j, rt_setup_ms *out)
{
    if (!j || !out) return fail(RT_ERR_ARG, "rt_job_setup_ms: NULL argument");
    if (!j->done.load()) return fail(RT_ERR_STATE, "rt_job_setup_ms: job still running");
    *out = j->setup;
    return RT_OK;
}
extern "C" void rt_job_destroy(rt_job *j)
{
    if (!j) return;
    j->stop.store(true);
    if (j->worker.joinable()) j->worker.join();
    delete j;
}

// ---- single-stage entry points --------------------------------------------------------------------------
extern "C" rt_status rt_trace_rays(rt_scene *s, int shade_model, int device, const float *rays, int64_t n, uint8_t *hit, float *z,
                                   float *p, float *N, int32_t *node, uint8_t *front)
{
    if (!s || n < 0) return fail(RT_ERR_ARG, "rt_trace_rays: NULL scene or negative count");
    if (n > 0 && (!rays || !hit || !z || !p || !N || !node || !front)) return fail(RT_ERR_ARG, "rt_trace_rays: NULL argument");
    if (shade_model < RT_SHADE_FIN || shade_model > RT_SHADE_P3) return fail(RT_ERR_ARG, "rt_trace_rays: unknown shade model");
    DeviceState *D = nullptr;
    rt_status st = prepare_device(s, device, &D);
    if (st) return st;
    if (n == 0) return RT_OK;
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_trace_rays: another call on this scene is using device %d", device);
    if ((st = order_after_pending(D, D->stream))) return st;
    const size_t sizes[6] = {(size_t)n, (size_t)n * 4, (size_t)n * 12, (size_t)n * 12, (size_t)n * 4, (size_t)n};
    if ((st = D->t_in.upload(rays, (size_t)n * 24))) return st;
    for (int k = 0; k < 6; k++) if ((st = D->t_out[k].ensure(sizes[k]))) return st;
    rtk_launch_trace(D->stream, D->scene, shade_model, (const float *)D->t_in.p, n, (uint8_t *)D->t_out[0].p, (float *)D->t_out[1].p,
                     (float *)D->t_out[2].p, (float *)D->t_out[3].p, (int32_t *)D->t_out[4].p, (uint8_t *)D->t_out[5].p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(D->stream));
    void *dst[6] = {hit, z, p, N, node, front};
    for (int k = 0; k < 6; k++) HIP_TRY(hipMemcpy(dst[k], D->t_out[k].p, sizes[k], hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" rt_status rt_estimate_irradiance(rt_scene *s, int device, int32_t k, float radius, const float *pos, const float *normal,
                                            int64_t n, float *irr, float *dir)
{
    if (!s || n < 0) return fail(RT_ERR_ARG, "rt_estimate_irradiance: NULL scene or negative count");
    if (n > 0 && (!pos || !normal || !irr || !dir)) return fail(RT_ERR_ARG, "rt_estimate_irradiance: NULL argument");
    if (k < 1 || k > 65536 || !(radius > 0)) return fail(RT_ERR_ARG, "rt_estimate_irradiance: bad k/radius");
    DeviceState *D = nullptr;
    rt_status st = prepare_device(s, device, &D);
    if (st) return st;
    if (n == 0) return RT_OK;
    if (n > (1LL << 28)) return fail(RT_ERR_LIMIT, "rt_estimate_irradiance: too many queries");
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_estimate_irradiance: another call on this scene is using device %d", device);
    if ((st = order_after_pending(D, D->stream))) return st;
    if (!D->scene.pm.n_leaves) { memset(irr, 0, (size_t)n * 12); memset(dir, 0, (size_t)n * 12); return RT_OK; }
    std::vector<float4> qa((size_t)n), qb((size_t)n), qc((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        qa[i] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], normal[3 * i]);
        qb[i] = make_float4(normal[3 * i + 1], normal[3 * i + 2], 0, 0);
        qc[i] = make_float4(0, 0, 0, 0);
    }
    std::vector<uint32_t> cnt(1 + (size_t)RT_GATHER_CTRS * RT_CTR_STRIDE, 0u);  // query count, then the work counters as k_gather lays them out
    cnt[0] = (uint32_t)n;
    if ((st = D->t_out[0].upload(qa.data(), (size_t)n * 16))) return st;
    if ((st = D->t_out[1].upload(qb.data(), (size_t)n * 16))) return st;
    if ((st = D->t_out[2].upload(qc.data(), (size_t)n * 16))) return st;
    if ((st = D->t_out[3].ensure((size_t)n * 12))) return st;
    if ((st = D->t_out[4].ensure((size_t)n * 12))) return st;
    if ((st = D->t_in.upload(cnt.data(), cnt.size() * 4))) return st;
    if ((st = ensure_cell_start(D, false, k, radius, D->stream))) return st;
    rtk_launch_gather(D->stream, D->scene.pm, (const float4 *)D->t_out[0].p, (const float4 *)D->t_out[1].p, (const float4 *)D->t_out[2].p,
                      (const uint32_t *)D->t_in.p, cnt[0], k, radius, nullptr, (float *)D->t_out[3].p, (float *)D->t_out[4].p, 1, nullptr, GATHER_BLOCKS,
                      (uint32_t *)D->t_in.p + 1, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(D->stream));
    HIP_TRY(hipMemcpy(irr, D->t_out[3].p, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dir, D->t_out[4].p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" rt_status rt_shade_rays(rt_scene *s, const rt_params *p, int device, const float *rays, int64_t n, uint8_t *hit, float *rgb, float *z)
{
    if (!s || !p || n < 0) return fail(RT_ERR_ARG, "rt_shade_rays: NULL scene/params or negative count");
    if (n > 0 && (!rays || !hit || !rgb || !z)) return fail(RT_ERR_ARG, "rt_shade_rays: NULL argument");
    rt_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.width = 1; cam.height = 1; cam.fov = 40; cam.focaldist = 1; cam.dir[2] = -1; cam.up[1] = 1;
    rt_tile_range tr = {1, 1, 0, 1};
    rt_params pv = *p;
    pv.min_sample = pv.max_sample = 1;
    rt_status st = validate_render(s, &cam, &pv, &tr);
    if (st) return st;
    DeviceState *D = nullptr;
    if ((st = prepare_device(s, device, &D))) return st;
    if (n == 0) return RT_OK;
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_shade_rays: another call on this scene is using device %d", device);
    if ((st = order_after_pending(D, D->stream))) return st;
    const size_t limit = chunk_samples_limit(true);
    const size_t chunk = (size_t)std::min<int64_t>(n, (int64_t)limit);
    if ((st = ensure_workspace(D, 0, chunk, pv.bounce, 1, 2, pv.caustic_k > 0 && D->scene.cm.n_leaves != 0))) return st;
    const DevWork W = make_work(D, 0);
    DevCamera dc; camera_setup(cam, dc);
    DevTiles dt; memset(&dt, 0, sizeof dt);
    HIP_TRY(stats_zero(W.stats, 0, ST_COUNT, D->stream));
    for (int64_t off = 0; off < n; off += (int64_t)chunk) {
        const uint32_t m = (uint32_t)std::min<int64_t>((int64_t)chunk, n - off);
        if ((st = D->t_in.upload(rays + 6 * off, (size_t)m * 24))) return st;
        HIP_TRY(hipMemsetAsync(W.sample_hit, 0, m, D->stream));
        HIP_TRY(hipMemsetAsync(W.sample_rgb, 0, (size_t)m * 12, D->stream));
        if ((st = run_pipeline(D, 0, D->stream, W, pv, nullptr, dc, dt, (uint32_t)off, m, 0, 1, 1, 2, (const float *)D->t_in.p))) return st;
        HIP_TRY(hipStreamSynchronize(D->stream));
        HIP_TRY(hipMemcpy(hit + off, W.sample_hit, m, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(rgb + 3 * off, W.sample_rgb, (size_t)m * 12, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(z + off, W.sample_z, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    unsigned long long hs[ST_COUNT];
    HIP_TRY(stats_read(W.stats, 0, ST_COUNT, hs));
    if (hs[ST_QUEUE_OVERFLOW]) return fail(RT_ERR_LIMIT, "rt_shade_rays: queue overflow");
    for (int64_t i = 0; i < n; i++) if (!hit[i]) z[i] = 1.0e30f;
    return RT_OK;
}

// ---- photon pass ----------------------------------------------------------------------------------------
// generatePhotonMap's two loops (P13/main.cpp:338-404): mode 0 = the photon map (stop when max_count photons are STORED),
// mode 1 = the caustic map (stop when max_count diffuse hits are COUNTED; only those behind more than one specular
// hit are stored).  Attempts are consumed in order, the count checked between attempts like the reference's while().
// Everything stays on the device: k_photon_trace writes each attempt's photons, rtk_photon_compact (rt_photon_build.hip)
// consumes the attempts in order and packs the stored photons into D->raw_photons (1-based, [0] zero), then
// ScalePhotonPowers.  The caller holds the device claim.
struct CompactStateHost { unsigned long long attempts, counted; uint32_t stored, pad; };
static rt_status photon_pass_device(rt_scene *s, DeviceState *D, uint32_t max_count, int photon_bounce, uint32_t seed, int mode,
                                    uint32_t *n_out, uint64_t *attempts_out, const char *who)
{
    if (max_count == 0 || photon_bounce < 1 || photon_bounce > 8) return fail(RT_ERR_ARG, "%s: need a positive count and 1 <= photon_bounce <= 8", who);
    if (max_count > (1u << 28)) return fail(RT_ERR_LIMIT, "%s: at most 2^28 photons", who);
    bool have_source = false;
    for (const rt_light &l : s->data.lights) if (l.type == RT_LIGHT_POINT) have_source = true;
    if (!have_source) return fail(RT_ERR_STATE, "%s: the scene has no photon source (point light)", who);
    rt_status st;
    {   // D->raw_photons is about to be overwritten: a generated map that lives only there goes to the host first
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->gen_dev == D && s->gen_n && s->photons_raw.empty() && (st = raw_to_host(s))) return st;
    }
    const uint32_t batch = 1u << 18;
    static_assert((1u << 18) / RT_BLOCK <= RT_SPILL_BLOCKS, "DevScene::bvh_spill is sized for RT_SPILL_BLOCKS workgroups");
    const uint32_t out_cap = max_count + 8 + 1;                 // index 0 unused, up to 7 photons of overshoot
    if ((st = D->t_out[0].ensure((size_t)batch * 8 * 9 * 4))) return st;
    if ((st = D->t_out[1].ensure((size_t)batch * 4))) return st;
    if ((st = D->t_out[2].ensure(2 * sizeof(CompactStateHost)))) return st;
    const size_t sbytes = rtk_photon_compact_scratch(batch);
    if ((st = D->t_out[3].ensure(sbytes))) return st;
    if ((st = D->raw_photons.ensure((size_t)out_cap * sizeof(rt_photon)))) return st;
    hipStream_t stream = D->stream;
    HIP_TRY(hipMemsetAsync(D->t_out[2].p, 0, 2 * sizeof(CompactStateHost), stream));
    HIP_TRY(hipMemsetAsync(D->raw_photons.p, 0, sizeof(rt_photon), stream));
    CompactStateHost h{0, 0, 0, 0};
    int empty_batches = 0;
    while (h.counted < max_count) {
        rtk_launch_photon_trace(stream, D->scene, h.attempts, batch, seed, photon_bounce, (float *)D->t_out[0].p, (uint32_t *)D->t_out[1].p, mode);
        rtk_photon_compact(stream, (const float *)D->t_out[0].p, (const uint32_t *)D->t_out[1].p, batch, mode, max_count, D->t_out[2].p,
                           (rt_photon *)D->raw_photons.p, out_cap, D->t_out[3].p, sbytes);
        HIP_TRY(hipGetLastError());
        const unsigned long long before = h.counted;
        HIP_TRY(hipMemcpyAsync(&h, D->t_out[2].p, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h.counted == before && ++empty_batches >= 4) return fail(RT_ERR_STATE, "%s: no photon is ever stored in this scene", who);
    }
    uint32_t n = h.stored;
    if (n + 1 > out_cap) n = out_cap - 1;
    if (n > 0) rtk_photon_scale(stream, (rt_photon *)D->raw_photons.p, n, (float)(1.0 * 4 * M_PI / n));      // ScalePhotonPowers(1.0*4*M_PI/NumPhotons), :366 / :400
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = n;
    if (attempts_out) *attempts_out = h.attempts;
    return RT_OK;
}

static rt_status photon_pass(rt_scene *s, int device, uint32_t max_count, int photon_bounce, uint32_t seed, int mode,
                             rt_photon *out, uint32_t out_cap, uint32_t *n_out, uint64_t *attempts_out, const char *who)
{
    if (!s || !out || !n_out) return fail(RT_ERR_ARG, "%s: NULL argument", who);
    if (max_count == 0 || photon_bounce < 1 || photon_bounce > 8) return fail(RT_ERR_ARG, "%s: need a positive count and 1 <= photon_bounce <= 8", who);
    if (out_cap < max_count + 8 + 1) return fail(RT_ERR_ARG, "%s: out must hold the count + 9 records (index 0 unused, up to 7 photons of overshoot)", who);
    DeviceState *D = nullptr;
    rt_status st = prepare_device(s, device, &D);
    if (st) return st;
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "%s: another call on this scene is using device %d", who, device);
    if ((st = order_after_pending(D, D->stream))) return st;
    uint32_t n = 0;
    if ((st = photon_pass_device(s, D, max_count, photon_bounce, seed, mode, &n, attempts_out, who))) return st;
    HIP_TRY(hipMemcpy(out, D->raw_photons.p, ((size_t)n + 1) * sizeof(rt_photon), hipMemcpyDeviceToHost));
    *n_out = n;
    return RT_OK;
}

extern "C" rt_status rt_photon_pass(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed,
                                    rt_photon *out, uint32_t out_cap, uint32_t *n_out, uint64_t *attempts_out)
{
    return photon_pass(s, device, max_photons, photon_bounce, seed, 0, out, out_cap, n_out, attempts_out, "rt_photon_pass");
}

extern "C" rt_status rt_caustic_pass(rt_scene *s, int device, uint32_t max_diffuse_hits, int photon_bounce, uint32_t seed,
                                     rt_photon *out, uint32_t out_cap, uint32_t *n_out, uint64_t *attempts_out)
{
    return photon_pass(s, device, max_diffuse_hits, photon_bounce, seed, 1, out, out_cap, n_out, attempts_out, "rt_caustic_pass");
}

// generatePhotonMap as a whole (FIN/main.cpp:350-402).  `own_job`: called from the job thread of rt_render_begin (the scene
// counts that job as live; no other caller can be inside the scene then).
static rt_status generate_photons(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed, const char *dat_path,
                                  rt_setup_ms *ms_out, bool own_job)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (!s) return fail(RT_ERR_ARG, "rt_scene_generate_photons: scene is NULL");
    rt_status st;
    if (!own_job && (st = check_idle(s, "rt_scene_generate_photons"))) return st;
    rt_setup_ms T;
    memset(&T, 0, sizeof T);
    const auto t_begin = clk::now();
    DeviceState *D = nullptr;
    if ((st = prepare_device(s, device, &D))) return st;
    DeviceClaim claim(D);
    if (!claim.ok) return fail(RT_ERR_STATE, "rt_scene_generate_photons: another call on this scene is using device %d", device);
    if ((st = order_after_pending(D, D->stream))) return st;
    if (D->last_pending && D->last_done) HIP_TRY(hipEventSynchronize(D->last_done));       // the structure is rebuilt in place
    {   // the map this call replaces: if it lives only on this device, it is simply dropped (not fetched to the host first)
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->gen_n && s->gen_dev == D && s->photons_raw.empty()) { s->drop_generated(); s->invalidate(false, true); }
    }
    const auto t0 = clk::now();
    uint32_t n = 0;
    if ((st = photon_pass_device(s, D, max_photons, photon_bounce, seed, 0, &n, nullptr, "rt_scene_generate_photons"))) return st;
    const auto t1 = clk::now();
    T.photon_pass = ms(t0, t1);
    // Which photons balancing would put out of LocatePhotons' reach (heap slots > ReachablePhotonSlots(n): the last three or four):
    // found ON THE DEVICE along the root paths of those slots (rt_photon_build.hip: photon_unreachable); the photons stay where
    // they are.  They come to the host only for the .dat dump, or when a median's key is not unique in its segment (then the
    // reference's swap sequence decides and the host replays it: rt::UnreachablePhotons).
    std::vector<rt_photon> raw;
    auto fetch = [&]() -> rt_status {
        if (!raw.empty()) return RT_OK;
        raw.resize((size_t)n + 1);
        HIP_TRY(hipMemcpy(raw.data(), D->raw_photons.p, raw.size() * sizeof(rt_photon), hipMemcpyDeviceToHost));
        return RT_OK;
    };
    const auto t2 = clk::now();
    if (dat_path && dat_path[0]) {
        if ((st = fetch())) return st;
        if ((st = rt_photons_write_dat(dat_path, raw.data(), n))) return st;
    }
    const auto t2b = clk::now();
    T.upload = ms(t2, t2b);
    std::vector<uint32_t> skip;
    const uint32_t reach = rt::ReachablePhotonSlots(n);
    if (n && reach < n) {
        bool on_device = false;
        {
            struct TempBuf : DevBuf { ~TempBuf() { release(); } } scratch;
            const size_t sbytes = rtk_photon_unreachable_scratch(n);
            uint32_t res[16];
            if (getenv("RT_UNREACHABLE_ON_HOST") == nullptr && scratch.ensure(sbytes) == RT_OK &&
                rtk_photon_unreachable(D->stream, (const rt_photon *)D->raw_photons.p, n, reach + 1, n, scratch.p, sbytes, res) == hipSuccess &&
                res[1] == 0 && res[0] == n - reach) {
                skip.assign(res + 2, res + 2 + res[0]);
                std::sort(skip.begin(), skip.end());
                on_device = true;
            } else (void)hipGetLastError();
        }
        if (!on_device) {
            // the host's replay of BalanceSegment on (position, index) records packed on the device: 16 bytes per photon come over
            struct TempBuf : DevBuf { ~TempBuf() { release(); } } packed;
            if ((st = packed.ensure(((size_t)n + 1) * 16))) return st;
            rtk_photon_pack_positions(D->stream, (const rt_photon *)D->raw_photons.p, n, packed.p);
            std::vector<rt::PhotonPosRec> recs((size_t)n + 1);
            HIP_TRY(hipMemcpyAsync(recs.data(), packed.p, recs.size() * 16, hipMemcpyDeviceToHost, D->stream));
            HIP_TRY(hipStreamSynchronize(D->stream));
            rt::UnreachablePhotonRecs(recs.data(), n, skip);
        }
    }
    const auto t3 = clk::now();
    T.balance = ms(t2b, t3);
    std::vector<uint32_t> skip0;
    for (uint32_t i : skip) skip0.push_back(i - 1);
    double up = 0, build = 0;
    if ((st = build_photon_structure(D, false, (const rt_photon *)D->raw_photons.p + 1, nullptr, n, skip0.data(), (uint32_t)skip0.size(), &up, &build))) return st;
    T.upload += up; T.structure_build = build;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->data.photons.clear();
        s->photons_raw.swap(raw);                 // empty unless the photons had to come to the host
        s->photons_skip.swap(skip);
        s->gen.valid = true; s->gen.count = max_photons; s->gen.bounce = photon_bounce; s->gen.seed = seed;
        s->gen_dev = D; s->gen_n = n;
        for (DeviceState *d : s->devs) if (d != D) d->photons_valid = false;
    }
    T.total = ms(t_begin, clk::now());
    if (ms_out) *ms_out = T;
    return RT_OK;
}

extern "C" rt_status rt_scene_generate_photons(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed,
                                               const char *dat_path, rt_setup_ms *ms_out)
{
    return generate_photons(s, device, max_photons, photon_bounce, seed, dat_path, ms_out, false);
}

extern "C" rt_status rt_scene_set_photon_dump(rt_scene *s, const char *dat_path)
{
    rt_status st = check_idle(s, "rt_scene_set_photon_dump");
    if (st) return st;
    std::lock_guard<std::mutex> lk(s->mu);
    s->photon_dump = dat_path ? dat_path : "";
    return RT_OK;
}

extern "C" rt_status rt_scene_get_photons(rt_scene *s, rt_photon *out, uint32_t cap, uint32_t *n_stored)
{
    if (!s) return fail(RT_ERR_ARG, "rt_scene_get_photons: scene is NULL");
    std::lock_guard<std::mutex> lk(s->mu);
    const uint32_t n = s->photon_count();
    if (n_stored) *n_stored = n;
    if (!out) return RT_OK;
    if (cap < n + 1) return fail(RT_ERR_ARG, "rt_scene_get_photons: buffer of %u records, %u needed", cap, n + 1);
    if (n == 0) { memset(out, 0, sizeof(rt_photon)); return RT_OK; }
    { rt_status st = raw_to_host(s); if (st) return st; }
    if (!s->photons_raw.empty() && s->data.photons.empty()) {
        // PrepareForIrradianceEstimation (cyPhotonMap.h:196-218) of the generated photons, bit for bit, on first request
        std::vector<rt_photon> tmp = s->photons_raw;
        s->data.photons.resize(tmp.size());
        rt::BalancePhotons(tmp.data(), n, s->data.photons.data());
    }
    memcpy(out, s->data.photons.data(), ((size_t)n + 1) * sizeof(rt_photon));
    return RT_OK;
}
