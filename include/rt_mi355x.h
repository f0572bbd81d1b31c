/*
 * rt_mi355x.h -- C ABI of librt_mi355x.so, the MI355X (gfx950) replacement for the
 * per-pixel render loop of Roia2529/RayTracing-folder (RayTracingFinal / RayTracingProj13).
 *
 * The reference has no C ABI of its own: its boundary is a set of C++ abstract classes plus
 * globals plus three free functions.  Every entry point below names
 * the reference interface it replaces (paths relative to /root/reference;
 * FIN = RayTracingFinal/RayTracingFinal, P13 = RayTracingProj13/RayTracingProj13).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types.
 *   - every function returns RT_OK (0) or a negative rt_status; it never throws.
 *     rt_last_error() returns a thread-local message for the last failure.
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device pointers
 *     (e.g. torch.Tensor.data_ptr() of a cuda tensor).
 *   - images are row-major, row 0 = top of the image (FIN/include/scene.h:540-656).
 *   - there is NO CPU fallback: calls that need the GPU return RT_ERR_NO_DEVICE when no
 *     gfx950 device is visible.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4

typedef int rt_status;
#define RT_OK               0
#define RT_ERR_ARG         -1   /* bad argument (null pointer, out-of-range index, ...)   */
#define RT_ERR_STATE       -2   /* call not allowed in this state (e.g. job still live)  */
#define RT_ERR_NO_DEVICE   -3   /* no HIP device / not gfx950                            */
#define RT_ERR_DEVICE      -4   /* a HIP call failed (message has the HIP error string)  */
#define RT_ERR_IO          -5   /* file could not be read / parsed                       */
#define RT_ERR_LIMIT       -6   /* input exceeds a documented limit                      */

/* ---- scene records (all little-endian, packed exactly as declared) --------------------- */

/* object kinds: the reference's Object subclasses, FIN/include/objects.h:21,79,124 */
#define RT_OBJ_NONE   0
#define RT_OBJ_SPHERE 1
#define RT_OBJ_PLANE  2
#define RT_OBJ_MESH   3

/* One scene-graph node = the reference's Node (+ its Transformation base),
 * FIN/include/scene.h:224-262,438-514.  Matrices are column-major float[9] exactly like
 * cyMatrix3f::data (FIN/include/cyMatrix.h:290-296).  Nodes are listed parent-before-child
 * in the depth-first order TraceNode visits them (FIN/main.cpp:108-130); node 0 is the root
 * (parent = -1).  100 bytes. */
typedef struct rt_node {
    float   tm[9];      /* Transformation::tm  */
    float   itm[9];     /* Transformation::itm */
    float   pos[3];     /* Transformation::pos */
    int32_t parent;     /* index of parent node, -1 for the root */
    int32_t obj_type;   /* RT_OBJ_*                              */
    int32_t mesh;       /* mesh index for RT_OBJ_MESH, else -1   */
    int32_t material;   /* index into materials, -1 = none       */
} rt_node;

/* BVH node in the reference's 28-byte layout, FIN/include/cyBVH.h:187-200:
 * data bit31 = leaf; leaf: bits28-30 = count-1, bits0-27 = offset into elements;
 * internal: bits0-30 = index of first child (second child = first+1).  Root = node 1. */
typedef struct rt_bvh_node {
    float    box[6];
    uint32_t data;
} rt_bvh_node;

/* MtlBlinn parameter block, FIN/include/materials.h:377-383 (88 bytes). */
typedef struct rt_blinn {
    float diffuse[3];
    float specular[3];
    float reflection[3];
    float refraction[3];
    float emission[3];
    float absorption[3];
    float glossiness;
    float ior;
    float reflection_glossiness;
    float refraction_glossiness;
} rt_blinn;

/* lights, FIN/include/lights.h:30-174 */
#define RT_LIGHT_AMBIENT 0
#define RT_LIGHT_DIRECT  1
#define RT_LIGHT_POINT   2
typedef struct rt_light {
    int32_t type;
    float   intensity[3];
    float   position[3];    /* point light            */
    float   direction[3];   /* direct light (unit)    */
    float   size;           /* point light disc size  */
} rt_light;

/* Camera, FIN/include/scene.h:518-536; dir and up already orthonormalised the way
 * LoadScene does it (FIN/xmlload.cpp:124-127). */
typedef struct rt_camera {
    float   pos[3], dir[3], up[3];
    float   fov, focaldist, dof;
    int32_t width, height;
} rt_camera;

/* 24-byte photon, FIN/include/cyPhotonMap.h:47-65 (wire format of the reference's .dat dump,
 * FIN/main.cpp:398-400). */
typedef struct rt_photon {
    float    position[3];
    float    power;
    uint8_t  color[3];
    uint8_t  plane_and_dirz;   /* bits0-1 split plane, bit3 = direction z is negative */
    int16_t  dir_x, dir_y;
} rt_photon;

/* Textures (FIN/include/texture.h, FIN/texture.cpp, FIN/include/scene.h:323-434).
 * RT_TEX_FILE: width x height RGB8 texels at texel_offset (bytes) of the texel array, sampled
 * bilinearly with tiling (TextureFile::Sample); RT_TEX_CHECKER: TextureChecker (color1/color2). */
#define RT_TEX_FILE    1
#define RT_TEX_CHECKER 2
typedef struct rt_texture {
    int32_t  type;
    int32_t  width, height;
    uint32_t texel_offset;
    float    color1[3], color2[3];
} rt_texture;
/* TextureMap = a texture reference + a Transformation of the uvw coordinate (scene.h:376-398).
 * texture: index into the texture array; RT_MAP_NONE = the colour has no map;
 * RT_MAP_EMPTY = a map whose texture failed to load (samples black, as in the reference). */
#define RT_MAP_NONE  (-1)
#define RT_MAP_EMPTY (-2)
typedef struct rt_texmap {
    int32_t texture;
    float   tm[9], itm[9], pos[3];
} rt_texmap;

/* shading semantics: which snapshot's MtlBlinn::Shade / primitives to follow */
#define RT_SHADE_FIN 0   /* FIN/main.cpp:516-708 (+ FIN primitives, two-sided triangles)      */
#define RT_SHADE_P13 1   /* P13/main.cpp:485-756 (+ P13 primitives, back-face-culled tris)    */
#define RT_SHADE_P12 2   /* RayTracingProj12 main.cpp:341-588: P13's tree + live path-traced GI  */
#define RT_SHADE_P6  3   /* RayTracingProj6 main.cpp:175-340 (BASELINE config 2): children gated by
                            reflection/refraction.Gray() > 0, no light fall-off, no environment;
                            its RenderPixel = 1 sample at the pixel centre, no gamma:
                            min_sample = max_sample = 1, gamma = 1                                 */
#define RT_SHADE_P3  4   /* RayTracingProj3 main.cpp:152-221 (BASELINE config 1): spheres without
                            bias, direct light only, V = camera - p                                */

/* The reference's compile-time #defines (FIN/main.cpp:19-32, FIN/include/lights.h:16-18,
 * FIN/include/materials.h:20-25, FIN/main.cpp:699) as one runtime block.
 * rt_params_default() fills the FIN values. */
typedef struct rt_params {
    int32_t  min_sample;         /* MIN_SAMPLE 4                                    */
    int32_t  max_sample;         /* MAX_SAMPLE 8                                    */
    float    threshold;          /* THRESHOLD 1e-3 (variance gate); <0 = never      */
    int32_t  bounce;             /* BOUNCE 4                                        */
    int32_t  hemisphere_sample;  /* HEMISPHERE_SAMPLE 30 (result discarded in FIN)  */
    int32_t  knn_k;              /* EstimateIrradiance<400>                         */
    float    knn_radius;         /* radius = 1                                      */
    int32_t  shade_model;        /* RT_SHADE_*                                      */
    int32_t  shadow_samples;     /* MIN_SHADOW_SAMPLES 4 (identical rays at size 0) */
    uint32_t seed;               /* counter-RNG seed for stochastic features        */
    double   gamma;              /* #define gamma 2.2 (a double literal)            */
    /* caustic map (RT_SHADE_P13 / P12 only; P13/main.cpp:518-531): EstimateIrradiance<caustic_k> with
     * `caustic_radius` on the scene's caustic photons at diffuse hits reached with specount > 2;
     * caustic_k = 0 (the default, as in the committed reference where the lookup is commented out)
     * switches it off.  prj13.html quotes k = 400, r = 0.5. */
    int32_t  caustic_k;
    float    caustic_radius;
    /* generatePhotonMap (FIN/main.cpp:27,29,350-402): MAX_NUM_OF_PHOTON 1000000 and PHOTON_BOUNCE 8.  rt_render_begin --
     * the replacement of BeginRender, which calls generatePhotonMap() before it spawns its workers (:984-998) -- runs the
     * photon pass first, on the job's thread, when shade_model is RT_SHADE_FIN, photon_count > 0 and the scene holds no
     * photon map yet (rt_scene_set_photons / rt_scene_generate_photons); photon_count = 0 renders the scene as it is.
     * The generator is seeded with `seed`.  The device-side entry points (rt_render_tiles_*) never generate.
     * A map the library generated is DERIVED from the scene (ABI 4): rt_scene_set_nodes / _mesh / _materials / _lights and
     * rt_scene_load_xml drop it, and rt_render_begin makes a new one when photon_count, photon_bounce or seed differ from
     * the ones it was made with -- the reference regenerates on every BeginRender (FIN/main.cpp:984-990).  A map the caller
     * handed over with rt_scene_set_photons is the caller's and stays until it is replaced. */
    int32_t  photon_count;
    int32_t  photon_bounce;
} rt_params;

/* Which tiles of the image this call renders.  Tiles are tile_w x tile_h pixels, numbered
 * row-major over the image; this call renders tiles first, first+stride, ... (< total).
 * first=0, stride=1 renders the whole image.  Replaces pixelIterator (FIN/main.cpp:65-85). */
typedef struct rt_tile_range {
    int32_t tile_w, tile_h;
    int32_t first, stride;
} rt_tile_range;

/* per-job statistics (rays by class, traversal work, kernel time) */
typedef struct rt_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
    uint64_t instance_visits;       /* leaf-object transforms applied            */
    uint64_t bvh_nodes_visited;
    uint64_t tris_tested;
    uint64_t photon_queries;
    uint64_t photons_visited;
    uint64_t pixels;
    uint64_t samples;
    double   ms_trace;              /* sum of HIP-event durations: trace+shade kernels   */
    double   ms_gather;             /* photon gather kernels                              */
    double   ms_resolve;            /* resolve kernels                                    */
    double   ms_total;              /* wall time of the job on its stream (events)        */
    uint64_t launches_trace, launches_gather, launches_resolve;
    uint64_t gather_rounds;         /* (query, trial radius) pairs processed by the gather          */
    uint64_t gather_slow;           /* of those, how many overflowed the LDS leaf list               */
    uint64_t gather_leaf_reads;     /* photon slots read by the gather, all passes, in units of 32 slots (1 KiB) */
    /* ABI 2: the trace+shade time by kernel (ms_trace = ms_primary + ms_bounce).  Every ms_* field is a sum
     * of HIP-event intervals on the stream that ran the kernels.  With `streams` == 1 one chunk is in
     * flight at a time and the intervals are EXCLUSIVE kernel times; with more, a kernel shares the GPU
     * with the other chunks' kernels, the intervals overlap in time and their sum exceeds ms_total --
     * do not divide bytes by them then (bench.py takes its roofline from a streams == 1 pass). */
    double   ms_primary, ms_bounce;
    uint64_t launches_primary, launches_bounce;
    uint64_t streams;               /* chunks in flight at once during this call                     */
    uint64_t peak_rays, peak_queries;   /* largest ray-queue level / photon-query count of any chunk */
    uint64_t attempts;              /* ABI 3: 1, or 2 when queues sized from history overflowed and the library rendered the
                                       frame again with worst-case queues (the statistics are those of the last attempt) */
} rt_stats;

typedef struct rt_scene rt_scene;   /* opaque */
typedef struct rt_job   rt_job;     /* opaque */

/* ---- library ---------------------------------------------------------------------------- */
int          rt_abi_version(void);
const char  *rt_last_error(void);
/* number of usable gfx950 devices (0 when none); never fails */
int          rt_device_count(void);
void         rt_params_default(rt_params *p);

/* ---- scene: replaces the global singletons rootNode/materials/lights/objList/photonmap
 *      (FIN/main.cpp:36-49) ------------------------------------------------------------- */
rt_status rt_scene_create(rt_scene **out);
void      rt_scene_destroy(rt_scene *s);

/* Node tree as LoadScene builds it (FIN/xmlload.cpp:65-132,168-261). */
rt_status rt_scene_set_nodes(rt_scene *s, const rt_node *nodes, int32_t n);
/* One TriObj (FIN/include/objects.h:124-303): cyTriMesh arrays V/F/VN/FN
 * (FIN/include/cyTriMesh.h:104-111) and its cyBVH node/element arrays
 * (FIN/include/cyBVH.h:202-203).  f/fn are 3 indices per face.  nodes[0] is unused. */
rt_status rt_scene_set_mesh(rt_scene *s, int32_t mesh,
                            const float *v, int32_t nv,
                            const uint32_t *f, int32_t nf,
                            const float *vn, int32_t nvn, const uint32_t *fn,
                            const rt_bvh_node *nodes, int32_t nnodes,
                            const uint32_t *elements);
/* cyTriMesh VT/FT (FIN/include/cyTriMesh.h:108-109): the texture vertices (uvw triples) and the
 * 3 texture indices per face of a mesh already set.  Read by the PROJ13-family triangle, whose
 * hit takes uvw = GetTexCoord(face, barycentric) (P13/include/objects.h:203); the FINAL triangle
 * never writes uvw (FIN/include/objects.h:226-267).  nvt == 0 removes them; a PROJ13 mesh without
 * them leaves uvw as it was (the reference dereferences a null vt there).
 * rt_scene_set_mesh clears them, so call this after it. */
rt_status rt_scene_set_mesh_texcoords(rt_scene *s, int32_t mesh, const float *vt, int32_t nvt,
                                      const uint32_t *ft);
rt_status rt_scene_set_materials(rt_scene *s, const rt_blinn *m, int32_t n);
rt_status rt_scene_set_lights(rt_scene *s, const rt_light *l, int32_t n);
/* environment / background colour (FIN/include/scene.h:406-434; textures: not yet) */
rt_status rt_scene_set_environment(rt_scene *s, const float env_rgb[3], const float bg_rgb[3]);
rt_status rt_scene_get_environment(const rt_scene *s, float env_rgb[3], float bg_rgb[3]);
/* Texture store (replaces the TextureList global, FIN/main.cpp:46) and the maps of the colours the
 * render path samples: per material its diffuse and specular maps (MtlBlinn::Shade samples only
 * those two, FIN/main.cpp:531-532), the environment (SampleEnvironment, :635) and the background
 * (:328).  maps = 2 * n_materials records: [2m] diffuse, [2m+1] specular. */
rt_status rt_scene_set_textures(rt_scene *s, const rt_texture *tex, int32_t n, const uint8_t *texels, uint64_t n_bytes);
rt_status rt_scene_set_material_maps(rt_scene *s, const rt_texmap *maps, int32_t n_materials);
rt_status rt_scene_set_environment_maps(rt_scene *s, const rt_texmap *environment, const rt_texmap *background);
rt_status rt_scene_get_textures(const rt_scene *s, rt_texture *tex, int32_t cap, uint8_t *texels, uint64_t texel_cap,
                                int32_t *n_tex, uint64_t *n_bytes);
rt_status rt_scene_get_maps(const rt_scene *s, rt_texmap *material_maps, int32_t cap, rt_texmap *environment, rt_texmap *background);

/* Balanced photon array exactly as PhotonMap::photons after PrepareForIrradianceEstimation
 * (FIN/include/cyPhotonMap.h:196-218): photons[0] unused, photons[1..n_stored] heap-ordered.
 * n_stored = 0 clears the map (photon term contributes 0). */
rt_status rt_scene_set_photons(rt_scene *s, const rt_photon *photons, uint32_t n_stored);
/* The second map of RayTracingProj13 (`causticmap`, P13/main.cpp:338,379-404): same format, looked up only
 * by the P13-family shading models when rt_params.caustic_k > 0. */
rt_status rt_scene_set_caustic_photons(rt_scene *s, const rt_photon *photons, uint32_t n_stored);

/* Host-side loader with the reference's XML/OBJ schema (FIN/xmlload.cpp:65-554,
 * FIN/include/cyTriMesh.h:263-547, FIN/include/objects.h:137-145).  OBJ names resolve
 * relative to the XML file's directory.  Fills nodes/meshes/materials/lights/camera. */
rt_status rt_scene_load_xml(rt_scene *s, const char *path);
rt_status rt_scene_get_camera(const rt_scene *s, rt_camera *out);

/* Export the host-side arrays (so a caller can hand the same bytes to another consumer).
 * Pass NULL pointers to query counts only. */
rt_status rt_scene_counts(const rt_scene *s, int32_t *n_nodes, int32_t *n_meshes,
                          int32_t *n_materials, int32_t *n_lights, uint32_t *n_photons);
rt_status rt_scene_get_nodes(const rt_scene *s, rt_node *out, int32_t cap);
rt_status rt_scene_get_materials(const rt_scene *s, rt_blinn *out, int32_t cap);
rt_status rt_scene_get_lights(const rt_scene *s, rt_light *out, int32_t cap);
rt_status rt_scene_mesh_counts(const rt_scene *s, int32_t mesh, int32_t *nv, int32_t *nf,
                               int32_t *nvn, int32_t *nnodes);
rt_status rt_scene_get_mesh(const rt_scene *s, int32_t mesh, float *v, uint32_t *f, float *vn,
                            uint32_t *fn, rt_bvh_node *nodes, uint32_t *elements);
rt_status rt_scene_get_mesh_texcoords(const rt_scene *s, int32_t mesh, int32_t *nvt, float *vt,
                                      uint32_t *ft);

/* Image files: what TextureFile::Load gets from lodepng::decode(..., LCT_RGB) / LoadPPM
 * (FIN/texture.cpp:33-91) and RenderImage::SavePNG from lodepng::encode (FIN/include/scene.h:645-655).
 * rt_image_read_rgb with rgb == NULL only reports the size. comps: 1 = grey, 3 = RGB. */
rt_status rt_image_read_rgb(const char *path, int32_t *w, int32_t *h, uint8_t *rgb, uint64_t cap);
rt_status rt_image_write_png(const char *path, const uint8_t *data, int32_t w, int32_t h, int32_t comps);
/* The two derived images of RenderImage (FIN/include/scene.h:591-613 ComputeZBufferImage: 255*(zmax-z)/
 * (zmax-zmin) truncated, BIGFLOAT pixels 0 and ignored by the extrema; :615-637 ComputeSampleCountImage:
 * integer 255*(c-smin)/(smax-smin), all zero when smax == smin).  Bit-exact integer maps; *smax (may be
 * NULL) receives ComputeSampleCountImage's return value.  Host code, no GPU involved. */
rt_status rt_image_zbuffer(const float *zbuffer, int32_t w, int32_t h, uint8_t *zbuffer_img);
rt_status rt_image_sample_count(const uint8_t *sample_count, int32_t w, int32_t h, uint8_t *sample_count_img, int32_t *smax);

/* generatePhotonMap as a whole (FIN/main.cpp:350-402) on the GPU: the photon pass (rt_photon_pass), ScalePhotonPowers,
 * the optional dump of the unbalanced photons (:397-400 fwrite to a path the reference hard-codes; dat_path == NULL: no
 * dump), PrepareForIrradianceEstimation -- and the result becomes the scene's photon map on every device.  ms_out (may be
 * NULL) receives the wall time of the stages in milliseconds. */
typedef struct rt_setup_ms {
    double photon_pass;      /* emission + bounces + compaction of the stored photons (GPU)                      */
    double balance;          /* what of PrepareForIrradianceEstimation is needed to know the photons LocatePhotons
                                can reach (cyPhotonMap.h:217,371); host                                          */
    double structure_build;  /* decode + median-split sub-leaves + boxes + density grid                          */
    double upload;           /* host <-> device copies of photon records                                         */
    double total;
} rt_setup_ms;
rt_status rt_scene_generate_photons(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed,
                                    const char *dat_path, rt_setup_ms *ms_out);
/* where the photon pass that rt_render_begin runs (rt_params.photon_count) leaves its .dat dump; NULL (default) = no dump */
rt_status rt_scene_set_photon_dump(rt_scene *s, const char *dat_path);
/* the scene's photon map as PhotonMap::photons after PrepareForIrradianceEstimation (balanced, [0] unused): what
 * rt_scene_set_photons was given, or the balanced form of what rt_scene_generate_photons made.  out == NULL only counts
 * (*n_stored); cap counts records including [0]. */
rt_status rt_scene_get_photons(rt_scene *s, rt_photon *out, uint32_t cap, uint32_t *n_stored);

/* ---- host helpers that mirror reference host code --------------------------------------- */
/* cyBVH build with MeanSplit (FIN/include/cyBVH.h:122-142,295-328) as TriObj::Load calls it
 * (maxElementsPerNode = 4, FIN/include/objects.h:143).  nodes_out needs room for 2*nf+1
 * nodes; returns the node count (including the unused node 0) in *nnodes. */
rt_status rt_bvh_build(const float *v, int32_t nv, const uint32_t *f, int32_t nf,
                       int32_t max_per_leaf, rt_bvh_node *nodes_out, int32_t *nnodes,
                       uint32_t *elements_out);
/* PhotonMap::PrepareForIrradianceEstimation (FIN/include/cyPhotonMap.h:196-284):
 * in = photons[1..n] unordered (in[0] unused), out = balanced heap order (out[0] = in[0]).
 * `in` is permuted in place exactly like the reference permutes its vector. */
rt_status rt_photon_balance(rt_photon *in, uint32_t n, rt_photon *out);

/* Which photons of an UNBALANCED array (in[1..n], in[0] as rt_photon_balance takes it) LocatePhotons will never reach once
 * the array is balanced: it descends only while index < halfStoredPhotons = n/2 - 1 (FIN/include/cyPhotonMap.h:217,371), so
 * the last three or four heap slots are never visited.  Found by running BalanceSegment's own partitions along the root
 * paths of those slots only (about 2n element visits); `in` is not modified.  raw_indices (1-based, ascending) needs room
 * for 4; *count receives how many there are. */
rt_status rt_photon_unreachable(const rt_photon *in, uint32_t n, uint32_t *raw_indices, uint32_t cap, uint32_t *count);
/* ABI 4: the same search ON THE DEVICE (what rt_scene_generate_photons runs: the root paths of those heap slots are followed with
 * radix selects, the photons never leave HBM).  *exact = 0 when the key of some median on those paths is not unique in its
 * segment -- then the reference's swap sequence decides which photon lands where, and only rt_photon_unreachable (the host's
 * replay of BalanceSegment, FIN/include/cyPhotonMap.h:222-284) gives the answer; *count is 0 in that case. */
rt_status rt_photon_unreachable_device(int device, const rt_photon *in, uint32_t n, uint32_t *raw_indices, uint32_t cap, uint32_t *count, int32_t *exact);

/* The photon dump generatePhotonMap leaves behind (FIN/main.cpp:397-400: fwrite of
 * Photon[NumPhotons], before balancing) and the way the reference's viewer reads it back
 * (PhotonMap/PhotonMapViz.cpp:172-193: whole 24-byte records, a trailing partial one dropped).
 * photons[0] is unused on both sides, records are photons[1..n].  rt_photons_read_dat with
 * out == NULL only counts. */
rt_status rt_photons_write_dat(const char *path, const rt_photon *photons, uint32_t n);
rt_status rt_photons_read_dat(const char *path, rt_photon *out, uint32_t cap, uint32_t *n);

/* Photon pass on the GPU: generatePhotonMap up to and including ScalePhotonPowers
 * (FIN/main.cpp:350-396; PhotonTracing :439-459; PointLight::RandomPhoton :489-497;
 * MtlBlinn::RandomPhotonBounce FIN/include/materials.h:99-256).  rand() is replaced by a
 * counter-based generator (Philox-4x32-10, key = seed, counter = emission attempt, draw), so the
 * result depends only on (scene, seed).  Emission attempts are consumed in order until at least
 * max_photons are stored (like the reference, the last attempt may overshoot by up to 7).
 * out[0] is unused, out[1..*n_out] are the photons in the reference's 24-byte .dat format, powers
 * already scaled by 4*pi/n; balance them with rt_photon_balance before rt_scene_set_photons. */
rt_status rt_photon_pass(rt_scene *s, int device, uint32_t max_photons, int photon_bounce, uint32_t seed,
                         rt_photon *out, uint32_t out_cap, uint32_t *n_out, uint64_t *attempts_out);

/* The caustic pass of RayTracingProj13 (P13/main.cpp:379-404 + CausticTracing :431-457; a comment block in the
 * committed file): every emitted photon is followed for up to photon_bounce (CAUSTIC_PHOTON_BOUNCE 5)
 * RandomPhotonBounce steps; a hit on a diffuse surface is STORED only after more than one specular hit on
 * the way (hitspec > 1: e.g. into and out of the glass sphere) but COUNTED either way, and emission stops
 * once `max_diffuse_hits` (MAX_NUM_OF_CAUSTIC_PHOTON) are counted.  Same generator, output format and
 * 4*pi/n_stored power scaling as rt_photon_pass.  out needs room for max_diffuse_hits + 9 records. */
rt_status rt_caustic_pass(rt_scene *s, int device, uint32_t max_diffuse_hits, int photon_bounce, uint32_t seed,
                          rt_photon *out, uint32_t out_cap, uint32_t *n_out, uint64_t *attempts_out);


/* ---- rendering: replaces BeginRender/StopRender + RenderPixel + RenderImage progress
 *      (FIN/main.cpp:202-344,984-1012; FIN/include/scene.h:586-589) ---------------------- */
/* Asynchronous: returns after the job's worker thread has started.  Output buffers are
 * caller-owned host memory of width*height pixels (rgb8: 3 bytes per pixel) and may be read
 * at any time: finished bands of rows are copied into them chunk by chunk while the job runs
 * (the reference's viewport shows renderImage.GetPixels() as it fills, viewport.cpp:367), and
 * rt_render_progress counts only pixels that have already arrived.  Pixels of tiles this call
 * does not own, and of chunks not reached before rt_render_stop, keep the caller's values.
 * One render at a time per (scene, device): a second one started while the first still runs
 * fails with RT_ERR_STATE (reported by rt_render_wait for jobs); different devices may render
 * the same scene concurrently. */
rt_status rt_render_begin(rt_scene *s, const rt_camera *cam, const rt_params *p,
                          const rt_tile_range *tiles, int device,
                          uint8_t *rgb8, float *z, uint8_t *count, rt_job **out);
/* Same, but the outputs are DEVICE pointers on `device` and the work is enqueued on
 * `hip_stream` (an explicit hipStream_t; NULL = the library's own non-blocking stream, which is NOT
 * ordered with the caller's legacy default stream -- a caller that works on the default stream must
 * render under an explicit stream of its own and pass that); only this call's tiles are written.  Synchronous with respect to enqueueing; completion follows stream
 * order unless `sync` is non-zero.  A ray or photon query dropped by a full queue makes the image
 * wrong: with `sync` != 0 the call then returns RT_ERR_LIMIT (with or without stats_out); after
 * `sync` == 0 calls the verdict is collected by rt_render_check.  While an asynchronous render is
 * still in flight the next call on the same (scene, device) -- on whatever stream -- is ordered
 * behind it on the GPU. */
rt_status rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_params *p,
                                 const rt_tile_range *tiles, int device, void *hip_stream,
                                 uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev,
                                 int sync, rt_stats *stats_out);
/* Multi-GPU tile exchange (the reference shares one atomic pixel counter between its threads,
 * FIN/main.cpp:71-78).  A rank renders its tiles first, first+stride, ... straight into the buffer it
 * contributes to the all-gather: tile k of the call at packed_dev + k*tile_w*tile_h*8, pixels row-major
 * inside the tile, one 8-byte record per pixel = {r, g, b, z as 4 little-endian bytes, count}; slots of a
 * ragged tile that fall outside the image are zero.  rt_tiles_packed_size gives the bytes / tile count of
 * a call.  rt_tiles_unpack_device turns the gathered buffer of `world` ranks (rank r's block of
 * tiles_per_rank tiles at offset r, its tiles being r, r+world, ...) back into the three RenderImage
 * planes with one small HIP kernel on `hip_stream` (stream-ordered, no host synchronisation). */
rt_status rt_render_tiles_packed_device(rt_scene *s, const rt_camera *cam, const rt_params *p,
                                        const rt_tile_range *tiles, int device, void *hip_stream,
                                        void *packed_dev, uint64_t packed_bytes, int sync, rt_stats *stats_out);
rt_status rt_tiles_packed_size(int32_t width, int32_t height, const rt_tile_range *tiles, uint64_t *bytes, int32_t *n_tiles);
/* (hip_stream == NULL here means the device's legacy null stream, NOT the library's own stream: the call has no scene to
 * take one from; pass the stream the gather ran on.) */
rt_status rt_tiles_unpack_device(int device, void *hip_stream, const void *gathered_dev, int32_t world, int32_t tiles_per_rank,
                                 int32_t width, int32_t height, int32_t tile_w, int32_t tile_h,
                                 uint8_t *rgb8_dev, float *z_dev, uint8_t *count_dev);
/* Waits for the asynchronous renders (sync == 0) issued so far on (scene, device) and returns
 * RT_ERR_LIMIT if any of them dropped rays or photon queries, RT_OK otherwise. */
rt_status rt_render_check(rt_scene *s, int device);
/* ABI 4: the device-side work counters (rays by class, traversal visits, photon queries / photons examined, gather rounds:
 * the uint64 counting fields of rt_stats; its timings, launches and pixels stay zero) accumulated by every render on
 * (scene, device) since they were last cleared -- a render that collects statistics (stats_out, a job) clears them at its
 * start, asynchronous renders only add.  Waits for the renders issued so far.  `reset` != 0 clears them after the read:
 * read-and-reset before a run of asynchronous frames and read again after it to count exactly those frames (bench.py's timed
 * region).  The reference has no counterpart (its only statistic is the wall-clock timer of viewport.cpp:442). */
rt_status rt_render_counters(rt_scene *s, int device, int reset, rt_stats *out);
int       rt_render_progress(rt_job *j);      /* pixels finished so far (monotonic)       */
rt_status rt_render_stop(rt_job *j);          /* cooperative cancel (StopRender)          */
rt_status rt_render_wait(rt_job *j);          /* join; returns the job's final status     */
rt_status rt_job_stats(rt_job *j, rt_stats *out);
/* the photon pass the job ran before rendering (rt_params.photon_count): stage times, all zero when it ran none */
rt_status rt_job_setup_ms(rt_job *j, rt_setup_ms *out);
void      rt_job_destroy(rt_job *j);

/* ---- single-stage entry points (used by parity tests and by hosts that keep their own
 *      RenderPixel): inputs/outputs are HOST arrays, the work runs on the GPU -------------- */
/* n closest-hit queries = n calls of TraceNode(rootNode, ray, hit) (FIN/main.cpp:94-130).
 * rays: n x 6 floats (p, dir).  Outputs per ray: hit flag, z, p[3], N[3], node index, front. */
rt_status rt_trace_rays(rt_scene *s, int shade_model, int device, const float *rays, int64_t n,
                        uint8_t *hit, float *z, float *p, float *N, int32_t *node,
                        uint8_t *front);
/* n irradiance estimates = n calls of photonmap.EstimateIrradiance<k>(irr, dir, radius, pos,
 * &normal, 1, CONSTANT) (FIN/include/cyPhotonMap.h:288-336).  pos, normal: n x 3 floats;
 * outputs irr, dir: n x 3 floats. */
rt_status rt_estimate_irradiance(rt_scene *s, int device, int32_t k, float radius,
                                 const float *pos, const float *normal, int64_t n,
                                 float *irr, float *dir);
/* n shades of primary-ray samples = Trace + MtlBlinn::Shade(ray, hit, lights, bounce, 0)
 * (FIN/main.cpp:294-297): rays n x 6; outputs hit flag, rgb (linear, before gamma), z. */
rt_status rt_shade_rays(rt_scene *s, const rt_params *p, int device, const float *rays,
                        int64_t n, uint8_t *hit, float *rgb, float *z);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
