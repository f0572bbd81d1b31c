/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, recursive CPU restatement of the reference's per-pixel render
 * loop (RayTracingFinal/RayTracingFinal = FIN).  It exists to CHECK the HIP path; nothing in
 * the product may include, link or call it (only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do).
 *
 * Pinning status (see DESIGN.md "Oracle"): PINNED, bit for bit, against the reference's own code compiled here --
 *   oracle/_ref/ref_harness_* (the headers where they lie in /root/reference):
 *       Sphere/Plane/TriObj::IntersectRay + TraceBVHNode + Box::IntersectRay, cyBVH build,
 *       Node::ToNodeCoords/FromNodeCoords, Halton, Color24, PhotonMap balance + kNN
 *       (EstimateIrradiance<400>), Photon pack/decode, PointLight::Illuminate, RandomPhotonBounce, textures;
 *   oracle/_ref/ref_main_harness_* (main.cpp of RayTracingFinal / RayTracingProj13 itself, compiled with only its
 *   `#include "viewport.cpp"` and `ShowViewport();` lines removed: no GL header needed):
 *       TraceNode, GenLight::Shadow, MtlBlinn::Shade, RenderPixel, PhotonTracing / CausticTracing, RandomPhoton.
 *   The Shade variants of RayTracingProj12 / 6 / 3 are restated from snapshots that are not compiled here.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh {
    const float *v;      int32_t nv;
    const uint32_t *f;   int32_t nf;
    const float *vn;     int32_t nvn;
    const uint32_t *fn;
    const rt_bvh_node *nodes; int32_t nnodes;
    const uint32_t *elements;
    const float *vt;     int32_t nvt;     /* texture vertices (uvw triples) or NULL */
    const uint32_t *ft;                   /* 3 texture indices per face or NULL */
} orc_mesh;

typedef struct orc_scene {
    const rt_node *nodes;       int32_t n_nodes;
    const orc_mesh *meshes;     int32_t n_meshes;
    const rt_blinn *materials;  int32_t n_materials;
    const rt_light *lights;     int32_t n_lights;
    const rt_photon *photons;   uint32_t n_photons;   /* balanced, [0] unused */
    float env[3], bg[3];
    /* textures (all optional) */
    const rt_texture *textures; int32_t n_textures; const uint8_t *texels;
    const rt_texmap *material_maps;                   /* 2 per material (diffuse, specular) or NULL */
    const rt_texmap *env_map, *bg_map;                /* NULL = none */
    const rt_photon *caustic;   uint32_t n_caustic;   /* the second map (P13's causticmap): balanced, [0] unused */
} orc_scene;

typedef struct orc_hit {
    float z; float p[3]; float N[3];
    int32_t node; int32_t front;
    float uvw[3];
} orc_hit;

typedef struct orc_counters {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
    uint64_t box_tests, tri_tests, node_visits;
    uint64_t photon_queries, photons_visited;
} orc_counters;

/* 1: shade_fin traces and shades FIN's HEMISPHERE_SAMPLE rays at primary hits like the reference does, and
 * discards them like the reference does (FIN/main.cpp:642-693) -- same pixels, the reference's cost */
void  orc_set_trace_discarded(int on);
uint64_t orc_discarded_rays(void);
void  orc_counters_reset(void);
void  orc_counters_get(orc_counters *out);

/* context of the counter RNG: which sample / ray-tree node the next Shade call belongs to */
void  orc_set_rng(uint32_t seed, uint32_t sample, uint32_t node);
float orc_halton(int index, int base);
void  orc_color24(const float rgb[3], uint8_t out[3]);

/* primitives in object space; hit->z carries the current closest distance in/out */
int   orc_sphere_intersect(int model, const float ray[6], orc_hit *hit);
int   orc_plane_intersect(int model, const float ray[6], orc_hit *hit);
int   orc_box_intersect(const float box[6], const float ray[6], float t_max);
int   orc_mesh_intersect(int model, const orc_mesh *m, const float ray[6], orc_hit *hit);

void  orc_to_node_coords(const rt_node *n, const float ray[6], float out[6]);
void  orc_from_node_coords(const rt_node *n, orc_hit *hit);

/* TraceNode(rootNode, ray, hit); returns 1 on hit (hit initialised by the callee) */
int   orc_trace(const orc_scene *s, int model, const float ray[6], orc_hit *hit);
float orc_shadow(const orc_scene *s, int model, const float ray[6], float t_max);
void  orc_light_direction(const rt_light *l, const float p[3], float out[3]);   /* Light::Direction */
void  orc_illuminate(const orc_scene *s, const rt_params *P, const rt_light *l,
                     const float p[3], const float N[3], float out[3]);
void  orc_shade(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h,
                int bounce, float out[3]);

/* textures */
void  orc_texture_sample(const rt_texture *t, const uint8_t *texels, const float uvw[3], float rgb[3]);
void  orc_texmap_transform(const rt_texmap *m, const float uvw[3], float out[3]);
void  orc_environment_coord(const float dir[3], float uvw[3]);
/* TexturedColor::Sample(uvw): color, or color * map sample */
void  orc_textured_color(const orc_scene *s, const float color[3], const rt_texmap *map, const float uvw[3], float out[3]);

/* photon map */
void  orc_photon_pack(const float pos[3], const float dir[3], const float power[3], rt_photon *out);
void  orc_photon_direction(const rt_photon *p, float dir[3]);
void  orc_photon_power(const rt_photon *p, float rgb[3]);
void  orc_photon_balance(rt_photon *in, uint32_t n, rt_photon *out);
void  orc_estimate_irradiance(const rt_photon *photons, uint32_t n, int k, float radius,
                              const float pos[3], const float normal[3],
                              float irr[3], float dir[3]);

/* test hooks (see rt_oracle.c): while a script is active, random draws consume raw rand() values
 * captured from the reference run and orc_shadow logs {p, dir, t_max} and returns scripted values */
void  orc_script_begin(const int32_t *raw_rand, int n_rand, const float *shadow_values, int n_shadow,
                       float *shadow_log, int log_cap);
int   orc_script_end(int *rand_used);
int   orc_random_photon_bounce(const rt_blinn *m, const orc_hit *h, float ray[6], float c[3]);
void  orc_attenuation(const float absorption[3], float l, float out[3]);
void  orc_coordinate_system(const float N[3], float Nt[3], float Nb[3]);
/* RenderImage::ComputeZBufferImage / ComputeSampleCountImage (scene.h:591-637) */
void  orc_zbuffer_image(const float *zbuffer, int width, int height, uint8_t *zbufferImg);
int   orc_sample_count_image(const uint8_t *sampleCount, int width, int height, uint8_t *sampleCountImg);

/* generatePhotonMap with the build's counter RNG (Philox); out is 1-based with room for
 * max_photons + 9 records; returns the photon count */
uint32_t orc_photon_pass(const orc_scene *s, uint32_t seed, uint32_t max_photons, int max_bounce,
                         rt_photon *out, uint64_t *attempts_out);

uint32_t orc_caustic_pass(const orc_scene *s, uint32_t seed, uint32_t max_diffuse_hits, int max_bounce,
                          rt_photon *out, uint64_t *attempts_out);

/* cyBVH build (MeanSplit), returns node count incl. unused node 0 */
int   orc_bvh_build(const float *v, const uint32_t *f, int32_t nf, int32_t max_per_leaf,
                    rt_bvh_node *nodes_out, uint32_t *elements_out);

/* primary ray of sample j of pixel (x,y) as RenderPixel builds it */
void  orc_primary_ray(const rt_camera *cam, int x, int y, int j, float ray[6]);
/* RenderPixel over the pixel rectangle [x0,x1) x [y0,y1); outputs are full-image buffers */
void  orc_render(const orc_scene *s, const rt_camera *cam, const rt_params *P,
                 int x0, int y0, int x1, int y1,
                 uint8_t *rgb8, float *z, uint8_t *count);
/* per-sample linear colour of one pixel (before averaging); returns number of hit samples.
 * rgb has room for max_sample*3 floats, hitmask one byte per sample index. */
int   orc_pixel_samples(const orc_scene *s, const rt_camera *cam, const rt_params *P,
                        int x, int y, int j0, int j1, float *rgb, uint8_t *hitmask, float *z);

#ifdef __cplusplus
}
#endif
#endif
