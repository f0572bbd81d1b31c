// rt_dev.h -- device-side data layout of a lowered scene, shared by rt_api.cpp (which builds
// and uploads it) and rt_kernels.hip (which reads it).  gfx950 only.
//
// HBM layout (all arrays are plain hipMalloc allocations owned by the rt_scene):
//   nodes      DevNodeXf[n_nodes]      96 B each: itm(9) pos(3) tm(9) -- read with scalar loads
//   objects    DevObject[n_objects]    one per node that carries an Object, in TraceNode order;
//                                      `chain` = ancestor node indices root..self (every level's
//                                      ToNodeCoords is applied in turn, exactly like the
//                                      recursion in FIN/main.cpp:108-130)
//   per mesh   DevBvhNode[ ]           128 B: four children's boxes + their refs (one visit = one 128-byte read
//                                      for two levels of a binary tree); built from the triangles (surface-area heuristic)
//              DevTri[ ]               48 B: A, B, C, unit face normal -- in LEAF order, so a
//                                      leaf's triangles are contiguous
//              nrm[ ]                  36 B per leaf-ordered triangle: its three vertex normals (read once per ray, on the
//                                      final hit: indexed by the triangle's slot, no face-id hop in between)
//   photons    slots in sub-leaf-major order, RT_SUB_PHOTONS (16) slots per SUB-LEAF (padded with +inf
//              positions), pa = (pos.xyz, dir.x), pb = (dir.y, dir.z, maxPower, colour bytes); sbox =
//              tight box of every sub-leaf; RT_LEAF_SUBS (8) consecutive sub-leaves (128 slots, three
//              median splits apart) form a LEAF of the tree the queries walk: tbox = heap-ordered
//              boxes of the complete binary tree over the leaves (root = 1, leaves at [n_leaves,
//              2 n_leaves)).  Built on the GPU (rt_photon_build.hip); tbox and sbox are the head and the
//              last level of ONE array of the boxes of all heap nodes.
#ifndef RT_DEV_H
#define RT_DEV_H

#include <stdint.h>
#include "../../include/rt_mi355x.h"

#define RT_MAX_DEPTH      8      // scene-graph nesting supported on the device
#define RT_MAX_OBJECTS    4096
#define RT_BVH_STACK      32     // per-lane traversal stack entries (LDS) of the per-level kernels; k_wavefront keeps 24 (rt_kernels.hip)
#define RT_BLOCK          256    // threads per workgroup of the trace/shade kernels
#ifndef RT_SUB_PHOTONS
#define RT_SUB_PHOTONS    16     // photon slots per sub-leaf (16 or 32): a wavefront examines 64 / RT_SUB_PHOTONS sub-leaves per step
#endif
#define RT_LEAF_SUBS      (128 / RT_SUB_PHOTONS)     // sub-leaves per leaf of the walked tree (128 photon slots)
#define RT_GATHER_WAVES   4      // waves per gather workgroup
#ifndef RT_LEAFLIST_CAP
#define RT_LEAFLIST_CAP   40     // leaf ids kept per query in LDS before the slow path (40 leaves = 5120 slots)
#endif

struct DevNodeXf { float itm[9]; float pos[3]; float tm[9]; float pad[3]; };

// 128 bytes, read through the scalar cache by the wave-uniform object loop of trace(): everything an iteration needs of the
// object sits in this ONE record -- bounds, kind, the chain of its ancestors and a copy of its OWN node's ToNodeCoords
// transform -- so the loads of an iteration are issued together and waited for once (the loop used to chase
// object -> chain entry -> node transform through three dependent scalar loads per object and ray batch).
struct DevObject {
    // conservative bounds of the object in the root node's coordinates (its local extent taken through
    // FromNodeCoords of every ancestor below the root, inflated): a ray that misses them, or enters them
    // beyond its closest hit so far, cannot be given a hit by the exact local-space test, so the
    // object's transforms are skipped
    float wlo[3], whi[3];
    int32_t type, mesh;
    int32_t chain_len, material, node, pad;
    int32_t chain[RT_MAX_DEPTH];       // root .. self
    float own_itm[9], own_pos[3];      // DevNodeXf::itm / pos of chain[chain_len - 1]
};
static_assert(sizeof(DevObject) == 128, "one object = two 64-byte scalar loads");

// What a closest hit needs of its object once the loop over the objects is over: Node::FromNodeCoords (scene.h:509-513) of the
// object's own node and of up to two ancestors below the root, INLINE -- every address depends on the object index alone, so
// a lane's loads go out together (chasing object -> chain entry -> node transform level by level was seven dependent round
// trips per closest hit: 8 % of k_wavefront's wave time on the Cornell frame).  lvl[0] = the object's own node, lvl[1] = its
// parent, ...; the root (chain[0], shared by every object) comes over the scalar cache; deeper chains finish through DevObject::chain.
#define RT_BACK_LEVELS 3
struct DevObjectBack {
    int32_t chain_len, node, pad[2];
    DevNodeXf lvl[RT_BACK_LEVELS];
};
static_assert(sizeof(DevObjectBack) == 16 + 96 * RT_BACK_LEVELS, "layout");

// 128 bytes: the boxes of up to four children, one axis after the other (lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4]: six
// 16-byte loads), and their refs.  Built on the host from the mesh's triangles (rt_api.cpp: binned surface-area heuristic,
// leaves of at most four triangles, collapsed four-wide by the largest child first).  An unused child has NaN bounds (never
// entered).  One visit = ONE dependent read for two levels of a binary tree.
struct DevBvhNode {
    float lo[3][4], hi[3][4];
    uint32_t c[4];               // bit31: leaf (bits28-30 count-1, bits0-27 first triangle slot); else the index of a DevBvhNode
    uint32_t pad[4];
};
static_assert(sizeof(DevBvhNode) == 128, "one visit = one 128-byte record");
// A thread's BVH traversal stack: RT_BVH_LDS entries per lane in LDS in k_wavefront (the per-level kernels keep RT_BVH_STACK) and
// RT_BVH_SPILL more per thread in HBM behind them (DevScene::bvh_spill; allocated for scenes whose trees can need more than the LDS
// part, touched only by a traversal that is that deep).  Measured on MI355X (frame ms Cornell / 102 k triangles / C3, LDS entries
// and the ray stack the freed LDS buys, see WfCfg): 24 + 592 rays: 38.5 / 22.9 / 196.7; 12 + 840: 37.0 / 22.5 / 185.1; 8 + 928: 37.0 /
// 22.5 / 185.1; 4 + 1020: 37.1 / 23.1 / 186.7; 2 + 1064: 37.1 / 23.8 / 193.3 -- the LDS is worth more as ray stack (rays that do not
// fit take the global queue and a second pass) than as traversal stack (a deep traversal's pops come from L2 instead).
#ifndef RT_BVH_LDS
#define RT_BVH_LDS 8
#endif
#ifndef RT_BVH_SPILL
#define RT_BVH_SPILL 32
#endif
#define RT_SPILL_BLOCKS 1280     // workgroups the spill buffer is sized for (every tracing launch stays within it)

struct DevTri { float A[3], B[3], C[3], N[3]; };   // 48 bytes

struct DevMesh {
    const DevBvhNode *nodes;
    const DevTri     *tris;
    const float      *nrm;       // 9 floats per triangle slot (leaf order, like tris): vn[fn0], vn[fn1], vn[fn2] of its face
    const float      *tex;       // 9 floats per triangle slot: vt[ft0], vt[ft1], vt[ft2]; NULL without texture vertices
    float    root_box[6];
    uint32_t root_ref;           // child-ref encoding of the root (leaf or node index)
    uint32_t n_tris;
};

struct DevPhotonMap {
    const float4 *pa;            // [(n_sub + 1) * RT_SUB_PHOTONS]  position.xyz, direction.x (the last sub-leaf: all slots empty)
    const float4 *pb;            //             direction.yz, GetMaxPower(), colour bytes r|g<<8|b<<16 (as uint bits)
    const float4 *tbox;          // [2*n_leaves][2]: (lo.xyz, -), (hi.xyz, -)
    const float4 *sbox;          // [n_sub][2], n_sub = RT_LEAF_SUBS * n_leaves: sub-leaf j of leaf l is RT_LEAF_SUBS*l + j
    uint32_t n_leaves;           // power of two, 0 = no photon map
    uint32_t n_photons;          // photons stored in the leaves
    // coarse density grid (photon count per cubic cell of side `cell`) used only to pick the first
    // trial radius of a query: any radius gives the exact answer, a good one saves work
    const uint32_t *grid;
    float grid_min[3]; float cell, inv_cell;
    int32_t grid_dim[3];
    // per grid cell: the deepest node of the leaf-box tree (even depth, internal) whose subtree alone can hold photons within
    // `start_radius` of ANY point of the cell -- every sibling subtree on the way down lies farther from the cell than that.
    // A query inside the grid with a radius <= start_radius starts its walk there instead of at the root (the levels above
    // are a chain of dependent box reads with one survivor each).  NULL / 0: start at the root.  Rebuilt when the gather
    // radius grows (rt_api.cpp).
    const uint32_t *cell_start;
    float start_radius;
};

struct DevScene {
    const DevNodeXf *nodes;
    const DevObject *objects;
    const DevObjectBack *objects_back;   // [n_objects], see DevObjectBack
    const DevMesh   *meshes;
    const rt_blinn  *materials;
    const rt_light  *lights;
    const int32_t   *node_material;   // material index per node
    int32_t n_nodes, n_objects, n_meshes, n_materials, n_lights;
    float env[3], bg[3];
    DevPhotonMap pm;
    DevPhotonMap cm;             // caustic map (P13-family models, rt_params.caustic_k > 0)
    // textures: all NULL / RT_MAP_NONE when the scene has none (then no uvw is computed either)
    const rt_texture *textures; const uint8_t *texels; int32_t n_textures;
    const rt_texmap *material_maps;          // 2 per material, or NULL
    rt_texmap env_map, bg_map;
    int32_t use_uvw;
    int32_t max_bvh_depth;       // traversal-stack entries the deepest mesh BVH can need (binary tree: its depth; four-wide: 3 per level): which tracer kernels' stacks it fits
    uint32_t *bvh_spill;         // [RT_SPILL_BLOCKS * RT_BLOCK][RT_BVH_SPILL]: stack entries beyond a kernel's LDS stack; set per launch (NULL: the LDS stack always suffices)
    int32_t stochastic;          // some light has a size or some material a glossy reflection/refraction: shading draws random numbers
};

// camera set-up computed once on the host the way RenderPixel does it per thread
// (FIN/main.cpp:205-224)
struct DevCamera {
    float pos[3];
    float m[9];          // Matrix3(x_new, up, z_new), column-major
    float b[3];          // pixel (0,0) sample origin incl. the half-pixel shift
    float u, v;          // pixel pitch
    float dof;           // lens radius (0 = pinhole)
    int32_t width, height;
};

// tile walk of one rt_render_* call
// Unsigned division by a launch-constant divisor without the ~30-instruction generic sequence
// (Granlund & Montgomery 1994, "round-up" form, exact for every 32-bit dividend):
//   m = floor(2^32 * (2^L - d) / d) + 1, L = ceil(log2 d);  t = mulhi(x, m);  q = (t + ((x - t) >> 1)) >> (L - 1)
struct FastDiv { uint32_t d, m, sh; };
static inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f; f.d = d; f.m = 0; f.sh = 0;
    if (d <= 1) return f;
    uint32_t L = 0;
    while ((1ull << L) < (unsigned long long)d) L++;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - (unsigned long long)d)) / (unsigned long long)d + 1ull);
    f.sh = L - 1;
    return f;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t fastdiv(uint32_t x, const FastDiv &f)
{
    if (f.d <= 1) return x;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(x, f.m);
#else
    const uint32_t t = (uint32_t)(((unsigned long long)x * (unsigned long long)f.m) >> 32);
#endif
    return (((x - t) >> 1) + t) >> f.sh;
}

struct DevTiles {
    int32_t tile_w, tile_h, first, stride;
    int32_t tiles_x, tiles_total;
    int32_t n_tiles;          // tiles owned by this call
    FastDiv div_per, div_tile_w, div_tiles_x;     // by tile_w*tile_h, tile_w, tiles_x (tiles_prepare)
};
static inline void tiles_prepare(DevTiles &t)
{
    t.div_per = fastdiv_make((uint32_t)(t.tile_w * t.tile_h));
    t.div_tile_w = fastdiv_make((uint32_t)t.tile_w);
    t.div_tiles_x = fastdiv_make((uint32_t)t.tiles_x);
}

// device-side statistics block (uint64 counters, see rt_stats).  Counter i lives at stats[ST_AT(i)]: RT_STAT_STRIDE uint64s (4 KB +
// 256 B) apart, so that the atomics on different counters go to different memory channels instead of queueing on one line (the
// note at RT_CTR_STRIDE below)
#ifndef RT_STAT_STRIDE
#define RT_STAT_STRIDE 544
#endif
#define ST_AT(i) ((size_t)(i) * RT_STAT_STRIDE)
enum {
    ST_RAYS_PRIMARY = 0, ST_RAYS_SHADOW, ST_RAYS_REFLECT, ST_RAYS_REFRACT,
    ST_INSTANCE_VISITS, ST_BVH_NODES, ST_TRIS, ST_PHOTON_QUERIES, ST_PHOTONS_VISITED,
    ST_QUEUE_OVERFLOW, ST_GATHER_ROUNDS, ST_GATHER_SLOW, ST_GATHER_LEAF_READS,
    ST_PEAK_RAYS, ST_PEAK_QUERIES,      // largest ray-queue level / photon-query count any chunk produced (atomicMax)
    ST_COUNT
};

// ray queue: structure of arrays of float4 (one 16-byte coalesced read per lane per array)
//   a = (o.xyz, d.x)   b = (d.yz, thr.r, thr.g)   c = (thr.b, absorb.rgb)   d = (slot, bounce|kind<<8, -, -) as uint bits
//   e = (side.dir.xyz, side.K.r), with side.K.gb in c.zw: a ray to be spawned when this one HITS (P6 only)
struct DevRayQueue { float4 *a, *b, *c; uint4 *d; float4 *e; uint32_t cap; };
// photon query queue: qa = (pos.xyz, N.x)  qb = (N.yz, w.r, w.g)  qc = (w.b, slot bits, -, -)
struct DevPhotonQueue { float4 *qa, *qb, *qc; uint32_t cap; };

// per-chunk working set
struct DevWork {
    float   *sample_rgb;      // [chunk_pixels*max_sample*3]
    float   *sample_z;        // [chunk_pixels*max_sample]
    uint8_t *sample_hit;      // [chunk_pixels*max_sample]
    DevRayQueue rq[2];
    DevPhotonQueue pq;
    DevPhotonQueue cq;        // queries against the caustic map (buffers exist only when it is in use)
    uint32_t *counts;         // see the CNT_* indices below
    uint32_t *pixel_list;     // pixels (chunk-local) that take the second sample batch
    unsigned long long *stats;
    uint32_t *bvh_spill;      // DevScene::bvh_spill of this working set's launches
};
// The hot ones -- work counters and queue tails that every wave of a launch adds to -- sit RT_CTR_STRIDE uint32s apart: device-scope
// atomics are executed at the memory side, one after the other per channel (about 15-20 ns each), so counters that share a line queue
// behind each other (r4: the end of a k_gather launch was 41 000 failing grabs on eight adjacent counters = 0.8 ms).  4 KB + 256 B
// steps to another channel whether the interleave is 256 B or 4 KB.
#ifndef RT_CTR_STRIDE
#define RT_CTR_STRIDE 1088
#endif
#define CNT_PHOTONQ      (16 + 0 * RT_CTR_STRIDE)
#define CNT_PRIMARY_NEXT (16 + 1 * RT_CTR_STRIDE)   // work counter of k_wavefront: batches of 256 primary samples handed out so far
#define CNT_CAUSTICQ     (16 + 2 * RT_CTR_STRIDE)   // caustic-map query count
#define CNT_WF2_NEXT     (16 + 3 * RT_CTR_STRIDE)   // work counter of k_wavefront's second pass (source: the overflow queue of the first)
// k_gather's work counters, each RT_CTR_STRIDE apart: [seg] = 32-query batches handed out of XCD segment seg (8), [8] = mask of the
// segments known to be used up
#define RT_GATHER_CTRS 9
#define CNT_GATHER_NEXT  (16 + 4 * RT_CTR_STRIDE)
#define CNT_GATHER_NEXT2 (CNT_GATHER_NEXT + RT_GATHER_CTRS * RT_CTR_STRIDE)    // the same for the caustic gather
#define CNT_RESET   (CNT_GATHER_NEXT2 + RT_GATHER_CTRS * RT_CTR_STRIDE)        // counters [0, CNT_RESET) are cleared before every pass
#define CNT_PIXLIST (CNT_RESET + 10)   // survives the passes of a chunk
#define CNT_TOTAL   (CNT_RESET + 12)

#endif
