// rt_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the render path.
//
//   k_wavefront K1+K2+K3+K4  the whole ray tree of a chunk in one persistent launch: primary-ray generation,
//                            closest-hit traversal, Blinn shade with an any-hit shadow ray per light, children
//                            pushed onto the workgroup's ray stack in LDS and popped 256 at a time (FIN, P13)
//   k_primary   K1+K2+K3+K4  the same for the primary rays only, children to the global SoA ray queues
//   k_bounce    K2+K3+K4     one level of the reflection/refraction ray tree from a global queue (the other
//                            shading models; rays k_wavefront could not keep in LDS)
//   k_gather    K5           k-nearest photon gather, one query per wavefront step
//   k_resolve   K6           per-pixel average / variance gate / gamma / Color24 pack
//   k_trace     K2           closest-hit only (parity entry point rt_trace_rays)
//
// The arithmetic of every device function follows the reference operation by operation (same
// expression order, thresholds and quirks; file:line cited per function, FIN = /root/reference/
// RayTracingFinal/RayTracingFinal) and the file is compiled with -ffp-contract=off, so hit
// records agree with the CPU oracle bit for bit and colours to the last ulp of powf/expf.
// What differs is control: the recursion of TraceNode/TraceBVHNode/Shade becomes loops, ray
// queues and a per-lane LDS stack; zero-weight subtrees are not traced; the shadow ray is an
// any-hit query; BVH boxes are culled against the closest hit so far.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include "rt_dev.h"

#define BIGFLOAT 1.0e30f
#define LEAF_BIT 0x80000000u
// k_gather tuning knobs (the defaults are the measured best on MI355X)
#ifndef RT_GATHER_GUESS
#define RT_GATHER_GUESS 1.2f   // photons expected inside the first trial radius, in units of k (round 2, sub-leaves: 1.1: 42.0 ms, 1.15: 40.8, 1.2: 40.7, 1.3: 41.7, 1.45: 43.3)
#endif
#ifndef RT_GATHER_RING
#define RT_GATHER_RING 128     // LDS entries for the photons around the predicted k-th distance (0: always re-read in pass 2)
#endif
#ifndef RT_GATHER_BAND_LO
#define RT_GATHER_BAND_LO 0.92f   // the ring keeps the photons between BAND_LO and BAND_HI times the predicted k-th squared distance
                                  // (0.85-1.18: 40.7 ms, 0.88-1.15: 39.7, 0.92-1.10: 39.4, 0.94-1.08: 39.3)
#endif
#ifndef RT_GATHER_BAND_HI
#define RT_GATHER_BAND_HI 1.10f
#endif
#ifndef RT_GATHER_BATCH
#define RT_GATHER_BATCH 32     // queries a wave lists per phase A (40 leaf ids each: lists + ring keep 5 waves/SIMD)
#endif
#ifndef RT_GATHER_CELL_GUESS
#define RT_GATHER_CELL_GUESS RT_GATHER_GUESS    // first trial radius^2 of a query whose cell remembers a k-th distance: that distance times this
#endif
// the photon loop of k_gather may contract mul+add into fma (d^2, dir.N, box distances, weighted sums; gate there: 2e-5); the rest of the
// file stays uncontracted: hit records are bit-exact
#define RT_FP_CONTRACT _Pragma("clang fp contract(fast)")
#ifndef RT_WF_GRAB
#define RT_WF_GRAB 2           // rounds of 256 primary samples (or queued rays) a workgroup of k_wavefront takes per atomic on the work counter
#endif
#ifndef RT_WF_PERWAVE
#define RT_WF_PERWAVE 1        // 1: for the P12 model every wave of k_wavefront runs its own rounds on its own part of the LDS ray stack (no workgroup barriers); 0: the four waves always share stack and rounds
#endif
#define RT_SUBS_PER_STEP (64 / RT_SUB_PHOTONS)                  // sub-leaves a wavefront examines per step
#define RT_SUBLIST_CAP (RT_LEAFLIST_CAP * RT_LEAF_SUBS)         // sub-leaf ids of one query

// ------------------------------------------------------------------------------------------------
// float3 algebra in the reference's evaluation order (cyPoint.h:259-350, cyMatrix.h:542-546)
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 p) { return mk(a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x); }
__device__ __forceinline__ float len2(V3 a) { return dot(a, a); }
__device__ __forceinline__ V3 normalize(V3 a) { return a / sqrtf(len2(a)); }
__device__ __forceinline__ V3 mmul(const float *d, V3 p)
{
    return mk(p.x * d[0] + p.y * d[3] + p.z * d[6], p.x * d[1] + p.y * d[4] + p.z * d[7], p.x * d[2] + p.y * d[5] + p.z * d[8]);
}
__device__ __forceinline__ V3 mtmul(const float *d, V3 v)      // TransposeMult, scene.h:254-261
{
    return mk(dot(mk(d[0], d[1], d[2]), v), dot(mk(d[3], d[4], d[5]), v), dot(mk(d[6], d[7], d[8]), v));
}
__device__ __forceinline__ float gray(V3 c) { return (c.x + c.y + c.z) / 3.0f; }      // Color::Gray, cyColor.h
#define RMAX(a, b) ((a) > (b) ? (a) : (b))     // the reference's macros, scene.h:48-54
#define RMIN(a, b) ((a) < (b) ? (a) : (b))

struct Counters { uint32_t inst, nodes, tris, shadow; uint32_t occ; };     // occ: 1 + the object that occluded this lane's last shadow ray (0: none)

// Halton, FIN/include/scene.h:131-140
__device__ __forceinline__ float halton(int index, int base)
{
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) { r += f * (i % base); f /= (float)base; }
    return r;
}


struct Hit { float z; V3 p, N; int node; int front; V3 uvw; };

// ------------------------------------------------------------------------------------------------
// primitives (object space)
// ------------------------------------------------------------------------------------------------
// Sphere::IntersectRay, FIN/include/objects.h:24-70
__device__ __forceinline__ bool sphere_hit(V3 rp, V3 rd, float &z, V3 &hp, V3 &hN, int &front)
{
    const float a = dot(rd, rd);
    const float c = dot(rp, rp) - 1;
    const float b = 2 * dot(rp, rd);
    const float insqrt = b * b - (4 * a * c);
    const float zero = 0.001f;
    if (insqrt >= zero) {
        const float sq = sqrtf(insqrt);
        const float t1 = (-b + sq) / (a * 2);
        const float t2 = (-b - sq) / (a * 2);
        const float prez = z;
        if (t2 >= prez) return false;
        if (t1 > zero && t2 < zero && t1 < prez) {
            z = t1; front = 0;
            hp = rd * z + rp; hN = normalize(hp);
            return true;
        } else if (t1 > zero && t2 > zero && t2 < prez) {
            z = t2; front = 1;
            hp = rd * z + rp; hN = normalize(hp);
            return true;
        }
    }
    return false;
}

// Sphere::IntersectRay of RayTracingProj3 (main.cpp:192-221): no bias, z = min(t1,t2), rejected when negative
// or not closer; N = p un-normalised; front untouched
__device__ __forceinline__ bool sphere_hit_p3(V3 rp, V3 rd, float &z, V3 &hp, V3 &hN)
{
    const float a = dot(rd, rd);
    const float c = dot(rp, rp) - 1;
    const float b = 2 * dot(rp, rd);
    const float insqrt = b * b - (4 * a * c);
    if (insqrt >= 0) {
        const float sq = sqrtf(insqrt);
        const float t1 = (-b + sq) / (a * 2);
        const float t2 = (-b - sq) / (a * 2);
        const float zz = RMIN(t1, t2);
        if (zz < 0 || zz >= z) return false;
        z = zz;
        hp = rd * z + rp; hN = hp;
        return true;
    }
    return false;
}

// Plane::IntersectRay, FIN/include/objects.h:84-111 (P13 flips `front`, P13/include/objects.h:98-101)
__device__ __forceinline__ bool plane_hit(int model, V3 P, V3 d, float &z, V3 &hp, V3 &hN, int &front)
{
    const float zero = 0.001f;
    const float t = -(P.z / d.z);
    if (t >= zero && t < BIGFLOAT && t < z) {
        const V3 Hitp = P + d * t;
        if (Hitp.x >= -1 && Hitp.x <= 1 && Hitp.y >= -1 && Hitp.y <= 1) {
            z = t; hp = Hitp; hN = mk(0, 0, 1);
            const float nd = dot(mk(0, 0, 1), d);
            front = (model != RT_SHADE_FIN) ? ((nd < 0.0f) ? 0 : 1) : ((nd <= 0.0f) ? 1 : 0);
            return true;
        }
    }
    return false;
}

// Slab test against a node box.  Box::IntersectRay (FIN/scene.cpp:11-65) is only conservative
// (accepts boxes behind the ray, never culls by distance); this one culls against the closest
// hit so far and is padded by 2e-6 relative so that reciprocal rounding never rejects a box
// whose triangle the exact test would accept.  Returns the entry distance, or BIGFLOAT*2.
__device__ __forceinline__ float box_entry(const float *lo, const float *hi, V3 o, V3 inv, float zbest)
{
    const float t0x = (lo[0] - o.x) * inv.x, t1x = (hi[0] - o.x) * inv.x;
    const float t0y = (lo[1] - o.y) * inv.y, t1y = (hi[1] - o.y) * inv.y;
    const float t0z = (lo[2] - o.z) * inv.z, t1z = (hi[2] - o.z) * inv.z;
    // fminf/fmaxf drop NaN operands (0*inf when the origin lies on a slab plane and dir == 0)
    const float tenter = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    const float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    const bool hit = (texit >= 0.0f) && (tenter <= texit * 1.000002f) && (tenter * 0.999998f <= zbest);
    return hit ? tenter : 3.0e30f;
}

// TriObj::TriangleArea, FIN/include/objects.h:146-157
__device__ __forceinline__ float tri_area(int i, V3 A, V3 B, V3 C)
{
    if (i == 0) return (B.y - A.y) * (C.z - A.z) - (C.y - A.y) * (B.z - A.z);
    if (i == 1) return (B.x - A.x) * (C.z - A.z) - (C.x - A.x) * (B.z - A.z);
    return (B.x - A.x) * (C.y - A.y) - (C.x - A.x) * (B.y - A.y);
}

// TriObj::IntersectTriangle, FIN/include/objects.h:226-267.  N (unit face normal) is precomputed
// with the same expression on the host.  Returns barycentrics; normal interpolation is deferred.
__device__ __forceinline__ bool tri_hit_fin(const DevTri &T, V3 rp, V3 rd, float &z, V3 &hp, V3 &bc, int &front)
{
    const V3 A = ld3(T.A), B = ld3(T.B), C = ld3(T.C), N = ld3(T.N);
    const float dz = dot(rd, N);
    if (fabsf(dz) < 1e-7f) return false;
    const float pz = dot(rp - A, N);
    const float t = -pz / dz;
    if (t <= 0.001f) return false;
    if (t < z) {
        const V3 p = rp + rd * t;
        int ign;
        const float ax = fabsf(N.x), ay = fabsf(N.y), az = fabsf(N.z);
        if (ax > ay && ax > az) ign = 0; else if (ay > az) ign = 1; else ign = 2;
        const float s = 1.f / tri_area(ign, A, B, C);
        const float a = tri_area(ign, p, B, C) * s;
        const float b = tri_area(ign, p, C, A) * s;
        const float c = 1.f - a - b;
        if (a < 0 || b < 0 || c < 0) return false;
        z = t; hp = p; bc = mk(a, b, c); front = (dz <= 0) ? 1 : 0;
        return true;
    }
    return false;
}

// TriObj::IntersectTriangle, P13/include/objects.h:148-206 (back-face culled, bias 1e-7)
__device__ __forceinline__ bool tri_hit_p13(const DevTri &T, V3 rp, V3 rd, float &z, V3 &hp, V3 &bc, int &front)
{
    const float bias = 1e-7f;
    const V3 A = ld3(T.A), B = ld3(T.B), C = ld3(T.C), tN = ld3(T.N);
    if (dot(tN, rp - A) < bias) return false;
    const float den = dot(tN, rd);
    if (den == 0) return false;
    const float t = dot(tN, C - rp) / den;
    if (t < bias || t >= z || t >= BIGFLOAT) return false;
    const float fx = fabsf(tN.x), fy = fabsf(tN.y), fz = fabsf(tN.z);
    const float maxN = fmaxf(fz, fmaxf(fx, fy));
    const V3 P = rp + rd * t;
    float pax, pay, pbx, pby, pcx, pcy, ppx, ppy;
    if (maxN == fx)      { pax = A.y; pay = A.z; pbx = B.y; pby = B.z; pcx = C.y; pcy = C.z; ppx = P.y; ppy = P.z; }
    else if (maxN == fy) { pax = A.x; pay = A.z; pbx = B.x; pby = B.z; pcx = C.x; pcy = C.z; ppx = P.x; ppy = P.z; }
    else                 { pax = A.x; pay = A.y; pbx = B.x; pby = B.y; pcx = C.x; pcy = C.y; ppx = P.x; ppy = P.y; }
    const float area_tri = (pax - pcx) * (pby - pcy) - (pay - pcy) * (pbx - pcx);
    const float area_bcp = (ppx - pcx) * (pby - pcy) - (ppy - pcy) * (pbx - pcx);
    const float area_acp = (pax - pcx) * (ppy - pcy) - (pay - pcy) * (ppx - pcx);
    const float alpha = area_bcp / area_tri;
    const float beta = area_acp / area_tri;
    const float gam = (float)(1.0 - (double)alpha - (double)beta);
    if (alpha < -bias || beta < -bias || gam < -bias || alpha > 1.0f || beta > 1.0f || gam > 1.0f) return false;
    z = t; bc = mk(alpha, beta, gam); front = 1;
    hp = A * alpha + B * beta + C * gam;
    return true;
}

// Scene tables (objects, node transforms, lights) are written before a launch and only read by it, and the loops over
// them are wave-uniform: read through the constant address space such a table entry comes over the scalar data cache
// into SGPRs (s_load, merged to x4/x8/x16) -- one short-latency read per wave instead of a 64-lane vector load of one
// address, and no vector registers held for data every lane shares.  (With a lane-varying index the same code falls
// back to a vector load.)
template <class T> __device__ __forceinline__ T cld(const T *p) { return *(const __attribute__((address_space(4))) T *)(uintptr_t)p; }
__device__ __forceinline__ V3 cld3(const float *p) { return mk(cld(p), cld(p + 1), cld(p + 2)); }
struct M9 { float m[9]; };
__device__ __forceinline__ rt_light cld_light(const rt_light *p)
{
    rt_light l;
    l.type = cld(&p->type); l.size = cld(&p->size);
    for (int i = 0; i < 3; i++) { l.intensity[i] = cld(p->intensity + i); l.position[i] = cld(p->position + i); l.direction[i] = cld(p->direction + i); }
    return l;
}
__device__ __forceinline__ M9 cld9(const float *p) { M9 r; for (int i = 0; i < 9; i++) r.m[i] = cld(p + i); return r; }

// A thread's BVH traversal stack: `cap` entries in LDS (entry i of this thread at lds[i * RT_BLOCK]) and RT_BVH_SPILL more in
// HBM behind them (spill, this thread's own 64 bytes; NULL when the scene's trees cannot outgrow the LDS part -- the host
// checks DevScene::max_bvh_depth against cap + RT_BVH_SPILL).  The spill part is touched only by a traversal that is
// deeper than the LDS part: a four-wide node can leave three entries per level.
typedef __attribute__((address_space(3))) uint32_t lds_u32;      // the LDS part is addressed as LDS (ds_read / ds_write), never through flat instructions
struct BvhStack { lds_u32 *lds; uint32_t *spill; uint32_t cap; };
__device__ __forceinline__ BvhStack bvh_stack(uint32_t *lds_base, uint32_t cap, uint32_t *spill_base)
{
    BvhStack st;
    st.lds = (lds_u32 *)lds_base + threadIdx.x; st.cap = cap;
    st.spill = spill_base ? spill_base + ((size_t)blockIdx.x * RT_BLOCK + threadIdx.x) * RT_BVH_SPILL : nullptr;
    return st;
}
__device__ __forceinline__ void bvh_push(const BvhStack &st, uint32_t &sp, uint32_t v)
{
    if (sp < st.cap) st.lds[sp * RT_BLOCK] = v;
    else if (st.spill && sp < st.cap + RT_BVH_SPILL) st.spill[sp - st.cap] = v;
    sp++;
}
__device__ __forceinline__ uint32_t bvh_pop(const BvhStack &st, uint32_t &sp)
{
    --sp;
    if (sp < st.cap) return st.lds[sp * RT_BLOCK];
    return st.spill[sp - st.cap];
}

// TriObj::IntersectRay -> TraceBVHNode (FIN/include/objects.h:127-133, 271-302) as an iterative,
// near-first traversal with a per-lane stack in LDS.  ANY: stop at the first accepted triangle.
template <bool ANY, int MODEL>
__device__ bool mesh_hit(const DevMesh *Mp, V3 o, V3 d, float &z, V3 &hp, V3 &hN, int &front,
                         const BvhStack &stack, Counters &cnt, V3 *uvw = nullptr)
{
    // the mesh record is the same for every lane: over the scalar cache (see cld), so that the array bases live in SGPRs
    DevMesh M;
    M.nodes = cld(&Mp->nodes); M.tris = cld(&Mp->tris); M.nrm = cld(&Mp->nrm); M.tex = cld(&Mp->tex);
    for (int i = 0; i < 6; i++) M.root_box[i] = cld(Mp->root_box + i);
    M.root_ref = cld(&Mp->root_ref);
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (box_entry(M.root_box, M.root_box + 3, o, inv, z) > 2.0e30f) return false;
    uint32_t cur = M.root_ref;
    uint32_t sp = 0;
    bool any = false;
    uint32_t best_slot = 0;
    V3 bc = mk(0, 0, 0);
    // "while-while" traversal: every lane first walks down to its next leaf (inner loop), then all lanes
    // that hold a leaf test its triangles together.  With one loop that does either step per iteration a
    // single lane at a leaf makes the whole wave sit through the triangle code; for the reflected rays of
    // the 100 k-triangle scene that costs about half the bounce kernel's time.  Each lane still visits
    // the same nodes and triangles in the same order (near child first, far child on the stack).
    const uint32_t DONE = 0xFFFFFFFFu;                  // has LEAF_BIT set: ends the inner loop
    while (cur != DONE) {
        while (!(cur & LEAF_BIT)) {
            // one 128-byte record = the boxes of four children (two levels of the reference's tree): six 16-byte loads of
            // bounds, one of refs, issued together -- ONE dependent round trip where the binary tree takes two
            const float4 *np = (const float4 *)(M.nodes + cur);
            const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5];
            const uint4 cr = *(const uint4 *)(np + 6);
            cnt.nodes++;
            const float l0[3] = {lx.x, ly.x, lz.x}, h0[3] = {hx.x, hy.x, hz.x}, l1[3] = {lx.y, ly.y, lz.y}, h1[3] = {hx.y, hy.y, hz.y};
            const float l2[3] = {lx.z, ly.z, lz.z}, h2[3] = {hx.z, hy.z, hz.z}, l3[3] = {lx.w, ly.w, lz.w}, h3[3] = {hx.w, hy.w, hz.w};
            float e0 = box_entry(l0, h0, o, inv, z), e1 = box_entry(l1, h1, o, inv, z);
            float e2 = box_entry(l2, h2, o, inv, z), e3 = box_entry(l3, h3, o, inv, z);      // an unused child's bounds are NaN: never entered
            uint32_t c0 = cr.x, c1 = cr.y, c2 = cr.z, c3 = cr.w;
            // nearest first: sort the four (entry distance, ref) pairs (misses carry 3e30 and end up last); equal distances keep
            // the reference's child order
#define RT_CSWAP(ea, ca, eb, cb) do { const bool s_ = (eb) < (ea); const float te_ = s_ ? (eb) : (ea); const uint32_t tc_ = s_ ? (cb) : (ca); \
                                      (eb) = s_ ? (ea) : (eb); (cb) = s_ ? (ca) : (cb); (ea) = te_; (ca) = tc_; } while (0)
            RT_CSWAP(e0, c0, e1, c1); RT_CSWAP(e2, c2, e3, c3); RT_CSWAP(e0, c0, e2, c2); RT_CSWAP(e1, c1, e3, c3); RT_CSWAP(e1, c1, e2, c2);
#undef RT_CSWAP
            if (e3 < 2.0e30f) bvh_push(stack, sp, c3);
            if (e2 < 2.0e30f) bvh_push(stack, sp, c2);
            if (e1 < 2.0e30f) bvh_push(stack, sp, c1);
            cur = (e0 < 2.0e30f) ? c0 : (sp ? bvh_pop(stack, sp) : DONE);
        }
        if (cur == DONE) break;
        const uint32_t count = ((cur >> 28) & 7u) + 1;
        const uint32_t first = cur & 0x0FFFFFFFu;
        // the next triangle of the leaf is fetched while this one is tested (one exposed round trip per leaf, not per triangle)
        DevTri T = M.tris[first];
        for (uint32_t i = 0; i < count; i++) {
            DevTri Tn = T;
            if (i + 1 < count) Tn = M.tris[first + i + 1];
            cnt.tris++;
            const bool h = (MODEL != RT_SHADE_FIN) ? tri_hit_p13(T, o, d, z, hp, bc, front)
                                                   : tri_hit_fin(T, o, d, z, hp, bc, front);
            if (h) { any = true; best_slot = first + i; }
            T = Tn;
        }
        if (ANY && any) return true;
        cur = sp ? bvh_pop(stack, sp) : DONE;
    }
    if (!any) return false;
    // cyTriMesh::GetNormal = vn[fn0]*bc.x + vn[fn1]*bc.y + vn[fn2]*bc.z (cyTriMesh.h:167,191)
    const float *n9 = M.nrm + 9 * (size_t)best_slot;
    const V3 Ni = ld3(n9) * bc.x + ld3(n9 + 3) * bc.y + ld3(n9 + 6) * bc.z;
    hN = (MODEL != RT_SHADE_FIN) ? normalize(Ni) : Ni;     // FIN leaves it un-normalised (:262)
    // the PROJ13 triangle also sets uvw = GetTexCoord(face, bc) (P13/include/objects.h:203)
    if (MODEL != RT_SHADE_FIN && uvw && M.tex) {
        const float *t9 = M.tex + 9 * (size_t)best_slot;
        *uvw = ld3(t9) * bc.x + ld3(t9 + 3) * bc.y + ld3(t9 + 6) * bc.z;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// TraceNode(rootNode, ray, hit), FIN/main.cpp:108-130, flattened: for every node that carries an
// Object (in the recursion's visiting order) the ray is taken through ToNodeCoords of each
// ancestor in turn (scene.h:502-508; direction NOT renormalised, so t is shared by all spaces),
// and the closest hit is brought back through FromNodeCoords of each ancestor (scene.h:509-513).
// ------------------------------------------------------------------------------------------------
template <bool ANY, int MODEL, bool TEX = false>
__device__ bool trace(const DevScene &S, V3 o, V3 d, float zinit, Hit &h, const BvhStack &stack, Counters &cnt)
{
    float z = zinit;
    int best = -1, bfront = 1;
    V3 bp = mk(0, 0, 0), bN = mk(0, 0, 0);
    // HitInfo::uvw starts at (0.5,0.5,0.5) (scene.h:163); spheres and planes overwrite it whenever they
    // accept a hit (objects.h:49-51,103), FIN's triangles never do (:226-267) -- so a mesh hit keeps the
    // coordinate of whatever sphere/plane hit the ray had accepted before it.  Reproduced as is.
    // (The PROJ13 triangle does write it, from the mesh's texture vertices.)
    V3 uvw = mk(0.5f, 0.5f, 0.5f);
    // Node::ToNodeCoords (scene.h:502-508) of one level; the direction is the image of p+d minus the image of p
    auto to_node = [&](int node, V3 &lp, V3 &ldir) {
        const DevNodeXf *X = S.nodes + node;
        const V3 pos = cld3(X->pos);
        const M9 itm = cld9(X->itm);
        const V3 rp = mmul(itm.m, lp - pos);
        ldir = mmul(itm.m, (lp + ldir) - pos) - rp;
        lp = rp;
    };
    // Every chain starts at the root (node 0) and siblings share their parent: the ray in the root's and
    // in the last visited level-1 group's coordinates is kept instead of being recomputed per object
    // (same operations on the same inputs, so the results do not change).  The object loop is
    // wave-uniform, so the bookkeeping is scalar.
    V3 p0 = o, d0 = d;
    to_node(0, p0, d0);
    V3 p1 = p0, d1 = d0;
    int cached1 = -1;
    // reciprocal direction (in the root's coordinates) for the bounds cull only: approximate is fine, the
    // bounds are inflated
    const V3 winv = mk(__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y), __builtin_amdgcn_rcpf(d0.z));
    // any-hit queries: the object that occluded the last shadow ray of this wave is tried first (its neighbours' rays mostly end on
    // the same occluder, and the first accepted hit ends a lane's query: the order of the objects does not change the answer)
    int hint = -1;
    if (ANY) { hint = __builtin_amdgcn_readfirstlane((int)cnt.occ) - 1; if (hint >= S.n_objects) hint = -1; }
    for (int it = (hint >= 0 ? -1 : 0); it < S.n_objects; it++) {
        const int oi = it < 0 ? hint : it;
        if (it >= 0 && oi == hint) continue;
        const DevObject *obp = S.objects + oi;
        // Everything this iteration needs of the object comes from its one 128-byte record, fetched by two 64-byte scalar loads
        // issued back to back and waited for ONCE.  Written as inline assembly because the compiler, short of scalar registers
        // in these kernels, otherwise sinks every field's load to its first use: five dependent scalar-cache round trips per
        // object (bounds -> chain length -> group id -> chain entry -> node transform) in a loop three waves per SIMD cannot hide.
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        u32x16 r0, r1;
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(obp) : "memory");
        static_assert(offsetof(DevObject, type) == 24 && offsetof(DevObject, chain_len) == 32 && offsetof(DevObject, chain) == 48 && offsetof(DevObject, own_itm) == 80, "record layout the loads assume");
        const float wlo[3] = {__uint_as_float(r0[0]), __uint_as_float(r0[1]), __uint_as_float(r0[2])}, whi[3] = {__uint_as_float(r0[3]), __uint_as_float(r0[4]), __uint_as_float(r0[5])};
        const int ob_type = (int)r0[6], ob_mesh = (int)r0[7], ob_chain_len = (int)r0[8], n1 = (int)r0[13];
        M9 own_itm;
        own_itm.m[0] = __uint_as_float(r1[4]); own_itm.m[1] = __uint_as_float(r1[5]); own_itm.m[2] = __uint_as_float(r1[6]);
        own_itm.m[3] = __uint_as_float(r1[7]); own_itm.m[4] = __uint_as_float(r1[8]); own_itm.m[5] = __uint_as_float(r1[9]);
        own_itm.m[6] = __uint_as_float(r1[10]); own_itm.m[7] = __uint_as_float(r1[11]); own_itm.m[8] = __uint_as_float(r1[12]);
        const V3 own_pos = mk(__uint_as_float(r1[13]), __uint_as_float(r1[14]), __uint_as_float(r1[15]));
        // skip the object when no lane's ray can reach its bounds before that lane's closest hit so far
        if (!__any(box_entry(wlo, whi, p0, winv, z) < 2.0e30f)) continue;
        V3 lp = p0, ldir = d0;
        int c = 1;
        if (ob_chain_len > 2) {
            if (n1 != cached1) { p1 = p0; d1 = d0; to_node(n1, p1, d1); cached1 = n1; }
            lp = p1; ldir = d1; c = 2;
        }
        for (; c < ob_chain_len - 1; c++) to_node(cld(&obp->chain[c]), lp, ldir);     // levels between the group and the object: deeper scene graphs only
        if (ob_chain_len > 1) {
            // the object's own level, Node::ToNodeCoords with the copy of its transform in the record (same arithmetic as to_node)
            const V3 rp = mmul(own_itm.m, lp - own_pos);
            ldir = mmul(own_itm.m, (lp + ldir) - own_pos) - rp;
            lp = rp;
        }
        cnt.inst++;
        V3 hp, hN;
        int fr = 1;
        bool hit = false;
        if (ob_type == RT_OBJ_SPHERE) hit = (MODEL == RT_SHADE_P3) ? sphere_hit_p3(lp, ldir, z, hp, hN) : sphere_hit(lp, ldir, z, hp, hN, fr);
        else if (ob_type == RT_OBJ_PLANE) hit = plane_hit(MODEL, lp, ldir, z, hp, hN, fr);
        else if (ob_type == RT_OBJ_MESH) hit = mesh_hit<ANY, MODEL>(S.meshes + ob_mesh, lp, ldir, z, hp, hN, fr, stack, cnt, (TEX && S.use_uvw) ? &uvw : nullptr);
        if (hit) {
            if (ANY) { cnt.occ = (uint32_t)oi + 1u; return true; }
            best = oi; bp = hp; bN = hN; bfront = fr;
            if (TEX && S.use_uvw) {
                if (ob_type == RT_OBJ_SPHERE)               // objects.h:49-51, atan2/asin in double
                    uvw = mk((float)(0.5 - atan2((double)hp.x, (double)hp.y) / (2 * M_PI)), (float)(0.5 + asin((double)hp.z) / M_PI), 0);
                else if (ob_type == RT_OBJ_PLANE) uvw = mk((hp.x + 1) / 2, (hp.y + 1) / 2, 0);    // objects.h:103
            }
        }
    }
    if (best < 0) return false;
    h.uvw = uvw;
    // Node::FromNodeCoords (scene.h:509-513) level by level, the object's own node first.  The levels below the root come inline
    // with the object's DevObjectBack record (addresses depend on `best` alone: one round trip); the root's over the scalar cache.
    const DevObjectBack *ob = S.objects_back + best;
    const int len = ob->chain_len;
    auto from_node = [&](const DevNodeXf &X) {
        bp = mmul(X.tm, bp) + ld3(X.pos);
        bN = normalize(mtmul(X.itm, bN));
    };
    const int inl = min(len - 1, RT_BACK_LEVELS);            // levels held inline (all but the root, up to RT_BACK_LEVELS)
#pragma unroll
    for (int k = 0; k < RT_BACK_LEVELS; k++) if (k < inl) from_node(ob->lvl[k]);
    if (len - 1 > RT_BACK_LEVELS) {                          // a deeper chain: the levels between, through the object's chain
        const DevObject &o2 = S.objects[best];
        for (int c = len - 1 - RT_BACK_LEVELS; c >= 1; c--) from_node(S.nodes[o2.chain[c]]);
    }
    {
        const DevNodeXf *R = S.nodes;                        // chain[0] is the root for every object
        const M9 tm = cld9(R->tm), itm = cld9(R->itm);
        const V3 pos = cld3(R->pos);
        bp = mmul(tm.m, bp) + pos;
        bN = normalize(mtmul(itm.m, bN));
    }
    h.z = z; h.p = bp; h.N = bN; h.node = ob->node; h.front = bfront;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Counter-based random numbers for the stochastic effects.  The reference calls
// libc rand() from all its threads at once, so its sequences are not even reproducible; here every
// draw is Philox-4x32-10 keyed by the seed and counted by WHAT it is for:
//   (sample id = (y*W+x)*max_sample+j, ray-tree node, purpose, index)
// so a pixel's value does not depend on tiling, chunking, scheduling or the number of GPUs, and the
// CPU oracle draws the very same numbers.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
#define RNG_LENS   1u     // DoF lens table entry (sample id = pixel id)
#define RNG_PICK   2u     // which lens entry a sample uses
#define RNG_SHADOW 3u     // area-light sample (index = light*64 + i, refinement batch at +32)
#define RNG_GLOSSR 4u     // glossy reflection jitter
#define RNG_GLOSST 5u     // glossy refraction jitter
#define RNG_GI     6u     // hemisphere sample
struct RngCtx { uint32_t seed, sample, node; };
// two uniforms in [0,1): each stands in for one rand() / (float) RAND_MAX
__device__ __forceinline__ void rng2(const RngCtx &c, uint32_t purpose, uint32_t index, float &u0, float &u1)
{
    uint32_t o[4];
    philox4x32(c.seed, 0x52544D49u, c.sample, c.node, (purpose << 24) | (index & 0xFFFFFFu), 0x5eed5eedu, o);
    u0 = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
    u1 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
}
// id of a child node in the ray tree (1 = primary ray): only has to be reproducible
__host__ __device__ inline uint32_t child_node(uint32_t node, uint32_t kind) { return node * 0x9E3779B1u + kind * 0x85EBCA6Bu + 0x27D4EB2Fu; }

// ------------------------------------------------------------------------------------------------
// the lanes where a condition holds: the builtin keeps the condition a lane mask (one s_and with exec); HIP's
// __ballot(int) round-trips it through a 0/1 register (v_cndmask + v_cmp_ne per call, on the VALU the gather is bound by)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// how many lanes of a mask lie below this one: v_mbcnt_lo + v_mbcnt_hi
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// queues: wave-aggregated append (one atomic per wave, __ballot + popcount for the lane offset)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_push(bool pred, uint32_t *counter)
{
    const unsigned long long mask = ballot64(pred);
    if (mask == 0) return 0xFFFFFFFFu;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    const uint32_t off = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    return pred ? base + off : 0xFFFFFFFFu;
}

#define KIND_REFLECT 0u
#define KIND_REFRACT 1u
#define KIND_GI      2u

// wave-wide maximum of a small non-negative int (loop bound for wave-collective pushes)
__device__ __forceinline__ int __reduce_max_sync_compat(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

struct PathIn { V3 o, d, thr, absorb; uint32_t slot; int bounce; uint32_t kind; bool primary; uint32_t node, sample; V3 side_dir, side_K;
                uint32_t spec; };      // Shade's `specount` argument (P13/main.cpp:485): lights seen on the way, for the caustic lookup

// How a sample slot of the chunk maps back to its global sample id (pixel * max_sample + j, the key of the counter RNG): the
// LDS ray records of k_wavefront do not carry the id, it is recomputed for the rays of a scene that draws random numbers
struct SlotMap { DevTiles tiles; int32_t width, height; uint32_t q0; int32_t max_sample, mode; FastDiv div_ms; };

struct ShadeCtx {
    DevScene S; DevWork W; rt_params P;
    DevRayQueue qout; uint32_t *qout_count;
    // k_wavefront only: the workgroup's ray stack in LDS, THREE 16-byte words per ray (48 B):
    //   a = (o.xyz, d.x)   b = (d.yz, thr.r, thr.g)   c = (thr.b, slot, bounce | kind << 4 | spec << 8 | material << 16, node)
    // -- the absorption the child needs on arrival travels as the index of the material that spawned it, the sample id is
    // recomputed from the slot (SlotMap).  Children go there first and to qout (the 64-byte global form) only when it is
    // full.  NULL in the per-level kernels.
    float4 *lds_a, *lds_b, *lds_c; uint32_t *lds_count; uint32_t lds_cap;
    SlotMap sm;
};

// ------------------------------------------------------------------------------------------------
// textures: TextureFile::Sample (FIN/texture.cpp:95-121: tiling + bilinear), TextureChecker::Sample
// (:125-133), TextureMap (scene.h:376-398: uvw -> itm*(uvw-pos)), TexturedColor::Sample (:422-423)
// and SampleEnvironment (:426-432).  duvw is always zero on this path, so the 32-tap filtered
// Texture::Sample (:331-349) reduces to one lookup.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 tile_clamp(V3 uvw)
{
    V3 u = mk(uvw.x - (int)uvw.x, uvw.y - (int)uvw.y, uvw.z - (int)uvw.z);
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
__device__ V3 texture_sample(const rt_texture &t, const uint8_t *texels, V3 uvw)
{
    const V3 u = tile_clamp(uvw);
    if (t.type == RT_TEX_CHECKER) {
        const float *c = (u.x <= 0.5f) ? (u.y <= 0.5f ? t.color1 : t.color2) : (u.y <= 0.5f ? t.color2 : t.color1);
        return ld3(c);
    }
    const int width = t.width, height = t.height;
    if (width + height == 0) return mk(0, 0, 0);
    const uint8_t *data = texels + t.texel_offset;
    const float x = width * u.x, y = height * u.y;
    int ix = (int)x, iy = (int)y;
    const float fx = x - ix, fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1;
    if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1;
    if (iyp >= height) iyp -= height;
    auto texel = [&](int X, int Y) { const uint8_t *q = data + 3 * ((size_t)Y * width + X); return mk(q[0] / 255.0f, q[1] / 255.0f, q[2] / 255.0f); };
    return texel(ix, iy) * ((1 - fx) * (1 - fy)) + texel(ixp, iy) * (fx * (1 - fy)) + texel(ix, iyp) * ((1 - fx) * fy) + texel(ixp, iyp) * (fx * fy);
}
__device__ V3 textured_color(const DevScene &S, V3 color, const rt_texmap &m, V3 uvw)
{
    if (m.texture == RT_MAP_NONE) return color;
    V3 t = mk(0, 0, 0);
    if (m.texture >= 0 && m.texture < S.n_textures) t = texture_sample(S.textures[m.texture], S.texels, mmul(m.itm, uvw - ld3(m.pos)));
    return color * t;
}
template <bool TEX>
__device__ V3 environment_color(const DevScene &S, V3 dir)
{
    if (!TEX || S.env_map.texture == RT_MAP_NONE) return ld3(S.env);
    const float z = asinf(-dir.z) / (float)M_PI + 0.5f;
    const float x = dir.x / (fabsf(dir.x) + fabsf(dir.y));
    const float y = dir.y / (fabsf(dir.x) + fabsf(dir.y));
    const V3 uvw = mk(0.5f, 0.5f, 0.0f) + (mk(0.5f, 0.5f, 0) * x + mk(-0.5f, 0.5f, 0) * y) * z;
    return textured_color(S, ld3(S.env), S.env_map, uvw);
}
template <bool TEX>
__device__ __forceinline__ void material_colors(const DevScene &S, const Hit &h, const rt_blinn &m, V3 &kd, V3 &ks)
{
    kd = ld3(m.diffuse); ks = ld3(m.specular);
    if (TEX && S.material_maps) {
        const int mi = S.node_material[h.node];
        kd = textured_color(S, kd, S.material_maps[2 * mi], h.uvw);       // diffuse.Sample(uvw, duvw), FIN/main.cpp:531
        ks = textured_color(S, ks, S.material_maps[2 * mi + 1], h.uvw);   // specular.Sample(uvw, duvw), :532
    }
}

// Attenuation, FIN/include/materials.h:60-66 (exp on floats resolves to the float overload)
__device__ __forceinline__ V3 attenuation(V3 a, float l) { return mk(expf(-a.x * l), expf(-a.y * l), expf(-a.z * l)); }

// Light::Direction (FIN/include/lights.h:35,51,159)
__device__ __forceinline__ V3 light_direction(const rt_light &l, V3 p)
{
    if (l.type == RT_LIGHT_POINT) return normalize(p - ld3(l.position));
    if (l.type == RT_LIGHT_DIRECT) return ld3(l.direction);
    return mk(0, 0, 0);
}

// Light::Illuminate.  Ambient: intensity (lights.h:34).  Direct: Shadow(Ray(p,-direction))*intensity
// (:50).  Point, FIN (FIN/include/lights.h:67-131): MIN_SHADOW_SAMPLES rays towards position + s*(1,1,1)
// with s = |xv|+|yv| (the reference adds the two LENGTHS to every coordinate, :103), xv,yv a random
// point of the disc of radius `size`; if their mean is neither 0 nor 1, MAX_SHADOW_SAMPLES (16) more
// replace them; result intensity*shadow/dist^2.  Point, P13 (P13/include/lights.h:65-91): 4 rays
// towards position + (dx,dy,dz), radius sqrt(Halton(i,2))*size, two random angles; P12 the same without
// the division by dist^2 (RayTracingProj12/include/lights.h:66-89).
// Each shadow ray is an any-hit query (GenLight::Shadow, FIN/main.cpp:499-513: occluded iff
// 1e-14 < z < t_max).  With size == 0 all samples coincide and ONE query decides them.
template <int MODEL>
__device__ V3 illuminate(const DevScene &S, const rt_params &P, const rt_light &l, int li, V3 p, const RngCtx &rc,
                         const BvhStack &stack, Counters &cnt)
{
    const V3 I = ld3(l.intensity);
    if (l.type == RT_LIGHT_AMBIENT) return I;
    Hit dummy;
    if (l.type == RT_LIGHT_DIRECT) {
        cnt.shadow++;
        bool occ;
        if (MODEL == RT_SHADE_P3) { Hit sh; occ = trace<false, MODEL>(S, p, -ld3(l.direction), BIGFLOAT, sh, stack, cnt) && sh.z > 1e-14f; }
        else occ = trace<true, MODEL>(S, p, -ld3(l.direction), BIGFLOAT, dummy, stack, cnt);
        return I * (occ ? 0.0f : 1.0f);
    }
    const V3 position = ld3(l.position);
    if (MODEL == RT_SHADE_P6 || MODEL == RT_SHADE_P3) {
        // PointLight::Illuminate of P6/P3 (include/lights.h:61): Shadow(Ray(p,position-p),1) * intensity.
        // P3's spheres have no bias: the CLOSEST hit decides (z in (1e-14, 1)), an any-hit would not do
        cnt.shadow++;
        bool occ;
        if (MODEL == RT_SHADE_P3) { Hit sh; occ = trace<false, MODEL>(S, p, position - p, BIGFLOAT, sh, stack, cnt) && sh.z > 1e-14f && sh.z < 1.0f; }
        else occ = trace<true, MODEL>(S, p, position - p, 1.0f, dummy, stack, cnt);
        return I * (occ ? 0.0f : 1.0f);
    }
    const int ns = P.shadow_samples > 0 ? P.shadow_samples : 4;
    if (l.size == 0) {
        cnt.shadow++;
        const bool occ = trace<true, MODEL>(S, p, position - p, 1.0f, dummy, stack, cnt);
        float coefsum = 0.0f;
        for (int i = 0; i < ns; i++) coefsum += occ ? 0.0f : 1.0f;     // the ns identical samples
        if (MODEL == RT_SHADE_P12) return (I * coefsum) / (float)ns;    // no fall-off yet in RayTracingProj12 (include/lights.h:86-88)
        if (MODEL != RT_SHADE_FIN) return ((I * coefsum) / (float)ns) / len2(p - position);
        const float shadow = coefsum / (float)ns;
        return (I * shadow) / len2(p - position);                       // lights.h:130
    }
    if (MODEL != RT_SHADE_FIN) {
        float coef = 0.0f;
        for (int i = 0; i < ns; i++) {
            float u0, u1;
            rng2(rc, RNG_SHADOW, (uint32_t)(li * 64 + i), u0, u1);
            float r = halton(i, 2);
            r = sqrtf(r) * l.size;
            const float theta = (float)(M_PI * 2.0 * (double)u0);
            const float gam = (float)(M_PI * (double)u1);
            const float dx = r * sinf(gam) * cosf(theta), dy = r * sinf(gam) * sinf(theta), dz = r * cosf(gam);
            const V3 lp = mk(dx, dy, dz) + position;
            cnt.shadow++;
            coef += trace<true, MODEL>(S, p, lp - p, 1.0f, dummy, stack, cnt) ? 0.0f : 1.0f;
        }
        if (MODEL == RT_SHADE_P12) return (I * coef) / (float)ns;       // RayTracingProj12/include/lights.h:86-88: avg_shadow, undivided
        return ((I * coef) / (float)ns) / len2(p - position);          // P13/include/lights.h:86-90: inverse square fall-off
    }
    const V3 dir = position - p;
    V3 v1 = dot(dir, mk(1, 0, 0)) > 0.8f ? cross(mk(0, 1, 0), dir) : cross(mk(1, 0, 0), dir);
    V3 v2 = cross(v1, dir);
    v2 = normalize(v2);
    v1 = normalize(v1);
    float shadow = 0.0f;
    for (int i = 0; i < ns; i++) {
        float u0, u1;
        rng2(rc, RNG_SHADOW, (uint32_t)(li * 64 + i), u0, u1);
        const float rRadius = sqrtf(u0) * l.size;
        const float rAngle = (float)((double)u1 * (2.0 * M_PI));
        const float xv = rRadius * cosf(rAngle), yv = rRadius * sinf(rAngle);
        const float lx = sqrtf(len2(v1 * xv)), ly = sqrtf(len2(v2 * yv));
        const V3 tgt = mk(position.x + lx + ly, position.y + lx + ly, position.z + lx + ly);
        cnt.shadow++;
        shadow += trace<true, MODEL>(S, p, tgt - p, 1.0f, dummy, stack, cnt) ? 0.0f : 1.0f;
    }
    shadow /= (float)ns;
    if (shadow != 0.0f && shadow != 1.0f) {
        shadow = 0.0f;
        for (int i = 0; i < 16; i++) {                                  // MAX_SHADOW_SAMPLES
            float u0, u1;
            rng2(rc, RNG_SHADOW, (uint32_t)(li * 64 + 32 + i), u0, u1);
            const float rRadius = sqrtf(u0) * l.size;
            const float rAngle = (float)((double)u1 * (2.0 * M_PI));
            const float xv = rRadius * cosf(rAngle), yv = rRadius * sinf(rAngle);
            const float lx = sqrtf(len2(v1 * (-xv))), ly = sqrtf(len2(v2 * (-yv)));
            const V3 tgt = mk(position.x + lx + ly, position.y + lx + ly, position.z + lx + ly);
            cnt.shadow++;
            shadow += trace<true, MODEL>(S, p, tgt - p, 1.0f, dummy, stack, cnt) ? 0.0f : 1.0f;
        }
        shadow /= 16.0f;
    }
    return (I * shadow) / len2(p - position);
}

__device__ __forceinline__ void add_sample(const ShadeCtx &C, uint32_t slot, V3 c, bool primary)
{
    float *dst = C.W.sample_rgb + 3 * (size_t)slot;
    if (primary) { dst[0] = c.x; dst[1] = c.y; dst[2] = c.z; }
    else { atomicAdd(dst, c.x); atomicAdd(dst + 1, c.y); atomicAdd(dst + 2, c.z); }
}

template <bool SIDE = false, bool TWO_ENDED = false>
__device__ __forceinline__ void push_ray(const ShadeCtx &C, bool pred, V3 o, V3 d, V3 thr, V3 absorb,
                                         uint32_t slot, int bounce, uint32_t kind, uint32_t node, uint32_t sample,
                                         V3 side_dir = V3{0, 0, 0}, V3 side_K = V3{0, 0, 0}, uint32_t spec = 0, uint32_t mat = 0)
{
    if (!SIDE && C.lds_count) {
        if constexpr (TWO_ENDED) {
            // two stacks in one array, keyed by the kind of ray (side 0: reflection rays grow up from 0, side 1: refraction and hemisphere
            // rays grow down from the top); ONE packed counter (low half: side 0, high half: side 1) so that an add sees both fills
            const uint32_t side = kind == KIND_REFLECT ? 0u : 1u;
            bool stored = false;
            for (uint32_t sd = 0; sd < 2; sd++) {
                const bool mine = pred && side == sd;
                const unsigned long long mask = ballot64(mine);
                if (!mask) continue;
                const int lane = __lane_id();
                const int leader = __ffsll((long long)mask) - 1;
                const uint32_t n = (uint32_t)__popcll(mask), sh = 16u * sd;
                uint32_t base = 0, room = 0;
                if (lane == leader) {
                    const uint32_t old = atomicAdd(C.lds_count, n << sh);
                    base = (old >> sh) & 0xFFFFu;
                    const uint32_t fill = (old & 0xFFFFu) + (old >> 16);
                    room = fill >= C.lds_cap ? 0u : C.lds_cap - fill;
                    if (room < n) atomicSub(C.lds_count, (n - room) << sh);      // only what was stored counts
                }
                base = __shfl(base, leader); room = __shfl(room, leader);
                const uint32_t off = lanes_below(mask);
                if (mine && off < room) {
                    const uint32_t at = sd ? C.lds_cap - 1u - (base + off) : base + off;
                    C.lds_a[at] = make_float4(o.x, o.y, o.z, d.x);
                    C.lds_b[at] = make_float4(d.y, d.z, thr.x, thr.y);
                    C.lds_c[at] = make_float4(thr.z, __uint_as_float(slot), __uint_as_float((uint32_t)bounce | (kind << 4) | (spec << 8) | (mat << 16)), __uint_as_float(node));
                    stored = true;
                }
            }
            pred = pred && !stored;
            if (!__any(pred)) return;
        } else {
            // the wave's own stack first (LDS atomic, one per wave); what does not fit goes to the global queue
            const uint32_t at = wave_push(pred, C.lds_count);
            if (pred && at < C.lds_cap) {
                C.lds_a[at] = make_float4(o.x, o.y, o.z, d.x);
                C.lds_b[at] = make_float4(d.y, d.z, thr.x, thr.y);
                C.lds_c[at] = make_float4(thr.z, __uint_as_float(slot), __uint_as_float((uint32_t)bounce | (kind << 4) | (spec << 8) | (mat << 16)), __uint_as_float(node));
            }
            pred = pred && at >= C.lds_cap;
            if (!__any(pred)) return;
        }
    }
    const uint32_t idx = wave_push(pred, C.qout_count);
    if (pred) {
        if (idx < C.qout.cap) {
            C.qout.a[idx] = make_float4(o.x, o.y, o.z, d.x);
            C.qout.b[idx] = make_float4(d.y, d.z, thr.x, thr.y);
            if (SIDE) {
                C.qout.c[idx] = make_float4(thr.z, absorb.x, side_K.y, side_K.z);
                C.qout.e[idx] = make_float4(side_dir.x, side_dir.y, side_dir.z, side_K.x);
            } else C.qout.c[idx] = make_float4(thr.z, absorb.x, absorb.y, absorb.z);
            C.qout.d[idx] = make_uint4(slot, (uint32_t)bounce | (kind << 8) | (spec << 16), node, sample);
        } else atomicAdd(&C.W.stats[ST_AT(ST_QUEUE_OVERFLOW)], 1ull);
    }
}

__device__ __forceinline__ void push_photon_query(const ShadeCtx &C, bool pred, V3 p, V3 N, V3 w, uint32_t slot, bool caustic = false)
{
    const DevPhotonQueue &Q = caustic ? C.W.cq : C.W.pq;
    const uint32_t idx = wave_push(pred, C.W.counts + (caustic ? CNT_CAUSTICQ : CNT_PHOTONQ));
    if (pred) {
        if (idx < Q.cap) {
            Q.qa[idx] = make_float4(p.x, p.y, p.z, N.x);
            Q.qb[idx] = make_float4(N.y, N.z, w.x, w.y);
            Q.qc[idx] = make_float4(w.z, __uint_as_float(slot), 0.f, 0.f);
        } else atomicAdd(&C.W.stats[ST_AT(ST_QUEUE_OVERFLOW)], 1ull);
    }
}

// What one Shade() call produces besides its local colour: up to two child rays with their weights
// and (FIN only) a photon-map query.
struct ShadeOut {
    V3 color;                        // local colour (emission + direct light), before the ray weight
    bool want_refl, want_refr, want_photon;
    V3 rdir, tdir;                   // child directions (reflection is normalised by the caller side)
    V3 rK, tK;                       // child weights relative to this ray
    V3 child_absorb;                 // what the children need to finish their own weight on arrival
    V3 kd, N;                        // photon query: diffuse colour and shading normal
    bool want_side;                  // P6: a reflection ray that exists only if the refraction ray hits
    V3 side_dir, side_K;
    int n_caustic;                   // P13: caustic-map lookups this hit makes (one per light past specount > 2, all identical)
    uint32_t spec_out;               // P13: specount handed to the children
    int n_gi;                        // P12: hemisphere rays to spawn (weights/dirs are drawn at push time)
    V3 gi_x, gi_y, gi_z;             // P12: frame of the hemisphere
    uint32_t mat;                    // index of the hit's material (what child_absorb was read from)
};

// MtlBlinn::Shade, FIN/main.cpp:516-708
template <bool TEX>
__device__ void shade_fin(const DevScene &S, const rt_params &P, const Hit &h, V3 ray_d, int bounce, const RngCtx &rc,
                          ShadeOut &o, const BvhStack &stack, Counters &cnt)
{
    const rt_blinn &m = S.materials[S.node_material[h.node]];
    V3 color = ld3(m.emission);                                         // :517
    const V3 p = h.p;
    const V3 N = normalize(h.N);                                        // :521-522
    const V3 direction = normalize(-ray_d);                             // :523-524
    V3 kd, ks;
    material_colors<TEX>(S, h, m, kd, ks);
    const float gloss = m.glossiness;
    const V3 reflection = ld3(m.reflection), refraction = ld3(m.refraction);
    const float ior = m.ior;
    const float coef = S.n_lights == 0 ? 1.0f : 1.0f / S.n_lights;      // :545
    for (int li = 0; li < S.n_lights; li++) {
        const rt_light light = cld_light(S.lights + li);
        // the reference still calls Illuminate (its shadow rays) for a back-face hit but uses the
        // result only for front hits (:551-553): nothing to add, nothing traced
        if (!h.front) continue;
        const V3 Il = illuminate<RT_SHADE_FIN>(S, P, light, li, p, rc, stack, cnt);
        if (light.type != RT_LIGHT_AMBIENT) {
            const V3 intensity = Il * coef;                             // :551
            V3 L = light_direction(light, p) * (float)(-1);             // :556
            L = normalize(L);
            const V3 H = normalize(L + direction);
            const float cosNL = RMAX(0.f, dot(N, L));
            const float cosNH = RMAX(0.f, dot(N, H));
            const V3 diffuse = (kd * intensity) * cosNL;                // :563
            const V3 specular = ((ks * intensity) * powf(cosNH, gloss)) * cosNL;   // :564 (std::pow(float,float))
            color = color + (diffuse + specular);                       // :566
        } else {
            color = color + kd * Il;                                    // :568-569
        }
    }
    // reflection / refraction set-up, :577-610
    float ein = 1, eout = ior;
    if (!h.front) { ein = ior; eout = 1; }
    const float eta = ein / eout;
    const float cosI = dot(N, direction);
    const V3 Y = cosI > 0.f ? N : -N;
    const V3 Z = cross(direction, Y);
    const V3 X = normalize(cross(Y, Z));
    // sqrtf(1 - cosI*cosI) is NaN in the reference when |cosI| exceeds 1 by rounding (:592);
    // clamped at 0 here (documented deviation)
    const float sinI = sqrtf(fmaxf(0.0f, 1 - cosI * cosI));
    const float sinO = RMAX(0.f, RMIN(1.f, sinI * eta));
    const float cosO = sqrtf(1.f - sinO * sinO);
    o.tdir = normalize((-X) * sinO - Y * cosO);                         // :596, tRay.Normalize() :627
    o.rdir = normalize((N * 2.f) * cosI - direction);                   // :597, r.Normalize() :615
    const float C0 = (eta - 1.f) * (eta - 1.f) / ((eta + 1.f) * (eta + 1.f));
    const float rC = C0 + (1.f - C0) * powf(1.f - fabsf(cosI), 5.f);    // :601
    const float tC = 1.f - rC;
    const bool totReflection = (eta * sinI) > 1.001f;                   // materials.h:20
    o.tK = totReflection ? mk(0.f, 0.f, 0.f) : refraction * tC;
    o.rK = totReflection ? (reflection + refraction) : (reflection + refraction * rC);
    const float th = 0.001f;                                            // materials.h:21-22
    o.want_refl = bounce > 0 && (o.rK.x > th || o.rK.y > th || o.rK.z > th);   // :613
    o.want_refr = bounce > 0 && (o.tK.x > th || o.tK.y > th || o.tK.z > th);   // :625
    // :642-693: at bounceCount == BOUNCE the hemisphere loop's result is assigned to a shadowing
    // variable and contributes exactly 0 -- not traced.  :695-705: every other hit adds
    // kd * irradiance * max(0, N.(-dir)); queued for k_gather.
    o.want_photon = (bounce != P.bounce) && S.pm.n_leaves != 0;
    o.color = color; o.kd = kd; o.N = N;
    o.child_absorb = ld3(m.absorption);      // children: K *= Attenuation(absorption, z) on a back-face hit (:620,:632)
    o.mat = (uint32_t)S.node_material[h.node];
}

// MtlBlinn::Shade, P13/main.cpp:485-756 (reflectionGlossiness == refractionGlossiness == 0).
// all = ambient + direct; all += re_color*reflection; all += refraction*(ra_ratio*absorb*ra_color +
// re_ratio*re_color): the reflection child weighs reflection + refraction*re_ratio, the refraction
// child refraction*ra_ratio*exp(-absorption.r * z_child) (z_child = BIGFLOAT on a miss).
template <int MODEL, bool TEX>
__device__ void shade_p13(const DevScene &S, const rt_params &P, const Hit &h, V3 ray_d, int bounce, const RngCtx &rc,
                          ShadeOut &o, const BvhStack &stack, Counters &cnt, uint32_t spec_in)
{
    constexpr bool p12 = MODEL == RT_SHADE_P12;
    const rt_blinn &m = S.materials[S.node_material[h.node]];
    V3 N = h.N;
    const V3 Pp = h.p;
    V3 Kd, Ks;
    material_colors<TEX>(S, h, m, Kd, Ks);
    const float alpha = m.glossiness;
    V3 ambient = mk(0, 0, 0), diffuse = mk(0, 0, 0);
    uint32_t spec = spec_in;
    int n_caustic = 0;
    for (int i = 0; i < S.n_lights; i++) {
        const rt_light l = cld_light(S.lights + i);
        const V3 Il = illuminate<MODEL>(S, P, l, i, Pp, rc, stack, cnt);
        if (l.type == RT_LIGHT_AMBIENT) ambient = ambient + Il * Kd;                     // :510
        else {
            // the caustic lookup the reference sketches in a comment (:518-533): per non-ambient light, on a photon
            // surface (diffuse.Gray() > 0), once specount has passed 2; specount++ after every such light
            if (gray(ld3(m.diffuse)) > 0 && spec > 2u) n_caustic++;
            spec++;
            V3 L = light_direction(l, Pp) * (float)-1;
            if (!p12) L = normalize(L);                                                   // P13 adds L.Normalize() (:540)
            const V3 V = normalize(-ray_d);
            const V3 H = normalize(L + V);
            const V3 kse = Ks * powf(dot(N, H), alpha) + Kd;                              // :547
            const float theta = dot(N, L);
            diffuse = diffuse + (Il * (theta > 0.0f ? theta : 0.0f)) * kse;               // :551
        }
    }
    o.color = ambient + diffuse;                                                          // :622
    o.spec_out = spec > 255u ? 255u : spec;
    o.n_caustic = (P.caustic_k > 0 && S.cm.n_leaves != 0) ? n_caustic : 0;
    o.n_gi = 0;
    if (p12) {
        // RayTracingProj12 main.cpp:393-448: all = ambient + ((diffuse/pi) + idr)*Kd, idr = mean over the
        // hemisphere rays of childColour * (dir.N): local part here, rays spawned by the caller
        o.color = ambient + (diffuse / (float)M_PI) * Kd;
        if (bounce > 0) {
            o.n_gi = (bounce == P.bounce) ? P.hemisphere_sample : 1;
            o.gi_z = h.N;
            V3 nx = dot(o.gi_z, mk(1, 0, 0)) < 0.4f ? cross(o.gi_z, mk(1, 0, 0)) : cross(o.gi_z, mk(0, 0, 1));
            o.gi_x = normalize(nx);
            o.gi_y = cross(o.gi_z, o.gi_x);
        }
    }
    V3 V = -normalize(ray_d);                                                             // :632
    // glossy reflection: the normal is jittered inside a disc of radius reflectionGlossiness (:635-647)
    V3 Nr = N;
    if (m.reflection_glossiness != 0) {
        const V3 newx = cross(Nr, mk(1, 0, 0));
        const V3 newy = cross(Nr, newx);
        float u0, u1;
        rng2(rc, RNG_GLOSSR, 0u, u0, u1);
        const float r = sqrtf(u0) * m.reflection_glossiness;
        const float theta = (float)(M_PI * 2.0 * (double)u1);
        Nr = normalize(Nr + (newx * (r * cosf(theta)) + newy * (r * sinf(theta))));
    }
    const float costheta = fminf(fmaxf(dot(Nr, V), -1.0f), 1.0f);                         // clamp :648
    o.rdir = normalize(Nr * (2 * costheta) - V);                                          // :649, :652
    // refraction block :671-751 (starts again from the unjittered normal, :672)
    N = h.N;
    if (m.refraction_glossiness != 0) {                                                   // :673-686
        const V3 newx = cross(N, mk(1, 0, 0));
        const V3 newy = cross(N, newx);
        float u0, u1;
        rng2(rc, RNG_GLOSST, 0u, u0, u1);
        const float r = sqrtf(u0) * m.refraction_glossiness;
        const float theta = (float)(M_PI * 2.0 * (double)u1);
        N = normalize(N + (newx * (r * cosf(theta)) + newy * (r * sinf(theta))));
    }
    V = normalize(V);
    const float costheta1 = fabsf(dot(V, N));
    const float sintheta1 = sqrtf(RMAX(0.0f, 1 - (costheta1 * costheta1)));
    float n1 = 1.0f, n2 = 1.0f;
    if (h.front) n2 = m.ior; else { n1 = m.ior; N = -N; }
    const float ratio_n = n1 / n2;
    const float sintheta2 = ratio_n * sintheta1;
    float re_ratio = 0.0f, ra_ratio = 0.0f;
    o.tdir = mk(0, 0, 1);
    bool refr_ok = false;
    if (sintheta2 <= 1.0f) {
        const float costheta2 = sqrtf(RMAX(0.0f, 1 - (sintheta2 * sintheta2)));
        V3 Sv = cross(N, cross(N, V));
        N = normalize(N);
        Sv = normalize(Sv);
        o.tdir = (-N) * costheta2 + Sv * sintheta2;                                       // not normalised (:718-720)
        float R0 = (n1 - n2) / (n1 + n2);
        R0 = R0 * R0;
        const double tmp = 1.0 - costheta1;
        re_ratio = (float)(R0 + (1.0 - R0) * pow(tmp, 5.0));                              // :733 (double)
        ra_ratio = (float)(1.0 - re_ratio);
        refr_ok = true;
    } else re_ratio = 1.0f;
    const V3 reflection = ld3(m.reflection), refraction = ld3(m.refraction);
    o.rK = reflection + refraction * re_ratio;
    o.tK = refraction * ra_ratio;
    o.want_refl = bounce > 0;
    o.want_refr = bounce > 0 && refr_ok;
    o.want_photon = false;
    o.kd = Kd; o.N = h.N;
    o.child_absorb = mk(m.absorption[0], 0, 0);          // refraction child: *= exp(-absorption.r * z) (:728)
    o.mat = (uint32_t)S.node_material[h.node];
}

// MtlBlinn::Shade of RayTracingProj3, main.cpp:152-190: ambient + Blinn with V = camera.pos - p (all P3
// rays start at the camera, so camera.pos is the ray origin); no children
template <bool TEX>
__device__ void shade_p3(const DevScene &S, const rt_params &P, const Hit &h, V3 ray_o, const RngCtx &rc, ShadeOut &o,
                         const BvhStack &stack, Counters &cnt)
{
    const rt_blinn &m = S.materials[S.node_material[h.node]];
    const V3 N = h.N, Pp = h.p;
    V3 Kd, Ks;
    material_colors<TEX>(S, h, m, Kd, Ks);
    V3 ambient = mk(0, 0, 0), diffuse = mk(0, 0, 0);
    for (int i = 0; i < S.n_lights; i++) {
        const rt_light l = cld_light(S.lights + i);
        const V3 Il = illuminate<RT_SHADE_P3>(S, P, l, i, Pp, rc, stack, cnt);
        if (l.type == RT_LIGHT_AMBIENT) ambient = ambient + Il * Kd;
        else {
            const V3 L = light_direction(l, Pp) * (float)-1;
            const V3 V = normalize(ray_o - Pp);
            const V3 LpV = L + V;
            const V3 H = normalize(LpV / sqrtf(len2(LpV)));
            const V3 kse = Ks * powf(dot(N, H), m.glossiness) + Kd;
            const float theta = dot(N, L);
            diffuse = diffuse + (Il * (theta > 0 ? theta : 0.0f)) * kse;
        }
    }
    o.color = ambient + diffuse;
    o.kd = Kd; o.N = N;
}

// MtlBlinn::Shade of RayTracingProj6, main.cpp:175-340.  Children: a reflection ray only when
// reflection.Gray() > 0 (weight reflection); a refraction ray only when refraction.Gray() > 0 (weight
// refraction*ra_ratio*exp(-absorption.r*z)); and refraction*re_ratio times a reflection colour that the
// reference adds ONLY IF THE REFRACTION RAY HIT something (re_ratio is set inside that branch, :296-303) --
// or under total internal reflection.  That conditional ray travels with the refraction ray as a
// "side" ray and is spawned when the refraction ray reports a hit.  Misses add nothing (no environment).
template <bool TEX>
__device__ void shade_p6(const DevScene &S, const rt_params &P, const Hit &h, V3 ray_d, int bounce, const RngCtx &rc,
                         ShadeOut &o, const BvhStack &stack, Counters &cnt)
{
    const rt_blinn &m = S.materials[S.node_material[h.node]];
    V3 N = h.N;
    const V3 Pp = h.p;
    V3 Kd, Ks;
    material_colors<TEX>(S, h, m, Kd, Ks);
    const V3 reflection = ld3(m.reflection), refraction = ld3(m.refraction);
    V3 ambient = mk(0, 0, 0), diffuse = mk(0, 0, 0);
    for (int i = 0; i < S.n_lights; i++) {
        const rt_light l = cld_light(S.lights + i);
        const V3 Il = illuminate<RT_SHADE_P6>(S, P, l, i, Pp, rc, stack, cnt);
        if (l.type == RT_LIGHT_AMBIENT) ambient = ambient + Il * Kd;                          // :199
        else {
            const V3 L = light_direction(l, Pp) * (float)-1;
            const V3 V = normalize(-ray_d);
            const V3 H = normalize(L + V);
            const V3 kse = Ks * powf(dot(N, H), m.glossiness) + Kd;                           // :210
            const float theta = dot(N, L);
            diffuse = diffuse + (Il * (theta > 0.0f ? theta : 0.0f)) * kse;                   // :214
        }
    }
    o.color = ambient + diffuse;
    o.kd = Kd; o.N = h.N;
    V3 V = -normalize(ray_d);                                                                 // :223
    const bool has_re = gray(reflection) > 0;
    const float ct0 = fminf(fmaxf(dot(N, V), -1.0f), 1.0f);
    const V3 R1 = normalize(N * (2 * ct0) - V);                                               // :227-229
    o.rdir = R1;
    o.rK = reflection;
    o.want_refl = has_re && bounce > 0;
    o.want_refr = false; o.want_side = false;
    o.tK = mk(0, 0, 0); o.tdir = mk(0, 0, 1); o.side_dir = mk(0, 0, 1); o.side_K = mk(0, 0, 0);
    o.child_absorb = mk(m.absorption[0], 0, 0);
    if (gray(refraction) > 0 && bounce > 0) {                                                 // :246
        V = normalize(V);
        const float costheta1 = fabsf(dot(V, N));
        const float sintheta1 = sqrtf(RMAX(0.0f, 1 - (costheta1 * costheta1)));
        float n1 = 1.0f, n2 = 1.0f;
        if (h.front) n2 = m.ior; else { n1 = m.ior; N = -h.N; }
        const float ratio_n = n1 / n2;
        const float sintheta2 = ratio_n * sintheta1;
        float re_ratio, ra_ratio = 0.0f;
        const bool transmit = sintheta2 <= 1.0f;
        if (transmit) {
            const float costheta2 = sqrtf(RMAX(0.0f, 1 - (sintheta2 * sintheta2)));
            V3 Sv = cross(N, cross(N, V));
            N = normalize(N);
            Sv = normalize(Sv);
            o.tdir = (-N) * costheta2 + Sv * sintheta2;
            float R0 = (n1 - n2) / (n1 + n2);
            R0 = R0 * R0;
            const double tmp = 1.0 - costheta1;
            re_ratio = (float)(R0 + (1.0 - R0) * pow(tmp, 5.0));                              // if the refraction ray hits
            ra_ratio = (float)(1.0 - re_ratio);
            o.tK = refraction * ra_ratio;
            o.want_refr = true;
        } else re_ratio = 1.0f;                                                               // total internal reflection
        // the reflection colour that refraction*re_ratio multiplies: the first reflection ray again when the
        // material is reflective (:310), else a second ray about the (flipped, normalised) normal, direction
        // NOT normalised (:314-319)
        V3 sdir = R1;
        if (!has_re) { const float ct = fminf(fmaxf(dot(N, V), -1.0f), 1.0f); sdir = N * (2 * ct) - V; }
        const V3 sK = refraction * re_ratio;
        if (re_ratio > 0.0f || has_re) {
            if (transmit) { o.want_side = true; o.side_dir = sdir; o.side_K = sK; }          // only if the refraction ray hits
            else if (has_re) o.rK = reflection + sK;                                          // same ray as the first reflection
            else { o.want_refl = true; o.rdir = sdir; o.rK = sK; }
        }
    }
}

// One node of the ray tree: Trace + MtlBlinn::Shade with the recursion unrolled into queue pushes.
// Shade is linear in its children (color += K*child), so a ray carries the product `thr` of the K
// factors above it and adds thr*local colour to its sample; the part of a child's K that depends on
// the child's own hit (FIN: Attenuation(parent absorption, z) on a back-face hit, FIN/main.cpp:620,
// 632; P13: exp(-absorption.r*z) on the refraction child, P13/main.cpp:728) is applied on arrival.
template <int MODEL, bool TEX>
__device__ void shade_path(const ShadeCtx &C, const PathIn &in, bool active, const BvhStack &stack, Counters &cnt)
{
    const DevScene &S = C.S;
    const rt_params &P = C.P;
    constexpr bool p13 = MODEL == RT_SHADE_P13 || MODEL == RT_SHADE_P12;     // they share lights and the ray tree
    Hit h;
    bool hit = false;
    if (active) hit = trace<false, MODEL, TEX>(S, in.o, in.d, BIGFLOAT, h, stack, cnt);
    V3 thr = in.thr;
    ShadeOut o;
    o.want_refl = o.want_refr = o.want_photon = false;
    o.rdir = o.tdir = mk(0, 0, 1); o.rK = o.tK = o.child_absorb = o.kd = o.N = mk(0, 0, 0);
    o.n_gi = 0; o.gi_x = o.gi_y = o.gi_z = mk(0, 0, 1);
    o.want_side = false; o.side_dir = mk(0, 0, 1); o.side_K = mk(0, 0, 0);
    o.n_caustic = 0; o.spec_out = in.spec; o.mat = 0;
    constexpr bool p6 = MODEL == RT_SHADE_P6, p3 = MODEL == RT_SHADE_P3;
    // the workgroup-shared ray stack of k_wavefront is two-ended (push_ray); a wave's own stack in the P12 rounds is plain
    constexpr bool TWO = !(RT_WF_PERWAVE && MODEL == RT_SHADE_P12);
    bool spawn_side = false;
    if (active && !in.primary) {
        if (p6) {
            // refraction ray of P6: weight *= exp(-absorption.r*z) when it hits (main.cpp:296); its side ray
            // (the reflection that refraction*re_ratio multiplies) exists only in that case
            if (in.kind == KIND_REFRACT && hit) { thr = thr * expf(-in.absorb.x * h.z); spawn_side = (in.side_K.x != 0.f || in.side_K.y != 0.f || in.side_K.z != 0.f); }
        } else if (p13) { if (in.kind == KIND_REFRACT) thr = thr * expf(-in.absorb.x * (hit ? h.z : BIGFLOAT)); }   // not for GI rays
        else if (hit && !h.front) thr = thr * attenuation(in.absorb, h.z);
    }
    if (active && !hit) {
        if (in.primary) C.W.sample_hit[in.slot] = 0;
        // a refraction ray that leaves the scene sees the environment (FIN/main.cpp:635); in P13 so
        // does a reflection ray (P13/main.cpp:660-662)
        else if (!p6 && !p3 && (in.kind != KIND_REFLECT || p13)) add_sample(C, in.slot, thr * environment_color<TEX>(S, in.d), false);
    }
    if (active && hit) {
        if (in.primary) { C.W.sample_hit[in.slot] = 1; C.W.sample_z[in.slot] = h.z; }
        RngCtx rc; rc.seed = P.seed; rc.sample = in.sample; rc.node = in.node;
        if (p3) shade_p3<TEX>(S, P, h, in.o, rc, o, stack, cnt);
        else if (p6) shade_p6<TEX>(S, P, h, in.d, in.bounce, rc, o, stack, cnt);
        else if (p13) shade_p13<MODEL, TEX>(S, P, h, in.d, in.bounce, rc, o, stack, cnt, in.spec);
        else shade_fin<TEX>(S, P, h, in.d, in.bounce, rc, o, stack, cnt);
        add_sample(C, in.slot, thr * o.color, in.primary);
        // a child (or query) whose accumulated weight is exactly zero cannot change the pixel
        const V3 wr = thr * o.rK, wt = thr * o.tK, wp = thr * o.kd;
        o.want_refl = o.want_refl && (wr.x != 0.f || wr.y != 0.f || wr.z != 0.f);
        o.want_refr = o.want_refr && (wt.x != 0.f || wt.y != 0.f || wt.z != 0.f);
        o.want_photon = o.want_photon && (wp.x != 0.f || wp.y != 0.f || wp.z != 0.f);
    }
    // pushes are wave-collective: every lane of the wave reaches them
    push_ray<false, TWO>(C, o.want_refl, h.p, o.rdir, thr * o.rK, o.child_absorb, in.slot, in.bounce - 1, KIND_REFLECT,
             child_node(in.node, 1u), in.sample, mk(0, 0, 0), mk(0, 0, 0), o.spec_out, o.mat);
    if (p6) {
        const V3 sK = thr * o.side_K;
        push_ray<true>(C, o.want_refr, h.p, o.tdir, thr * o.tK, o.child_absorb, in.slot, in.bounce - 1, KIND_REFRACT,
                       child_node(in.node, 2u), in.sample, o.side_dir, o.want_side ? sK : mk(0, 0, 0));
        // the side ray of an arriving refraction ray: same origin and level as that ray
        push_ray<false, TWO>(C, spawn_side, in.o, in.side_dir, in.side_K, mk(0, 0, 0), in.slot, in.bounce, KIND_REFLECT, child_node(in.node, 4u), in.sample);
    } else
    push_ray<false, TWO>(C, o.want_refr, h.p, o.tdir, thr * o.tK, o.child_absorb, in.slot, in.bounce - 1, KIND_REFRACT,
             child_node(in.node, 2u), in.sample, mk(0, 0, 0), mk(0, 0, 0), o.spec_out, o.mat);
    push_photon_query(C, o.want_photon, h.p, o.N, thr * o.kd, in.slot);
    if (p13) {
        // cau_Color += Kd * causticrad * theta, once per counted light (P13/main.cpp:518-531): one query, weight x count
        const V3 wc = (thr * o.kd) * (float)o.n_caustic;
        push_photon_query(C, active && hit && o.n_caustic > 0 && (wc.x != 0.f || wc.y != 0.f || wc.z != 0.f), h.p, o.N, wc, in.slot, true);
    }
    if (MODEL == RT_SHADE_P12) {
        // hemisphere rays: idr += childColour * (dir.N) / Nofsample, times Kd (main.cpp:420-446)
        const int n_max = __reduce_max_sync_compat(o.n_gi);
        for (int i = 0; i < n_max; i++) {
            bool want = i < o.n_gi;
            V3 hd = mk(0, 0, 1), w = mk(0, 0, 0);
            if (want) {
                RngCtx rc; rc.seed = P.seed; rc.sample = in.sample; rc.node = in.node;
                float u0, u1;
                rng2(rc, RNG_GI, (uint32_t)i, u0, u1);
                const float phi = (float)(2 * M_PI * (double)u0);
                const float cosphi = cosf(phi);
                const float sintheta = sqrtf(u1), costheta = sqrtf(1 - u1);
                hd = normalize(o.gi_x * (sintheta * cosphi) + o.gi_y * (sintheta * sinf(phi)) + o.gi_z * costheta);
                const float dotN_wi = dot(hd, o.gi_z);
                w = thr * (o.kd * (dotN_wi / (float)o.n_gi));
                hd = normalize(hd);                        // Ray_idr.dir.Normalize() (:433)
                want = (w.x != 0.f || w.y != 0.f || w.z != 0.f);
            }
            push_ray<false, TWO>(C, want, h.p, hd, w, mk(0, 0, 0), in.slot, in.bounce - 1, KIND_GI, child_node(in.node, 3u + (uint32_t)i), in.sample, mk(0, 0, 0), mk(0, 0, 0), o.spec_out);
        }
    }
}

// The whole workgroup calls this once, at the end of its kernel.  The waves' counts meet in LDS and ONE thread per counter adds them
// to the statistics block: a wave-level flush was 7 device-scope atomics per wave on one 128-byte line, all of them when the launch
// drains -- 21 000 per k_wavefront launch queueing at the memory side while the kernel waits to retire (r4: 0.4 ms of the Cornell
// frame's tracer, 0.3 of its gather).  The block's counters sit RT_STAT_STRIDE apart for the same reason (rt_dev.h).
__device__ __forceinline__ void flush_counters(unsigned long long *stats, const Counters &c, uint32_t nprim, uint32_t nrefl, uint32_t nrefr)
{
    __shared__ unsigned long long s_acc[7];
    if (threadIdx.x < 7) s_acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t v[7] = {c.inst, c.nodes, c.tris, c.shadow, nprim, nrefl, nrefr};
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint32_t x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        v[i] = x;
    }
    if (__lane_id() == 0) {
#pragma unroll
        for (int i = 0; i < 7; i++) if (v[i]) atomicAdd(&s_acc[i], (unsigned long long)v[i]);
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int slot = threadIdx.x == 0 ? ST_INSTANCE_VISITS : threadIdx.x == 1 ? ST_BVH_NODES : threadIdx.x == 2 ? ST_TRIS : threadIdx.x == 3 ? ST_RAYS_SHADOW
                       : threadIdx.x == 4 ? ST_RAYS_PRIMARY : threadIdx.x == 5 ? ST_RAYS_REFLECT : ST_RAYS_REFRACT;
        const unsigned long long x = s_acc[threadIdx.x];
        if (x) atomicAdd(&stats[ST_AT(slot)], x);
    }
}

// chunk-local pixel q -> image pixel, walking this call's tiles (tile-major, row-major inside)
__device__ __forceinline__ bool pixel_of(const DevTiles &T, const DevCamera &cam, uint32_t q, int &x, int &y)
{
    const uint32_t per = (uint32_t)(T.tile_w * T.tile_h);
    const uint32_t k = fastdiv(q, T.div_per), w = q - k * per;
    if (k >= (uint32_t)T.n_tiles) return false;
    const uint32_t t = (uint32_t)T.first + k * (uint32_t)T.stride;
    const uint32_t ty = fastdiv(t, T.div_tiles_x), tx = t - ty * (uint32_t)T.tiles_x;
    const uint32_t wy = fastdiv(w, T.div_tile_w), wx = w - wy * (uint32_t)T.tile_w;
    x = (int)(tx * (uint32_t)T.tile_w + wx);
    y = (int)(ty * (uint32_t)T.tile_h + wy);
    return x < cam.width && y < cam.height;
}

// ------------------------------------------------------------------------------------------------
// K1: one thread per (pixel, sample).  generateSample + ray build, FIN/main.cpp:147-162, 281-292
// (dof == 0).  Lanes of a wave hold consecutive samples of the same pixel(s): coherent traversal.
//   mode 0: pixels q0..q0+npix of the chunk, samples j0..j0+ns
//   mode 1: pixels from W.pixel_list[0..counts[CNT_PIXLIST]), samples j0..j0+ns   (second batch)
//   mode 2: rays[] supplied by the caller, one sample each (rt_shade_rays)
// ------------------------------------------------------------------------------------------------
struct PrimaryArgs {
    DevCamera cam; DevTiles tiles;
    uint32_t q0, npix;           // chunk range (mode 0)
    int j0, ns, max_sample, mode;
    const float *rays;           // mode 2
    FastDiv div_ns;
    uint32_t lds_rays;           // k_wavefront: entries of the LDS ray stack to use (0: all of them; the test hook RT_WF_LDS_RAYS makes the stack overflow on small frames)
    DevRayQueue qsrc;            // mode 3 (k_wavefront only): the work is a ray queue (count in *qsrc_count), not primary samples
    const uint32_t *qsrc_count;
};

// Register budget: the texture-free instantiations are held to 168 VGPRs (3 waves/SIMD; a few values
// spill to scratch) -- measured 4 % faster on MI355X than letting them grow to 192 VGPRs at 2 waves.
// sample gid of this launch -> its primary ray (generateSample + ray build, FIN/main.cpp:147-162, 281-292); false = no such sample
__device__ __forceinline__ bool primary_setup(const ShadeCtx &C, const PrimaryArgs &A, unsigned long long gid, unsigned long long total,
                                              bool h_table, const float *s_h2, const float *s_h3, PathIn &in)
{
    in.thr = mk(1.f, 1.f, 1.f); in.absorb = mk(0, 0, 0); in.bounce = C.P.bounce; in.kind = KIND_REFLECT; in.primary = true;
    in.o = mk(0, 0, 0); in.d = mk(0, 0, 1); in.slot = 0; in.node = 1; in.sample = 0;
    in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0); in.spec = 0;
    if (gid >= total) return false;
    uint32_t pi, jj;
    if (total <= 0xFFFFFFFFull) { pi = fastdiv((uint32_t)gid, A.div_ns); jj = (uint32_t)gid - pi * (uint32_t)A.ns; }
    else { pi = (uint32_t)(gid / (unsigned long long)A.ns); jj = (uint32_t)(gid % (unsigned long long)A.ns); }
    const int j = A.j0 + (int)jj;
    const uint32_t ql = (A.mode == 1) ? C.W.pixel_list[pi] : pi;      // chunk-local pixel
    in.slot = ql * (uint32_t)A.max_sample + (uint32_t)j;
    if (A.mode == 2) {
        const float *r = A.rays + 6 * (size_t)ql;
        in.o = ld3(r); in.d = ld3(r + 3);
        in.sample = A.q0 + ql;                     // caller's ray index
        return true;
    }
    int x, y;
    if (!pixel_of(A.tiles, A.cam, A.q0 + ql, x, y)) return false;
    const V3 tmp = mk(x * A.cam.u, y * A.cam.v, 0) + ld3(A.cam.b);     // :235-236
    float sx = (h_table ? s_h2[jj] : halton(j, 2)) * A.cam.u;          // :153
    float sy = A.cam.v * (h_table ? s_h3[jj] : halton(j, 3));          // :154
    sx += tmp.x; sy += tmp.y;
    const V3 sample = mk(sx, sy, tmp.z);
    const uint32_t pixel_id = (uint32_t)y * (uint32_t)A.cam.width + (uint32_t)x;
    in.sample = pixel_id * (uint32_t)A.max_sample + (uint32_t)j;
    V3 d_campos = mk(0, 0, 0);
    if (A.cam.dof != 0) {
        // :246-262: a table of CAM_SAMPLE lens points per pixel (radius sqrt(Halton(i,2))*dof,
        // random angle), of which every sample picks one at random (:284)
        RngCtx pc; pc.seed = C.P.seed; pc.sample = in.sample; pc.node = 0;
        float u0, u1;
        rng2(pc, RNG_PICK, 0u, u0, u1);
        int pick = (int)(u0 * 64.0f);
        pick = pick > 63 ? 63 : pick;
        RngCtx lc; lc.seed = C.P.seed; lc.sample = pixel_id; lc.node = 0;
        rng2(lc, RNG_LENS, (uint32_t)(pick + 1), u0, u1);
        float r = halton(pick + 1, 2);
        r = sqrtf(r) * A.cam.dof;
        const float theta = (float)(M_PI * 2.0 * (double)u0);
        d_campos = mmul(A.cam.m, mk(r * cosf(theta), r * sinf(theta), 0));
    }
    in.o = ld3(A.cam.pos) + d_campos;                                  // :288
    in.d = normalize(mmul(A.cam.m, sample) - d_campos);                // :289-292
    return true;
}

#define RT_TRACE_OCC __attribute__((amdgpu_waves_per_eu(TEX ? 2 : 3)))
template <int MODEL, bool TEX>
RT_TRACE_OCC __global__ __launch_bounds__(RT_BLOCK) void k_primary(ShadeCtx C, PrimaryArgs A)
{
    __shared__ uint32_t s_stack[RT_BVH_STACK * RT_BLOCK];
    const BvhStack stack = bvh_stack(s_stack, RT_BVH_STACK, C.S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    uint32_t nprim = 0;
    const uint32_t npix = (A.mode == 1) ? C.W.counts[CNT_PIXLIST] : A.npix;
    const unsigned long long total = (unsigned long long)npix * (unsigned long long)A.ns;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    // Halton(j, 2) and Halton(j, 3) of this launch's sample indices, once per workgroup (the loop in
    // halton() divides; the values are the same floats either way)
    __shared__ float s_h2[RT_BLOCK], s_h3[RT_BLOCK];
    const bool h_table = A.mode != 2 && A.ns <= RT_BLOCK;
    if (h_table && (int)threadIdx.x < A.ns) { s_h2[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 2); s_h3[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 3); }
    __syncthreads();
    // every lane iterates the same number of times (wave-collective pushes inside shade_path)
    for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < total; base += stride) {
        PathIn in;
        const bool active = primary_setup(C, A, base + threadIdx.x, total, h_table, s_h2, s_h3, in);
        if (active) nprim++;
        shade_path<MODEL, TEX>(C, in, active, stack, cnt);
    }
    flush_counters(C.W.stats, cnt, nprim, 0, 0);
}

// K2-K4 for one level of the ray tree: reads queue `qin` (count in counts[level]).
template <int MODEL, bool TEX>
RT_TRACE_OCC __global__ __launch_bounds__(RT_BLOCK) void k_bounce(ShadeCtx C, DevRayQueue qin, int level)
{
    __shared__ uint32_t s_stack[RT_BVH_STACK * RT_BLOCK];
    const BvhStack stack = bvh_stack(s_stack, RT_BVH_STACK, C.S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    uint32_t nrefl = 0, nrefr = 0;
    uint32_t total = C.W.counts[level];
    if (total > qin.cap) total = qin.cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += stride) {
        const uint32_t gid = base + threadIdx.x;
        const bool active = gid < total;
        PathIn in;
        in.primary = false; in.thr = mk(0, 0, 0); in.absorb = mk(0, 0, 0); in.o = mk(0, 0, 0); in.d = mk(0, 0, 1);
        in.slot = 0; in.bounce = 0; in.kind = 0; in.node = 1; in.sample = 0; in.spec = 0;
        in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0);
        if (active) {
            const uint32_t src = gid;
            const float4 a = qin.a[src], b = qin.b[src], c = qin.c[src];
            const uint4 dd = qin.d[src];
            in.o = mk(a.x, a.y, a.z); in.d = mk(a.w, b.x, b.y);
            in.thr = mk(b.z, b.w, c.x); in.absorb = mk(c.y, c.z, c.w);
            in.slot = dd.x; in.bounce = (int)(dd.y & 0xFFu); in.kind = (dd.y >> 8) & 0xFFu; in.spec = (dd.y >> 16) & 0xFFu; in.node = dd.z; in.sample = dd.w;
            if (MODEL == RT_SHADE_P6 && in.kind == KIND_REFRACT) {
                const float4 e = qin.e[src];
                in.side_dir = mk(e.x, e.y, e.z); in.side_K = mk(e.w, c.z, c.w);
                in.absorb = mk(c.y, 0, 0);
            }
            if (in.kind == KIND_REFLECT) nrefl++; else nrefr++;
        }
        shade_path<MODEL, TEX>(C, in, active, stack, cnt);
    }
    flush_counters(C.W.stats, cnt, 0, nrefl, nrefr);
}

// ------------------------------------------------------------------------------------------------
// The whole ray tree of a chunk in ONE launch: a persistent grid of workgroups, each with its own ray stack in
// LDS (BASELINE.json north_star: "persistent-threads wavefront tracer with ray queues in LDS").  A workgroup
// takes 256 primary samples at a time from a device-side counter; the reflection/refraction children its rays
// spawn are pushed onto ITS stack (wave-aggregated LDS atomic) and popped 256 at a time -- whenever 256 are
// waiting, or when the primary samples have run out -- so the deep, sparse levels of the tree are traced by
// full wavefronts next to other workgroups' primary rays instead of in launches of their own that cannot fill
// 256 CUs.  Nothing is handed between workgroups (no cross-CU visibility protocol needed); a ray that finds
// the stack full goes to the global queue and is picked up by the per-level k_bounce launches that follow
// (normally empty).  FIN and P13 models (at most two children per hit); the others keep the per-level kernels.
// ------------------------------------------------------------------------------------------------
// Configuration: the texture-free instantiations run THREE workgroups per CU (3 waves/SIMD: the traversal waits on dependent
// loads for 0.44 of its wave cycles at two) -- 168 VGPRs, and per workgroup at most 53 760 B of LDS (160 KB / 3 in 1280-byte
// granules): RT_BVH_LDS = 8 BVH stack entries per lane (8 KB; a deeper traversal continues in the HBM spill part, rt_dev.h) + a
// 928-ray stack of 48-byte records (44.5 KB) + two 64-entry Halton tables.  (Round 3 and the start of round 4: 24 entries + 592
// rays.  The rays that do not fit the stack take the global queue, a second pass and the per-level launches: 7.8 M of the Cornell
// frame's 42 M secondary rays and 1.8 ms at 592 rays, 0.6 M and 0.36 ms at 928; C3: 27 -> 13 ms.)  The textured
// instantiations get the same configuration (the compiler would take 255 VGPRs for them; held to 168 the texture lookups'
// temporaries go to scratch, which costs less than the third wave buys).  RT_WF_WAVES=2 / RT_WF_TEX_WAVES=2: two workgroups
// per CU, 32-entry BVH stacks, 1008 rays (A/B builds).
#ifndef RT_WF_WAVES
#define RT_WF_WAVES 3          // waves per SIMD = workgroups per CU of the texture-free instantiations
#endif
#ifndef RT_WF_TEX_WAVES
#define RT_WF_TEX_WAVES 3      // ... of the textured ones (measured on the 102 k-triangle frame under a PNG sky: 40.5 ms at two, 31.2 ms at three)
#endif
#define RT_WF_HALTON 64
template <bool TEX> struct WfCfg {
    static constexpr int WAVES = TEX ? RT_WF_TEX_WAVES : RT_WF_WAVES;
#ifndef RT_WF_STACK
#define RT_WF_STACK 928
#endif
    static constexpr int BVH = WAVES >= 3 ? RT_BVH_LDS : RT_BVH_STACK;
    static constexpr int STACK = WAVES >= 3 ? RT_WF_STACK : 1008;
    // pop a round as soon as a round of primary rays (two children each) could no longer be sure to fit
    static constexpr int POP = STACK - 2 * RT_BLOCK + 1 < RT_BLOCK ? STACK - 2 * RT_BLOCK + 1 : RT_BLOCK;
    static_assert(BVH * RT_BLOCK * 4 + STACK * 48 + 2 * RT_WF_HALTON * 4 + 16 <= (WAVES >= 3 ? 53760 : 81920), "LDS budget of the occupancy the kernel is built for");
};
// global sample id of a chunk slot (see SlotMap)
__device__ __forceinline__ uint32_t sample_of_slot(const SlotMap &M, uint32_t slot)
{
    if (M.mode == 2) return M.q0 + slot;
    const uint32_t ql = fastdiv(slot, M.div_ms), j = slot - ql * (uint32_t)M.max_sample;
    DevCamera cam; cam.width = M.width; cam.height = M.height;
    int x = 0, y = 0;
    pixel_of(M.tiles, cam, M.q0 + ql, x, y);
    return ((uint32_t)y * (uint32_t)M.width + (uint32_t)x) * (uint32_t)M.max_sample + j;
}
template <int MODEL, bool TEX>
__attribute__((amdgpu_waves_per_eu(TEX ? RT_WF_TEX_WAVES : RT_WF_WAVES, TEX ? RT_WF_TEX_WAVES : RT_WF_WAVES))) __global__ __launch_bounds__(RT_BLOCK) void k_wavefront(ShadeCtx C, PrimaryArgs A)
{
    using Cfg = WfCfg<TEX>;
    __shared__ uint32_t s_stack[Cfg::BVH * RT_BLOCK];
    __shared__ float4 s_qa[Cfg::STACK], s_qb[Cfg::STACK], s_qc[Cfg::STACK];
    __shared__ uint32_t s_count, s_batch;
    __shared__ float s_h2[RT_WF_HALTON], s_h3[RT_WF_HALTON];
  if constexpr (RT_WF_PERWAVE && MODEL == RT_SHADE_P12) {
    // RayTracingProj12's paths (one hemisphere ray per hit, eight levels deep: every round is a pop round): every WAVE runs its
    // own rounds on its own quarter of the ray stack, no workgroup barrier anywhere in the loop.  A round's length varies by
    // an order of magnitude with what its 64 rays meet, and with a shared stack the four waves of a workgroup meet at three
    // barriers per round (C3: 307 -> 272 ms).  Not for the FIN / P13 trees: their glass doubles the rays level by level, a
    // quarter stack overflows into the global queue and its per-level launches (Cornell tracer 16.8 -> 32.2 ms, measured).
    constexpr int NW = RT_BLOCK / 64;
    constexpr uint32_t STACK_W = (uint32_t)Cfg::STACK / NW;
    constexpr uint32_t POP_W = STACK_W > 128u ? (STACK_W - 127u < 64u ? STACK_W - 127u : 64u) : 1u;   // pop before a round of primaries (two children each) could overflow
    __shared__ uint32_t s_countw[NW];
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    float4 *qa = s_qa + wv * STACK_W, *qb = s_qb + wv * STACK_W, *qc = s_qc + wv * STACK_W;
    const BvhStack stack = bvh_stack(s_stack, Cfg::BVH, C.S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    uint32_t nprim = 0, nrefl = 0, nrefr = 0;
    const uint32_t npix = (A.mode == 1) ? C.W.counts[CNT_PIXLIST] : A.npix;
    const unsigned long long total = A.mode == 3 ? (unsigned long long)min(*A.qsrc_count, A.qsrc.cap)
                                                 : (unsigned long long)npix * (unsigned long long)A.ns;
    const unsigned long long n_batches = (total + 63ull) / 64ull;
    const bool h_table = A.mode < 2 && A.ns <= RT_WF_HALTON;
    uint32_t *next_batch = C.W.counts + (A.mode == 3 ? CNT_WF2_NEXT : CNT_PRIMARY_NEXT);
    if (h_table && (int)threadIdx.x < A.ns) { s_h2[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 2); s_h3[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 3); }
    if (lane == 0) s_countw[wv] = 0;
    C.lds_a = qa; C.lds_b = qb; C.lds_c = qc; C.lds_count = &s_countw[wv]; C.lds_cap = STACK_W;
    __syncthreads();                                      // the Halton table (the only thing the waves share)
    (void)s_count; (void)s_batch;
    auto wsync = []() {                                   // LDS operations of one wave complete in issue order: only the compiler has to be held
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    bool more_primaries = true;                           // wave-uniform
    for (;;) {
        wsync();                                          // last round's pushes are complete
        uint32_t waiting = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_countw[wv]);
        if (waiting > STACK_W) waiting = STACK_W;         // the excess went to the global queue
        const bool pop = waiting >= POP_W || (!more_primaries && waiting > 0);
        if (!pop && !more_primaries) break;
        PathIn in;
        bool active;
        if (pop) {
            const uint32_t n = min(waiting, 64u);
            active = (uint32_t)lane < n;
            in.primary = false; in.thr = mk(0, 0, 0); in.absorb = mk(0, 0, 0); in.o = mk(0, 0, 0); in.d = mk(0, 0, 1);
            in.slot = 0; in.bounce = 0; in.kind = 0; in.node = 1; in.sample = 0; in.spec = 0;
            in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0);
            if (active) {
                const uint32_t src = waiting - 1u - (uint32_t)lane;       // newest (deepest) first: the stack stays shallow
                const float4 a = qa[src], b = qb[src], c = qc[src];
                in.o = mk(a.x, a.y, a.z); in.d = mk(a.w, b.x, b.y);
                in.thr = mk(b.z, b.w, c.x);
                in.slot = __float_as_uint(c.y);
                const uint32_t pk = __float_as_uint(c.z);
                in.bounce = (int)(pk & 0xFu); in.kind = (pk >> 4) & 0xFu; in.spec = (pk >> 8) & 0xFFu; in.node = __float_as_uint(c.w);
                const float *ab = C.S.materials[pk >> 16].absorption;
                in.absorb = (MODEL == RT_SHADE_FIN) ? ld3(ab) : mk(ab[0], 0, 0);
                if (C.S.stochastic || MODEL == RT_SHADE_P12) in.sample = sample_of_slot(C.sm, in.slot);     // P12 draws its hemisphere rays
                if (in.kind == KIND_REFLECT) nrefl++; else nrefr++;
            }
            wsync();                                      // all pops read before anything is pushed over them
            if (lane == 0) s_countw[wv] = waiting - n;
        } else {
            uint32_t bt = 0;
            if (lane == 0) { bt = atomicAdd(next_batch, 1u); if (s_countw[wv] > STACK_W) s_countw[wv] = STACK_W; }
            const unsigned long long batch = (uint32_t)__builtin_amdgcn_readfirstlane((int)bt);
            if (batch >= n_batches) { more_primaries = false; continue; }
            if (A.mode == 3) {
                const unsigned long long src = batch * 64ull + (unsigned)lane;
                active = src < total;
                in.primary = false; in.thr = mk(0, 0, 0); in.absorb = mk(0, 0, 0); in.o = mk(0, 0, 0); in.d = mk(0, 0, 1);
                in.slot = 0; in.bounce = 0; in.kind = 0; in.node = 1; in.sample = 0; in.spec = 0;
                in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0);
                if (active) {
                    const float4 a = A.qsrc.a[src], b = A.qsrc.b[src], c = A.qsrc.c[src];
                    const uint4 dd = A.qsrc.d[src];
                    in.o = mk(a.x, a.y, a.z); in.d = mk(a.w, b.x, b.y);
                    in.thr = mk(b.z, b.w, c.x); in.absorb = mk(c.y, c.z, c.w);
                    in.slot = dd.x; in.bounce = (int)(dd.y & 0xFFu); in.kind = (dd.y >> 8) & 0xFFu; in.spec = (dd.y >> 16) & 0xFFu; in.node = dd.z; in.sample = dd.w;
                    if (in.kind == KIND_REFLECT) nrefl++; else nrefr++;
                }
            } else {
                active = primary_setup(C, A, batch * 64ull + (unsigned)lane, total, h_table, s_h2, s_h3, in);
                if (active) nprim++;
            }
        }
        wsync();                                          // the count is settled before this round's pushes
        shade_path<MODEL, TEX>(C, in, active, stack, cnt);
    }
    flush_counters(C.W.stats, cnt, nprim, nrefl, nrefr);
  } else {
    const BvhStack stack = bvh_stack(s_stack, Cfg::BVH, C.S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    uint32_t nprim = 0, nrefl = 0, nrefr = 0;
    const uint32_t npix = (A.mode == 1) ? C.W.counts[CNT_PIXLIST] : A.npix;
    // mode 3: the rays that did not fit the LDS stacks of the pass before (they sit in a global queue) are the work items;
    // their descendants live on this pass's LDS stacks like everybody else's
    const unsigned long long total = A.mode == 3 ? (unsigned long long)min(*A.qsrc_count, A.qsrc.cap)
                                                 : (unsigned long long)npix * (unsigned long long)A.ns;
    const unsigned long long n_batches = (total + RT_BLOCK - 1) / RT_BLOCK;
    const bool h_table = A.mode < 2 && A.ns <= RT_WF_HALTON;
    uint32_t *next_batch = C.W.counts + (A.mode == 3 ? CNT_WF2_NEXT : CNT_PRIMARY_NEXT);
    if (h_table && (int)threadIdx.x < A.ns) { s_h2[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 2); s_h3[threadIdx.x] = halton(A.j0 + (int)threadIdx.x, 3); }
    if (threadIdx.x == 0) s_count = 0;
    C.lds_a = s_qa; C.lds_b = s_qb; C.lds_c = s_qc; C.lds_count = &s_count; C.lds_cap = (A.lds_rays && A.lds_rays < (uint32_t)Cfg::STACK) ? A.lds_rays : (uint32_t)Cfg::STACK;
    bool more_primaries = true;                           // workgroup-uniform
    // A workgroup takes RT_WF_GRAB rounds of 256 per atomic on the work counter when the launch has plenty: the returning atomic is a
    // round trip to the memory side with the whole workgroup at the barrier behind it (r4, pairs: -2.7 % on the 102 k-triangle frame,
    // Cornell tracer 13.67 -> 13.35 ms with its gather 0.13 ms slower for the changed order of its queue; fours: the gather loses more).
    // The per-wave rounds of P12 keep one round per atomic: pairs cost C3 0.9 %, fours 3 %.
    const uint32_t grab = n_batches >= 128ull * gridDim.x ? (uint32_t)RT_WF_GRAB : 1u;
    unsigned long long grabbed = 0; uint32_t in_hand = 0; // workgroup-uniform: the rounds this workgroup holds
    for (;;) {
        __syncthreads();                                  // last round's pushes are complete
        const uint32_t packed = s_count;
        const uint32_t w0 = packed & 0xFFFFu, w1 = packed >> 16;
        const uint32_t pside = w1 > w0 ? 1u : 0u;         // the fuller side is popped
        const uint32_t waiting = w0 + w1;                 // never beyond the stack: a push that does not fit takes its count back (push_ray)
        // pop a full workgroup's worth when there is one -- and already earlier when a round of primary rays (up to two
        // children each) could no longer be sure to fit: rays that do not fit go through the global queue and the
        // per-level launches, the slow path (measured: 5 ms per Cornell frame before this rule)
        const bool pop = waiting >= (uint32_t)Cfg::POP || (!more_primaries && waiting > 0);
        if (!pop && !more_primaries) break;
        __syncthreads();                                  // everyone has read s_count
        PathIn in;
        bool active;
        if (pop) {
            // always a whole workgroup's worth when there is one: shrinking the round so that a nearly full stack could not
            // overflow was measured slower (rounds of n >= 64 / 128 / 192: Cornell trace 28.1 / 26.8 / 25.7 ms vs 24.9; with no
            // floor, rounds of a handful of rays whose children refill the stack at once: 66.5 ms) -- the few rays that do not
            // fit take the global queue
            const uint32_t wside = pside ? w1 : w0;
            const uint32_t n = min(wside, (uint32_t)RT_BLOCK);
            active = threadIdx.x < n;
            in.primary = false; in.thr = mk(0, 0, 0); in.absorb = mk(0, 0, 0); in.o = mk(0, 0, 0); in.d = mk(0, 0, 1);
            in.slot = 0; in.bounce = 0; in.kind = 0; in.node = 1; in.sample = 0; in.spec = 0;
            in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0);
            if (active) {
                const uint32_t src = pside ? C.lds_cap - wside + threadIdx.x : wside - 1u - threadIdx.x;
                const float4 a = s_qa[src], b = s_qb[src], c = s_qc[src];
                in.o = mk(a.x, a.y, a.z); in.d = mk(a.w, b.x, b.y);
                in.thr = mk(b.z, b.w, c.x);
                in.slot = __float_as_uint(c.y);
                const uint32_t pk = __float_as_uint(c.z);
                in.bounce = (int)(pk & 0xFu); in.kind = (pk >> 4) & 0xFu; in.spec = (pk >> 8) & 0xFFu; in.node = __float_as_uint(c.w);
                // what the child needs of its parent's material on arrival: Attenuation(absorption, z) (FIN/main.cpp:620,632),
                // exp(-absorption.r * z) (P13/main.cpp:728)
                const float *ab = C.S.materials[pk >> 16].absorption;
                in.absorb = (MODEL == RT_SHADE_FIN) ? ld3(ab) : mk(ab[0], 0, 0);
                if (C.S.stochastic || MODEL == RT_SHADE_P12) in.sample = sample_of_slot(C.sm, in.slot);     // P12 draws its hemisphere rays
                if (in.kind == KIND_REFLECT) nrefl++; else nrefr++;
            }
            __syncthreads();                              // all pops read before anything is pushed over them
            if (threadIdx.x == 0) s_count = packed - (n << (16u * pside));
        } else {
            if (in_hand == 0) {                           // `grab` rounds per visit of the work counter
                if (threadIdx.x == 0) s_batch = atomicAdd(next_batch, 1u);
                __syncthreads();
                grabbed = (unsigned long long)s_batch * grab;
                in_hand = grab;
            }
            const unsigned long long batch = grabbed + (grab - in_hand);
            in_hand--;
            if (batch >= n_batches) { more_primaries = false; continue; }
            if (A.mode == 3) {
                const unsigned long long src = batch * RT_BLOCK + threadIdx.x;
                active = src < total;
                in.primary = false; in.thr = mk(0, 0, 0); in.absorb = mk(0, 0, 0); in.o = mk(0, 0, 0); in.d = mk(0, 0, 1);
                in.slot = 0; in.bounce = 0; in.kind = 0; in.node = 1; in.sample = 0; in.spec = 0;
                in.side_dir = mk(0, 0, 1); in.side_K = mk(0, 0, 0);
                if (active) {
                    const float4 a = A.qsrc.a[src], b = A.qsrc.b[src], c = A.qsrc.c[src];
                    const uint4 dd = A.qsrc.d[src];
                    in.o = mk(a.x, a.y, a.z); in.d = mk(a.w, b.x, b.y);
                    in.thr = mk(b.z, b.w, c.x); in.absorb = mk(c.y, c.z, c.w);
                    in.slot = dd.x; in.bounce = (int)(dd.y & 0xFFu); in.kind = (dd.y >> 8) & 0xFFu; in.spec = (dd.y >> 16) & 0xFFu; in.node = dd.z; in.sample = dd.w;
                    if (in.kind == KIND_REFLECT) nrefl++; else nrefr++;
                }
            } else {
                active = primary_setup(C, A, batch * RT_BLOCK + threadIdx.x, total, h_table, s_h2, s_h3, in);
                if (active) nprim++;
            }
        }
        __syncthreads();                                  // s_count settled before this round's pushes
        shade_path<MODEL, TEX>(C, in, active, stack, cnt);
    }
    flush_counters(C.W.stats, cnt, nprim, nrefl, nrefr);
  }
}

// K2 alone: n closest-hit queries (rt_trace_rays)
template <int MODEL>
__global__ __launch_bounds__(RT_BLOCK) void k_trace(DevScene S, const float *rays, long long n,
                                                    uint8_t *hit, float *z, float *p, float *N, int32_t *node, uint8_t *front)
{
    __shared__ uint32_t s_stack[RT_BVH_STACK * RT_BLOCK];
    const BvhStack stack = bvh_stack(s_stack, RT_BVH_STACK, S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        Hit h;
        h.z = BIGFLOAT; h.p = mk(0, 0, 0); h.N = mk(0, 0, 0); h.node = -1; h.front = 1;
        const bool ok = trace<false, MODEL>(S, ld3(rays + 6 * i), ld3(rays + 6 * i + 3), BIGFLOAT, h, stack, cnt);
        hit[i] = ok ? 1 : 0;
        z[i] = ok ? h.z : BIGFLOAT;
        p[3 * i] = h.p.x; p[3 * i + 1] = h.p.y; p[3 * i + 2] = h.p.z;
        N[3 * i] = h.N.x; N[3 * i + 1] = h.N.y; N[3 * i + 2] = h.N.z;
        node[i] = ok ? h.node : -1;
        front[i] = (uint8_t)h.front;
    }
}

// ------------------------------------------------------------------------------------------------
// Photon pass: generatePhotonMap / PhotonTracing (FIN/main.cpp:350-459),
// PointLight::RandomPhoton (:489-497) and MtlBlinn::RandomPhotonBounce (FIN/include/materials.h:
// 99-256), one thread per emission attempt.  The reference draws from libc rand(); here every draw
// is Philox-4x32-10 keyed by the seed and counted by (attempt index, draw index), so the photon set
// is a pure function of (scene, seed) whatever the scheduling -- and the CPU oracle, using the same
// generator, reproduces it photon for photon (up to libm rounding).
// ------------------------------------------------------------------------------------------------
struct Philox {
    uint32_t key0, key1, c0, c1, blk; uint32_t o0, o1, o2, o3; int used;
    __host__ __device__ void refill()
    {
        uint32_t x0 = c0, x1 = c1, x2 = blk, x3 = 0, k0 = key0, k1 = key1;
        for (int r = 0; r < 10; r++) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        o0 = x0; o1 = x1; o2 = x2; o3 = x3;
        blk++; used = 0;
    }
    // uniform in [0,1): stands in for rand() / (float) RAND_MAX
    __host__ __device__ float next()
    {
        if (used >= 4) refill();
        const uint32_t v = used == 0 ? o0 : (used == 1 ? o1 : (used == 2 ? o2 : o3));
        used++;
        return (float)(v >> 8) * (1.0f / 16777216.0f);
    }
};

struct PhotonArgs {
    unsigned long long first_attempt; uint32_t n_attempts;
    uint32_t seed; int max_bounce;
    float *out;            // [n_attempts][RT_PHOTON_SLOTS][9]: pos, dir, power
    uint32_t *count;       // [n_attempts]: photons stored | diffuse hits counted << 16
    int mode;              // 0: photon map (PhotonTracing), 1: caustic map (CausticTracing)
};


// MtlBlinn::RandomPhotonBounce, FIN/include/materials.h:99-256 (all branches, incl. the glossy ones)
__device__ bool random_photon_bounce(const rt_blinn &m, const Hit &h, V3 &rp, V3 &rd, V3 &c, Philox &rng)
{
    const V3 V = -rd;
    const V3 N = h.N;
    const float NV = dot(N, V);
    const V3 Y = NV > 0.f ? N : -N;
    float ein = 1, eout = m.ior;
    if (!h.front) { ein = m.ior; eout = 1; }
    const float eta = ein / eout;
    const V3 Z = cross(V, Y);
    const V3 X = normalize(cross(Y, Z));
    const float cosI = NV;
    const float sinI = sqrtf(fmaxf(0.0f, 1 - cosI * cosI));
    const float sinO = RMAX(0.f, RMIN(1.f, sinI * eta));
    const float cosO = sqrtf(1.f - sinO * sinO);
    const V3 tDir = (-X) * sinO - Y * cosO;
    const V3 rDir = (N * 2.f) * NV - V;
    const float C0 = (eta - 1.f) * (eta - 1.f) / ((eta + 1.f) * (eta + 1.f));
    const float rC = C0 + (1.f - C0) * powf(1.f - fabsf(cosI), 5.f);
    const float tC = 1.f - rC;
    const bool tot = (eta * sinI) > 1.001f;
    const V3 tK = ld3(m.refraction), rK = ld3(m.reflection);
    const V3 sRefr = tot ? mk(0, 0, 0) : tK * tC;
    const V3 sRefl = tot ? (rK + tK) : (rK + tK * rC);
    const V3 sDiff = ld3(m.diffuse), sSpec = ld3(m.specular);
    const float random = rng.next();                                    // :150
    float diffuseProb = gray(sDiff), refractionProb = gray(sRefr), reflectionProb = gray(sRefl), absorptionProb = gray(ld3(m.absorption));
    const float total = diffuseProb + reflectionProb + refractionProb + absorptionProb;
    diffuseProb /= total; refractionProb /= total; reflectionProb /= total;
    const float rcp = 1.f / total;
    const float select = random * total;                                // :163 (compared with the NORMALISED probabilities)
    const float luma = 0.00001f;
    int selected; float scale = 1.f;
    if (select <= refractionProb && refractionProb > luma) { selected = 0; scale = refractionProb * rcp; }
    else if (select > refractionProb && select <= refractionProb + reflectionProb && reflectionProb > luma) { selected = 1; scale = reflectionProb * rcp; }
    else if (select > refractionProb + reflectionProb && select < refractionProb + reflectionProb + diffuseProb && diffuseProb > luma) { selected = 2; scale = diffuseProb * rcp; }
    else selected = 3;
    V3 dir, BxDF;
    if (selected == 0) {
        if (m.refraction_glossiness > 0.f) {               // :183-190: SampleHemisphere (:40-48), in ITS frame, used as is
            const float u1 = rng.next(), u2 = rng.next();
            const float r = sqrtf(1.0f - u1 * u1);
            const float phi = (float)(2 * M_PI * (double)u2);
            dir = mk(cosf(phi) * r, sinf(phi) * r, u1);
            const V3 L = normalize(dir);
            const V3 H = normalize(V + L);
            const float cosVH = RMAX(0.f, dot(V, H));
            BxDF = sRefr * powf(cosVH, m.refraction_glossiness);
        } else { dir = tDir; BxDF = sRefr; }
    } else if (selected == 1) {
        if (m.reflection_glossiness > 0.f) {               // :200-207: CosineSampleHemisphere (:27-38)
            const float u1 = rng.next(), u2 = rng.next();
            const float r = sqrtf(u1);
            const float theta = (float)(2 * M_PI * (double)u2);
            dir = mk(r * cosf(theta), r * sinf(theta), sqrtf(fmaxf(0.0f, 1 - u1)));
            const V3 L = normalize(dir);
            const V3 H = normalize(V + L);
            const float cosNH = RMAX(0.f, dot(N, H));
            BxDF = sRefl * powf(cosNH, m.reflection_glossiness);
        } else { dir = rDir; BxDF = sRefl; }
    } else if (selected == 2) {
        if (!h.front) return false;
        // createCoordinateSystem, materials.h:50-59
        V3 Nt = dot(N, mk(1, 0, 0)) < 0.4f ? cross(N, mk(1, 0, 0)) : cross(N, mk(0, 0, 1));
        Nt = normalize(Nt);
        const V3 Nb = cross(N, Nt);
        const float theta = (float)((double)rng.next() * M_PI_2);       // :227
        const float phi = (float)((double)rng.next() * (2.0 * M_PI));   // :228
        dir = (Nt * cosf(phi)) * sinf(theta) + (Nb * sinf(phi)) * sinf(theta) + N * cosf(theta);
        const V3 L = normalize(dir);
        const V3 H = normalize(V + L);
        const float cosNH = RMAX(0.f, dot(N, H));
        BxDF = sDiff + sSpec * powf(cosNH, m.glossiness);
    } else return false;
    rp = h.p; rd = normalize(dir);
    c = (c * BxDF) / (1.f * scale);                                     // c * BxDF / (PDF * scale)
    if (!h.front) c = c * attenuation(ld3(m.absorption), h.z);
    return true;
}

#define RT_PHOTON_SLOTS 8
__global__ __launch_bounds__(RT_BLOCK) void k_photon_trace(DevScene S, PhotonArgs A)
{
    __shared__ uint32_t s_stack[RT_BVH_STACK * RT_BLOCK];
    const BvhStack stack = bvh_stack(s_stack, RT_BVH_STACK, S.bvh_spill);
    Counters cnt = {0, 0, 0, 0, 0};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_attempts) return;
    const unsigned long long attempt = A.first_attempt + i;
    Philox rng;
    rng.key0 = A.seed; rng.key1 = 0x52544D49u; rng.c0 = (uint32_t)attempt; rng.c1 = (uint32_t)(attempt >> 32); rng.blk = 0; rng.used = 4;
    // choose one photon source uniformly (the reference hard-codes two lights, :367-370)
    int npl = 0;
    for (int l = 0; l < S.n_lights; l++) if (S.lights[l].type == RT_LIGHT_POINT) npl++;
    uint32_t stored = 0, hits = 0;
    float *out = A.out + (size_t)i * RT_PHOTON_SLOTS * 9;
    if (npl > 0) {
        int pick = (int)(rng.next() * (float)npl);
        if (pick >= npl) pick = npl - 1;
        int li = 0;
        for (int l = 0; l < S.n_lights; l++) if (S.lights[l].type == RT_LIGHT_POINT) { if (pick == 0) { li = l; break; } pick--; }
        const rt_light &L = S.lights[li];
        V3 c = ld3(L.intensity);
        const V3 position = ld3(L.position);
        // PointLight::RandomPhoton, :489-497
        const float x = 2 * rng.next() - 1, y = 2 * rng.next() - 1, z = 2 * rng.next() - 1;
        V3 rp = position;
        V3 rd = normalize((mk(x, y, z) + position) - position);
        Hit h;
        if (A.mode == 1) {
            // caustic pass (P13/main.cpp:383-398 + CausticTracing :431-457): the first hit sets hitspec (0 on a photon
            // surface, else 1); then every diffuse hit is COUNTED and, behind more than one specular hit, STORED
            if (trace<false, RT_SHADE_FIN>(S, rp, rd, BIGFLOAT, h, stack, cnt)) {
                const rt_blinn *m = &S.materials[S.node_material[h.node]];
                int hitspec = gray(ld3(m->diffuse)) > 0 ? 0 : 1;
                int bounce = A.max_bounce;
                while (bounce > 0 && random_photon_bounce(*m, h, rp, rd, c, rng)) {
                    Hit nh;
                    if (!trace<false, RT_SHADE_FIN>(S, rp, rd, BIGFLOAT, nh, stack, cnt)) break;
                    m = &S.materials[S.node_material[nh.node]];
                    if (gray(ld3(m->diffuse)) > 0) {
                        if (hitspec > 1 && stored < RT_PHOTON_SLOTS) {
                            float *o = out + 9 * stored;
                            o[0] = nh.p.x; o[1] = nh.p.y; o[2] = nh.p.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = c.x; o[7] = c.y; o[8] = c.z;
                            stored++;
                        }
                        hits++;
                    } else hitspec++;
                    bounce--;
                    h = nh;
                }
            }
        } else
        if (trace<false, RT_SHADE_FIN>(S, rp, rd, BIGFLOAT, h, stack, cnt)) {
            const rt_blinn *m = &S.materials[S.node_material[h.node]];
            if (gray(ld3(m->diffuse)) > 0) {                             // IsPhotonSurface, materials.h:97
                int bounce = A.max_bounce;
                while (bounce > 0 && random_photon_bounce(*m, h, rp, rd, c, rng)) {      // PhotonTracing :439-459
                    Hit nh;
                    if (!trace<false, RT_SHADE_FIN>(S, rp, rd, BIGFLOAT, nh, stack, cnt)) break;
                    m = &S.materials[S.node_material[nh.node]];
                    if (gray(ld3(m->diffuse)) > 0 && stored < RT_PHOTON_SLOTS) {
                        float *o = out + 9 * stored;
                        o[0] = nh.p.x; o[1] = nh.p.y; o[2] = nh.p.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = c.x; o[7] = c.y; o[8] = c.z;
                        stored++;
                    }
                    bounce--;
                    h = nh;
                }
            }
        }
    }
    A.count[i] = stored | (hits << 16);
}

void rtk_launch_photon_trace(hipStream_t st, const DevScene &S, unsigned long long first_attempt, uint32_t n_attempts,
                             uint32_t seed, int max_bounce, float *out, uint32_t *count, int mode)
{
    PhotonArgs A; A.first_attempt = first_attempt; A.n_attempts = n_attempts; A.seed = seed; A.max_bounce = max_bounce; A.out = out; A.count = count; A.mode = mode;
    hipLaunchKernelGGL(k_photon_trace, dim3((n_attempts + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, S, A);
}

// ------------------------------------------------------------------------------------------------
// K5: PhotonMap::EstimateIrradiance<k>(irr, dir, radius, pos, &N, 1, CONSTANT)
// (FIN/include/cyPhotonMap.h:288-336, LocatePhotons :365-440).
//
// The reference walks its heap-ordered kd-tree recursively per query and keeps the k nearest
// accepted photons (inside the radius, photonDir.N < 0) in a max-heap, shrinking the search radius
// once the heap is full; the estimate only needs
//   sum of power, sum of dir*maxPower over that set, and r_k^2 (= radius^2 while at most k photons
//   qualify, else the k-th smallest squared distance).
// Here each wavefront serves 64 queries at a time.
//   Phase A, one query per lane: a stackless walk of the complete binary tree of leaf boxes lists
//     the leaves within the query's CURRENT trial radius (ids in LDS).
//   Phase B, the whole wave per query: every listed leaf is one coalesced 64-photon read
//     (lane = photon).  Pass 1 counts the accepted photons into a 256-bin histogram of a 24-bit
//     fixed-point distance key (LDS atomics) while summing all of them.  If the trial radius is
//     smaller than the requested one and at most k photons qualified, the query is retried with a
//     larger radius predicted from the count (photons lie on surfaces: count ~ r^2); a trial that
//     finds MORE than k is exact, because the k nearest all lie inside it.  If more than k qualify,
//     the bin holding the k-th is located with a wave scan, pass 2 sums the bins below it and
//     collects that bin (<= 64 entries, else one more 8-bit level) for an exact rank selection.
// Sums are per-lane partials combined by a fixed butterfly: deterministic.
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    DevPhotonMap pm;
    const float4 *qa, *qb, *qc;      // query queue
    const uint32_t *count_ptr;       // number of queries (device)
    uint32_t *next_batch;            // zero at launch, RT_CTR_STRIDE apart: [seg] = batches handed out of XCD segment seg (8), [8] = mask of the segments used up
    uint32_t count_cap;
    int k; float radius;
    float *sample_rgb;               // mode 0: atomicAdd w * irr * max(0, N.(-dir)) into the slot
    float *out_irr, *out_dir;        // mode 1: write irr[3], dir[3] per query (rt_estimate_irradiance)
    int mode;
    unsigned long long *stats;
    float *cell_rk2;                 // per density-grid cell: the k-th squared distance of the last query answered there (0 = none yet); may be NULL
};

// Wave-wide inclusive scans on the DPP path (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions, no LDS
// crossbar traffic (__shfl is ds_bpermute: an LDS round trip per step).  All 64 lanes must be active.
// Lanes without a source keep `old` = the identity.  Fixed order => deterministic float sums.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t identity, uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, CTRL, ROW_MASK, 0xF, false);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
__device__ __forceinline__ float wave_scan_add(float x)
{
    x += dpp_f<DPP_ROW_SHR(1), 0xF>(0.0f, x); x += dpp_f<DPP_ROW_SHR(2), 0xF>(0.0f, x);
    x += dpp_f<DPP_ROW_SHR(4), 0xF>(0.0f, x); x += dpp_f<DPP_ROW_SHR(8), 0xF>(0.0f, x);
    x += dpp_f<DPP_ROW_BCAST15, 0xA>(0.0f, x); x += dpp_f<DPP_ROW_BCAST31, 0xC>(0.0f, x);
    return x;
}
__device__ __forceinline__ uint32_t wave_scan_add_u(uint32_t x)
{
    x += dpp_u<DPP_ROW_SHR(1), 0xF>(0u, x); x += dpp_u<DPP_ROW_SHR(2), 0xF>(0u, x);
    x += dpp_u<DPP_ROW_SHR(4), 0xF>(0u, x); x += dpp_u<DPP_ROW_SHR(8), 0xF>(0u, x);
    x += dpp_u<DPP_ROW_BCAST15, 0xA>(0u, x); x += dpp_u<DPP_ROW_BCAST31, 0xC>(0u, x);
    return x;
}
__device__ __forceinline__ float wave_scan_max0(float x)          // x >= 0
{
    x = fmaxf(x, dpp_f<DPP_ROW_SHR(1), 0xF>(0.0f, x)); x = fmaxf(x, dpp_f<DPP_ROW_SHR(2), 0xF>(0.0f, x));
    x = fmaxf(x, dpp_f<DPP_ROW_SHR(4), 0xF>(0.0f, x)); x = fmaxf(x, dpp_f<DPP_ROW_SHR(8), 0xF>(0.0f, x));
    x = fmaxf(x, dpp_f<DPP_ROW_BCAST15, 0xA>(0.0f, x)); x = fmaxf(x, dpp_f<DPP_ROW_BCAST31, 0xC>(0.0f, x));
    return x;
}
// totals: lane 63 of the inclusive scan, as a scalar
__device__ __forceinline__ float wave_sum(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_add(x)), 63)); }
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add_u(x), 63); }
__device__ __forceinline__ float wave_max0(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_max0(x)), 63)); }

// Six wave totals at once.  v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even rows of TWO
// registers, so one swap + one add folds two values at a time: after the 32-lane and the 16-lane fold four values share
// one register (a row of 16 lanes each), and the last four steps (row_shr 8, 4, 2, 1) run on two registers instead of
// six: 25 vector instructions instead of 48 for six separate scans.  Fixed order => deterministic.
__device__ __forceinline__ void fold32(float a, float b, float &ab)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    ab = __uint_as_float(r[0]) + __uint_as_float(r[1]);          // lanes 0-31: a folded, lanes 32-63: b folded
}
__device__ __forceinline__ void fold16(float x, float y, float &xy)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    xy = __uint_as_float(r[0]) + __uint_as_float(r[1]);          // rows 0..3: x.lo, y.lo, x.hi, y.hi folded to 16 lanes
}
__device__ __forceinline__ float row_total(float x)               // lane 15 of every row: the row's sum
{
    x += dpp_f<DPP_ROW_SHR(8), 0xF>(0.0f, x); x += dpp_f<DPP_ROW_SHR(4), 0xF>(0.0f, x);
    x += dpp_f<DPP_ROW_SHR(2), 0xF>(0.0f, x); x += dpp_f<DPP_ROW_SHR(1), 0xF>(0.0f, x);
    return x;
}
__device__ __forceinline__ void wave_sum6(float v0, float v1, float v2, float v3, float v4, float v5, float out[6])
{
    float s01, s23, s45, t0123, t45;
    fold32(v0, v1, s01); fold32(v2, v3, s23); fold32(v4, v5, s45);
    fold16(s01, s23, t0123);          // rows: v0, v2, v1, v3
    fold16(s45, s45, t45);            // rows: v4, v4, v5, v5
    t0123 = row_total(t0123); t45 = row_total(t45);
    out[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0123), 15));
    out[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0123), 31));
    out[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0123), 47));
    out[3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0123), 63));
    out[4] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t45), 15));
    out[5] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t45), 47));
}

// value of lane l (wave-uniform l) as a scalar: v_readlane, no LDS traffic, result lives in an SGPR
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t lane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// boxes are two aligned 16-byte words (lo.xyz, -), (hi.xyz, -): one visit = two dwordx4 loads
__device__ __forceinline__ float box_dist2(const float4 *b, float px, float py, float pz)
{
    RT_FP_CONTRACT
    const float4 lo = b[0], hi = b[1];
    const float dx = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// LDS hand-off between lanes of ONE wavefront: LDS operations of a wave complete in issue order, so
// only the compiler has to be kept from reordering, plus a wait for outstanding LDS returns.
__device__ __forceinline__ void wave_sync()
{
    // "wavefront" scope: the hardware already executes one wave's LDS instructions in issue order, so a write by one lane is
    // seen by a later read of another lane of the SAME wave without waiting for anything; the fences only keep the compiler
    // from moving LDS accesses across this point.  ("workgroup" scope made every one of the ~10 hand-offs per query an
    // s_waitcnt vmcnt(0) lgkmcnt(0): it also drained the photon loads in flight.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GatherLds {
    uint16_t leaves[RT_GATHER_BATCH][RT_LEAFLIST_CAP];   // per query (lane) leaf ids
    uint32_t subs[RT_SUBLIST_CAP + RT_SUBS_PER_STEP];    // the current query's sub-leaf ids, padded to whole steps with the dummy sub-leaf
    union alignas(16) {                                  // never live at the same time:
        uint32_t hist[256];                              //   the distance-key histogram while the k-th photon's bin is located
        struct { float sel_d[64]; uint32_t sel_i[64]; uint32_t sel_n; };   //   then that bin's photons for the exact rank selection
    };
    // everything pass 1 read about a photon whose distance lies in the band around the predicted k-th one,
    // so that the exact selection does not have to read the leaves a second time
    float4   ring_a[RT_GATHER_RING];      // d2, dir.x, dir.y, dir.z
    float2   ring_b[RT_GATHER_RING];      // max power, colour bytes
};

// Color24 -> Color (cyColor.h): byte / 255.0f, correctly rounded, without the ~10-instruction IEEE division and without
// a table: 1/255 as a two-term constant, r_hi = RN(1/255) and r_lo = RN(1/255 - r_hi); fma(c, r_hi, RN(c * r_lo)) adds the
// exact product c * r_hi to a correction that is itself good to 2^-48 of the result, and lands on the correctly rounded
// quotient for every byte value (all 256 checked exactly, in rational arithmetic: tests/test_host.py; on the device:
// test_gpu_parity.py::test_irradiance_single_photon_colour_bytes).  Two instructions; c * RN(1/255) alone is wrong for
// 121 of the 256 bytes, and the Newton form used before took three.
__device__ __forceinline__ float byte_over_255(uint32_t c)
{
    const float r_hi = 0x1.010102p-8f, r_lo = -0x1.fdfdfep-33f;
    const float x = (float)c;
    return __fmaf_rn(x, r_hi, x * r_lo);
}

// one lane's photon of one sub-leaf against one query (the test of LocatePhotons :383-392).  d2 is the squared distance
// for a photon that faces the surface and +infinity for one that does not (or for an empty slot, whose position is
// 3e38): "accepted" is then the single compare d2 < rq2, and every narrower test (below t_lo, inside the band) is one
// compare as well -- a compare's lane mask is its ballot, while a ballot of a combined condition costs two more
// vector instructions, on the unit this kernel is bound by.
struct Cand { float d2; float4 pa, pb; };
struct GatherQuery { float px, py, pz, nx, ny, nz, rq2, kscale; };
__device__ __forceinline__ Cand make_cand(float4 pa, float4 pb, const GatherQuery &Q)
{
    RT_FP_CONTRACT
    Cand c;
    c.pa = pa; c.pb = pb;
    const float dfx = pa.x - Q.px, dfy = pa.y - Q.py, dfz = pa.z - Q.pz;       // dif = p.position - np.pos
    const float d2 = dfx * dfx + dfy * dfy + dfz * dfz;                         // LengthSquared
    const bool away = (pa.w * Q.nx + pb.x * Q.ny + pb.y * Q.nz) >= 0;           // dir.N >= 0 rejects
    c.d2 = away ? __builtin_inff() : d2;                                        // dist2 < dist2[0] is tested by the caller
    return c;
}
// 24-bit fixed-point distance key of an ACCEPTED photon: kscale = 16777000 / rq2, so d2 < rq2 gives at most
// 16777000 * (1 + 2^-22) < 2^24 whatever the roundings
__device__ __forceinline__ uint32_t cand_key(const Cand &c, const GatherQuery &Q) { return (uint32_t)(c.d2 * Q.kscale); }

// Visit every photon slot of the n_sub sub-leaves listed in LDS (ids[]; padded to whole steps with the dummy sub-leaf,
// whose slots are all empty), 64 / RT_SUB_PHOTONS sub-leaves per step (lanes 0-31 the first, 32-63 the second):
// f(candidate, slot) is called wave-uniformly (all 64 lanes) so it may use ballots.  A lane's share of a step is one LDS
// read (its sub-leaf id), one shift-or (the byte offset, 32 bits) and two coalesced 16-byte loads from scalar bases --
// nothing else is fetched per photon.  The loads of step it+1 are issued before step it is processed, so a wave
// always has a step in flight while it works: measured on MI355X the un-pipelined version spent 78 % of its wave
// cycles parked on s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES).
template <class F>
__device__ __forceinline__ void scan_subleaves(const DevPhotonMap &pm, const uint32_t *ids, uint32_t n_sub, int lane,
                                               const GatherQuery &Q, F &&f)
{
    if (n_sub == 0) return;
    const uint32_t n_iter = (n_sub + RT_SUBS_PER_STEP - 1u) / RT_SUBS_PER_STEP;
    const uint32_t *mine = ids + (uint32_t)lane / RT_SUB_PHOTONS;           // which of a step's sub-leaves this lane reads
    const uint32_t lane_off = ((uint32_t)lane % RT_SUB_PHOTONS) * 16u;
    const char *pa = (const char *)pm.pa, *pb = (const char *)pm.pb;
    auto ld = [&](uint32_t it, float4 &a, float4 &b, uint32_t &slot) {
        const uint32_t off = (mine[RT_SUBS_PER_STEP * it] * (RT_SUB_PHOTONS * 16u)) | lane_off;   // < 2^32: checked at upload
        slot = off;
        a = *(const float4 *)(pa + off);
        b = *(const float4 *)(pb + off);
    };
    // two register sets used alternately, each refilled right after it was consumed; the reload index
    // is clamped instead of branched over (the last step may be fetched twice) so that neither set
    // is a loop-carried copy of the other
    float4 a0, b0, a1, b1;
    uint32_t s0, s1;
    ld(0u, a0, b0, s0);
    uint32_t it = 0;
    for (; it + 1 < n_iter; it += 2) {
        ld(it + 1, a1, b1, s1);
        f(make_cand(a0, b0, Q), s0);
        ld(min(it + 2, n_iter - 1), a0, b0, s0);
        f(make_cand(a1, b1, Q), s1);
    }
    if (it < n_iter) f(make_cand(a0, b0, Q), s0);
}

#ifndef RT_GATHER_WAVES_PER_EU
#define RT_GATHER_WAVES_PER_EU 5     // 96 registers: five waves per SIMD is what the LDS footprint allows too
#endif
__attribute__((amdgpu_waves_per_eu(RT_GATHER_WAVES_PER_EU, RT_GATHER_WAVES_PER_EU)))
__global__ __launch_bounds__(64 * RT_GATHER_WAVES) void k_gather(GatherArgs G)
{
    __shared__ GatherLds lds_all[RT_GATHER_WAVES];
    GatherLds &L = lds_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    uint32_t nq = *G.count_ptr;
    if (nq > G.count_cap) nq = G.count_cap;
    if (nq == 0) return;                                 // most chunks of a frame see no photon query at all
    const uint32_t n_leaves = G.pm.n_leaves;
    const uint32_t n_sub_total = n_leaves * RT_LEAF_SUBS;
    const float r2 = G.radius * G.radius;
    const uint32_t K = (uint32_t)G.k;
    unsigned long long visited = 0;
    uint32_t n_rounds = 0, n_slow = 0, n_reads = 0;       // wave-uniform tallies
    float pred_rk2 = 0.0f;                                // k-th squared distance of this wave's previous query (a hint only)

    // batches of RT_GATHER_BATCH queries are handed out dynamically (one atomic per batch): query cost varies by
    // two orders of magnitude with the local photon density, so a static split leaves a long tail
    const uint32_t n_batches = (nq + (uint32_t)RT_GATHER_BATCH - 1u) / (uint32_t)RT_GATHER_BATCH;
    const float guess_c = RT_GATHER_GUESS * (float)K * G.pm.cell * G.pm.cell / (float)M_PI;

    // XCD affinity: the queue is in sample order, so neighbouring batches look up neighbouring points and read the
    // same sub-leaves.  The queue is cut into eight contiguous segments, one per XCD: the waves of an XCD (640 of
    // them) then work inside a narrow window of the queue at any time and share its photons through their XCD's
    // 4 MB L2 instead of each XCD streaming every window's photons from the Infinity Cache.  A wave whose segment
    // is used up takes batches from the next ones (query cost varies 100x: no static split).  Speed only: any
    // assignment gives the same results.
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const uint32_t seg_len = (n_batches + 7u) / 8u;
    uint32_t seg = xcc;
    // A wave that finds a segment used up says so in a mask the others READ before they try it: without it every wave ends with
    // eight failing atomics -- 41 000 of them queueing at the memory side (device-scope atomics are executed there, one after the
    // other per channel) while nothing else is left to do.  The counters sit RT_CTR_STRIDE apart for the same reason (rt_dev.h).
    // Measured and not kept (r4, profiles/r04_experiments.json): the last batches of a segment handed out as 8-query units, a
    // segment handed out from its end.
    for (;;) {
        const uint32_t seg_first = seg * seg_len;
        const uint32_t seg_size = seg_first >= n_batches ? 0u : min(seg_len, n_batches - seg_first);
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(G.next_batch + seg * RT_CTR_STRIDE, 1u);
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (got >= seg_size) {                               // this segment is finished: move on to one that is not known to be, or stop
            uint32_t *const done_mask = G.next_batch + 8 * RT_CTR_STRIDE;
            uint32_t done = 0;                               // (what this wave marked earlier is in the mask it reads: same address, program order)
            if (lane == 0) {
                done = __hip_atomic_load(done_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!((done >> seg) & 1u)) atomicOr(done_mask, 1u << seg);
            }
            done = ((uint32_t)__builtin_amdgcn_readfirstlane((int)done) | (1u << seg)) & 255u;
            if (done == 255u) break;
            const uint32_t rot = ((done ^ 255u) | ((done ^ 255u) << 8)) >> (seg + 1u);     // segments still open, seen from seg + 1
            seg = (seg + 1u + ((uint32_t)__ffs((int)rot) - 1u)) & 7u;
            continue;
        }
        const uint32_t qbase = (seg_first + got) * (uint32_t)RT_GATHER_BATCH;
        const uint32_t qi = qbase + lane;
        const bool have = lane < RT_GATHER_BATCH && qi < nq;
        float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0), c = make_float4(0, 0, 0, 0);
        if (have) { a = G.qa[qi]; b = G.qb[qi]; c = G.qc[qi]; }
        bool pending = have;
        float f_pr = 0, f_pg = 0, f_pb = 0, f_dx = 0, f_dy = 0, f_dz = 0, f_area = -1.0f;     // a finished query's sums, in its lane
        bool finish = false;
        // first trial radius from the density grid: about RT_GATHER_GUESS * k photons expected inside (count ~ r^2
        // on a surface through a cell of side h: c photons per h^2)
        float r2cur = r2;
        float cell_pred = 0.0f;                              // this lane's query: what its grid cell remembers
        uint32_t cell_index = 0;
        uint32_t walk_start = 1;                             // where this lane's query starts its tree walk (DevPhotonMap::cell_start)
        if (have && n_leaves > 1) {
            const float fx = (a.x - G.pm.grid_min[0]) * G.pm.inv_cell, fy = (a.y - G.pm.grid_min[1]) * G.pm.inv_cell, fz = (a.z - G.pm.grid_min[2]) * G.pm.inv_cell;
            const int gx = min(max((int)fx, 0), G.pm.grid_dim[0] - 1);
            const int gy = min(max((int)fy, 0), G.pm.grid_dim[1] - 1);
            const int gz = min(max((int)fz, 0), G.pm.grid_dim[2] - 1);
            cell_index = (uint32_t)(((size_t)gz * G.pm.grid_dim[1] + gy) * G.pm.grid_dim[0] + gx);
            // only for a point that really lies in its cell (points outside the photons' bounding box are clamped to the rim)
            const bool in_grid = fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && (int)fx == gx && (int)fy == gy && (int)fz == gz;
            if (G.pm.cell_start && in_grid && G.radius <= G.pm.start_radius) walk_start = G.pm.cell_start[cell_index];
            const uint32_t cnt = G.pm.grid[cell_index];
            r2cur = fminf(fmaxf(guess_c / (float)(cnt > 0u ? cnt : 1u), r2 * 1.0e-4f), r2);
            if (G.cell_rk2) {
                cell_pred = G.cell_rk2[cell_index];
                // a cell that has seen a query also knows a better first radius than the density estimate: a little above its k-th distance
                if (cell_pred > 0.0f) r2cur = fminf(fmaxf(cell_pred * RT_GATHER_CELL_GUESS, r2 * 1.0e-4f), r2);
                else if (cell_pred < 0.0f) r2cur = r2;       // "sparse here": the full radius at once
            }
        }

        while (ballot64(pending)) {
            // ---------------- phase A: pending lanes list the leaves inside their trial radius ----
            // Two lanes walk for one query: lane l and lane l + 32 hold the same point and radius, keep the same walk state
            // and split the box tests of every visit between them (grandchildren 0-1 / 2-3, child 0 / 1); one
            // v_permlane32_swap per visit gives both the combined result.  The queries sit in lanes 0-31 only (32 per
            // batch), so the upper half of the wave would otherwise idle through the phase.
            static_assert(RT_GATHER_BATCH == 32, "phase A pairs lane l with lane l + 32");
            const bool upper = lane >= 32;
            auto from_lower = [](float v) { return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false)[0]); };
            const float wx = from_lower(a.x), wy = from_lower(a.y), wz = from_lower(a.z), wr2 = from_lower(r2cur);
            const uint32_t wstart = __builtin_amdgcn_permlane32_swap(walk_start, walk_start, false, false)[0];
            const bool walking = ((uint32_t)ballot64(pending) >> (lane & 31)) & 1u;
            auto both_halves = [](uint32_t mine, int bits) {       // my half's result bits -> lower half's | upper half's << bits, in every lane
                const auto r = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                return r[0] | (r[1] << bits);
            };
            uint32_t nl = 0;
            auto list_leaf = [&](uint32_t leaf) {
                if (!upper && nl < RT_LEAFLIST_CAP) L.leaves[lane][nl] = (uint16_t)leaf;
                nl++;
            };
            if (walking && n_leaves) {
                // Depth-first, left to right (ascending leaf ids).  Phase A waits for a chain of dependent box reads, so a
                // visit reads as much as one aligned line pair gives: the boxes of all four GRANDCHILDREN of an internal node
                // (heap order: nodes 4n..4n+3, 128 contiguous bytes) -- a grandchild the ball cuts implies its parent is cut,
                // so the level between needs no test of its own.  Which grandchildren are still to be visited is kept as a
                // 4-bit mask per pair of levels; no box is fetched twice.  An odd last level is a two-child visit.
                if (box_dist2(G.pm.tbox + 2, wx, wy, wz) < wr2) {
                    if (n_leaves == 1) list_leaf(0u);
                    else {
                        uint32_t node = wstart;                 // current internal node (the root, or the cell's start node: even depth) ...
                        uint32_t pd = (31u - (uint32_t)__clz((int)node)) >> 1;      // ... and its pair-depth (tree depth = 2 * pd)
                        uint32_t todo_mask = 0;                 // nibble pd: grandchildren of the path's node at pair-depth pd still to visit (at most 65536 leaves: 8 nibbles)
                        for (;;) {
                            if (4u * node < 2u * n_leaves && 2u * node < n_leaves) {
                                // grandchildren exist (they are internal nodes, or the leaves themselves)
                                const float4 *gb = G.pm.tbox + 8 * (size_t)node + (upper ? 4 : 0);      // boxes of 4*node .. 4*node + 3: two of them for me
                                uint32_t m2 = 0;
                                if (box_dist2(gb, wx, wy, wz) < wr2) m2 |= 1u;
                                if (box_dist2(gb + 2, wx, wy, wz) < wr2) m2 |= 2u;
                                const uint32_t m4 = both_halves(m2, 2);
                                if (4u * node >= n_leaves) {     // the grandchildren are leaves
                                    for (int i = 0; i < 4; i++) if ((m4 >> i) & 1u) list_leaf(4u * node + (uint32_t)i - n_leaves);
                                } else if (m4) {
                                    const uint32_t i = (uint32_t)__ffs((int)m4) - 1u;
                                    todo_mask |= (m4 & ~(1u << i)) << (4u * pd);
                                    node = 4u * node + i; pd++;
                                    continue;
                                }
                            } else {
                                // one level left: the children are leaves
                                const float4 *cb = G.pm.tbox + 4 * (size_t)node + (upper ? 2 : 0);
                                const uint32_t m2 = both_halves(box_dist2(cb, wx, wy, wz) < wr2 ? 1u : 0u, 1);
                                if (m2 & 1u) list_leaf(2u * node - n_leaves);
                                if (m2 & 2u) list_leaf(2u * node + 1u - n_leaves);
                            }
                            if (!todo_mask) break;
                            const uint32_t bit = 31u - (uint32_t)__clz((int)todo_mask);            // deepest pair-depth with work left
                            const uint32_t d = bit >> 2;
                            const uint32_t nib = (todo_mask >> (4u * d)) & 15u;
                            const uint32_t i = (uint32_t)__ffs((int)nib) - 1u;                    // its leftmost grandchild not yet visited
                            todo_mask &= ~(1u << (4u * d + i));
                            node = ((node >> (2u * (pd - d))) << 2) + i;                          // that ancestor's grandchild i
                            pd = d + 1u;
                        }
                    }
                }
            }
            wave_sync();
            // ---------------- phase B: the wave takes the pending queries one by one --------------
            unsigned long long todo = ballot64(pending);
            while (todo) {
                const int q = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                GatherQuery Q;
                Q.px = lane_f(a.x, q); Q.py = lane_f(a.y, q); Q.pz = lane_f(a.z, q);
                Q.nx = lane_f(a.w, q); Q.ny = lane_f(b.x, q); Q.nz = lane_f(b.y, q);
                Q.rq2 = lane_f(r2cur, q);
                Q.kscale = 16777000.0f / Q.rq2;            // 24-bit fixed-point distance key (see cand_key)
                const float rq2 = Q.rq2;
                const uint32_t qnl = lane_u(nl, q);
                const bool final_round = rq2 >= r2;
                const bool slow = qnl > RT_LEAFLIST_CAP;   // the LDS list overflowed
                // ---- the query's sub-leaves: RT_LEAF_SUBS boxes per listed leaf, tested 64 at a time, ids compacted in LDS ----
                // (a leaf holds 128 photon slots; most query balls cut only part of one: testing its four 32-slot
                // sub-boxes examines about a fifth fewer photons than reading whole 64-slot leaves did)
                const uint32_t dummy_sub = n_sub_total;    // one more sub-leaf after the real ones, every slot empty
                uint32_t n_sub = 0;
                wave_sync();                               // the previous query's passes are done with L.subs
                if (!slow) {
                    for (uint32_t base = 0; base < qnl * RT_LEAF_SUBS; base += 64u) {
                        const uint32_t e = base + (uint32_t)lane;
                        const bool have_e = (e / RT_LEAF_SUBS) < qnl;
                        const uint32_t sub = have_e ? (uint32_t)L.leaves[q][e / RT_LEAF_SUBS] * RT_LEAF_SUBS + (e % RT_LEAF_SUBS) : 0u;
                        const float bd = have_e ? box_dist2(G.pm.sbox + 2 * (size_t)sub, Q.px, Q.py, Q.pz) : __builtin_inff();
                        const unsigned long long m = ballot64(bd < rq2);
                        if (bd < rq2) L.subs[n_sub + lanes_below(m)] = sub;
                        n_sub += (uint32_t)__popcll(m);
                    }
                    if (lane < RT_SUBS_PER_STEP) L.subs[n_sub + lane] = dummy_sub;
                    wave_sync();
                } else {
                    // list too long for LDS: every pass walks ALL sub-leaf boxes instead (64 per step) -- no list is kept
                    for (uint32_t base = 0; base < n_sub_total; base += 64u) {
                        const uint32_t sub = base + (uint32_t)lane;
                        const float bd = sub < n_sub_total ? box_dist2(G.pm.sbox + 2 * (size_t)sub, Q.px, Q.py, Q.pz) : __builtin_inff();
                        n_sub += (uint32_t)__popcll(ballot64(bd < rq2));
                    }
                }
                n_rounds++; n_slow += slow ? 1u : 0u; n_reads += n_sub;
                // run one pass over the query's sub-leaves
                auto for_each = [&](auto &&f) {
                    if (!slow) { scan_subleaves(G.pm, L.subs, n_sub, lane, Q, f); return; }
                    for (uint32_t base = 0; base < n_sub_total; base += 64u) {
                        const uint32_t sub = base + (uint32_t)lane;
                        const float bd = sub < n_sub_total ? box_dist2(G.pm.sbox + 2 * (size_t)sub, Q.px, Q.py, Q.pz) : __builtin_inff();
                        const unsigned long long m = ballot64(bd < rq2);
                        if (!m) continue;
                        wave_sync();
                        if (bd < rq2) L.subs[lanes_below(m)] = sub;
                        const uint32_t cnt = (uint32_t)__popcll(m);
                        if (lane < RT_SUBS_PER_STEP) L.subs[cnt + lane] = dummy_sub;
                        wave_sync();
                        scan_subleaves(G.pm, L.subs, cnt, lane, Q, f);
                    }
                };
                float s_pr = 0, s_pg = 0, s_pb = 0, s_dx = 0, s_dy = 0, s_dz = 0;   // pass 1: sums over ALL candidates
                // sum of power (GetPower = Color24 -> Color times power) and of dir * maxPower for one photon
                // (branch-free variant: take == false adds exact zeros; measured slower than the branch)
                auto accumulate5 = [&](float dirx, float diry, float dirz, float maxp, uint32_t cbits, bool take) {
                    RT_FP_CONTRACT
                    if (!take) return;
                    const float mp = take ? maxp : 0.0f;
                    s_pr += byte_over_255(cbits & 255u) * mp; s_pg += byte_over_255((cbits >> 8) & 255u) * mp; s_pb += byte_over_255((cbits >> 16) & 255u) * mp;
                    s_dx += dirx * mp; s_dy += diry * mp; s_dz += dirz * mp;
                };
                auto accumulate = [&](const float4 &pa, const float4 &pb, bool take) { accumulate5(pa.w, pb.x, pb.y, pb.z, __float_as_uint(pb.w), take); };

                uint32_t M = 0;                            // accepted photons (wave-uniform: popcount of the ballots)
                float area_d2 = rq2;
                bool done_plain = false;
                // "Sparse here": the cell's last query found no more than k photons inside the FULL radius.  Then all accepted
                // photons count and dist2[0] stays radius^2 (cyPhotonMap.h:309-326): one plain pass sums them -- no histogram, no ring,
                // no selection.  If more than k turn up after all, the sums are dropped and the normal path below does the query.
                if (final_round && lane_f(cell_pred, q) < 0.0f) {
                    for_each([&](const Cand &cd, uint32_t) {
                        M += (uint32_t)__popcll(ballot64(cd.d2 < rq2));
                        accumulate(cd.pa, cd.pb, cd.d2 < rq2);
                    });
                    visited += n_sub;
                    if (M <= K) done_plain = true;
                    else { M = 0; s_pr = s_pg = s_pb = s_dx = s_dy = s_dz = 0; n_reads += n_sub; }
                }
                if (!done_plain) {
                *(uint4 *)&L.hist[4 * lane] = make_uint4(0u, 0u, 0u, 0u);     // the 256 bins in one 16-byte store per lane
                wave_sync();
                // pass 1: count + histogram of every accepted photon.  Photons closer than t_lo (safely inside the
                // k nearest if the prediction holds) are summed right away; those between t_lo and t_hi, the band
                // the k-th distance is expected in, are parked in the LDS ring with all their data.
                // The histogram bin is the top 8 bits of the 24-bit key: (uint)(d2 * kscale) >> 16 == (uint)(d2 * (kscale / 65536)),
                // the scaling by a power of two being exact.
                const float kscale_bin = Q.kscale * (1.0f / 65536.0f);
                { const float cp = lane_f(cell_pred, q); if (cp > 0.0f) pred_rk2 = cp; }
                const float pk = (pred_rk2 > 0.0f && pred_rk2 < rq2) ? pred_rk2 : rq2 * (1.0f / RT_GATHER_GUESS);
                const float t_lo = RT_GATHER_BAND_LO * pk;
                const float t_hi = final_round ? rq2 : fminf(RT_GATHER_BAND_HI * pk, rq2);     // <= rq2: inside the band implies accepted
                uint32_t n_ring = 0;                       // wave-uniform (ballot popcounts)
                for_each([&](const Cand &cd, uint32_t) {
                    const unsigned long long m_ok = ballot64(cd.d2 < rq2);
                    const unsigned long long m_lo = ballot64(cd.d2 < t_lo);
                    const unsigned long long mr = ballot64(cd.d2 < t_hi) & ~m_lo;
                    M += (uint32_t)__popcll(m_ok);
                    if (cd.d2 < rq2) atomicAdd(&L.hist[(uint32_t)(cd.d2 * kscale_bin)], 1u);
                    accumulate(cd.pa, cd.pb, cd.d2 < t_lo);
                    if (mr) {
                        if (cd.d2 < t_hi && !(cd.d2 < t_lo)) {
                            const uint32_t at = n_ring + lanes_below(mr);
                            if (at < (uint32_t)RT_GATHER_RING) {
                                L.ring_a[at] = make_float4(cd.d2, cd.pa.w, cd.pb.x, cd.pb.y);
                                L.ring_b[at] = make_float2(cd.pb.z, cd.pb.w);
                            }
                        }
                        n_ring += (uint32_t)__popcll(mr);
                    }
                });
                visited += n_sub;
                if (!final_round && M <= K) {
                    // not enough inside the trial radius: grow it (count ~ r^2 on a surface) and retry
                    float grow = 1.5f * (float)K / (float)(M > 0 ? M : 1u);
                    grow = fminf(fmaxf(grow, 2.0f), 16.0f);
                    if (lane == q) r2cur = fminf(rq2 * grow, r2);
                    continue;
                }
                area_d2 = rq2;                             // dist2[0]; only reached with rq2 == r2 when M <= K
                if (M > K) {
                    // ---- locate the k-th smallest: refine 8 bits of the key per level --------------
                    uint32_t need = K;                     // rank (1-based) inside the current range
                    uint32_t prefix = 0;                   // key bits fixed so far
                    int shift = 16;                        // the level's digit = (key >> shift) & 255
                    uint32_t in_bin = 0;
                    for (;;) {
                        wave_sync();
                        const uint4 h4 = *(const uint4 *)&L.hist[4 * lane];
                        const uint32_t h0 = h4.x, h1 = h4.y, h2 = h4.z, h3 = h4.w;
                        const uint32_t mine = h0 + h1 + h2 + h3;
                        const uint32_t incl = wave_scan_add_u(mine);
                        const uint32_t excl = incl - mine;
                        const unsigned long long m = ballot64(incl >= need);
                        const int owner = __ffsll((long long)m) - 1;      // first lane whose range reaches `need`
                        uint32_t digit = 0, before = 0, cntb = 0;
                        if (lane == owner) {
                            const uint32_t cum = excl;
                            if (cum + h0 >= need) { digit = 4 * lane; before = cum; cntb = h0; }
                            else if (cum + h0 + h1 >= need) { digit = 4 * lane + 1; before = cum + h0; cntb = h1; }
                            else if (cum + h0 + h1 + h2 >= need) { digit = 4 * lane + 2; before = cum + h0 + h1; cntb = h2; }
                            else { digit = 4 * lane + 3; before = cum + h0 + h1 + h2; cntb = h3; }
                        }
                        digit = lane_u(digit, owner); before = lane_u(before, owner); cntb = lane_u(cntb, owner);
                        need -= before;
                        prefix |= digit << shift;
                        in_bin = cntb;
                        if (in_bin <= 64u || shift == 0) break;
                        // one more level: histogram of the next 8 bits over the photons inside this bin
                        shift -= 8;
                        wave_sync();
                        *(uint4 *)&L.hist[4 * lane] = make_uint4(0u, 0u, 0u, 0u);
                        wave_sync();
                        const uint32_t hi_mask = ~((1u << (shift + 8)) - 1u) & 0xFFFFFFu;
                        for_each([&](const Cand &cd, uint32_t) {
                            const uint32_t key = cand_key(cd, Q);
                            if (cd.d2 < rq2 && (key & hi_mask) == prefix) atomicAdd(&L.hist[(key >> shift) & 255u], 1u);
                        });
                    }
                    const uint32_t bin_mask = ~((1u << shift) - 1u) & 0xFFFFFFu;
                    wave_sync();                           // the histogram is dead from here on: its LDS now holds the selection
                    if (lane == 0) L.sel_n = 0;
                    wave_sync();
                    uint32_t tie_taken = 0;                // only used when in_bin > 64 (identical keys)
                    float tmax = 0.0f;
                    bool from_ring = false;
                    {
                        // The ring serves the selection when (1) it did not overflow, (2) the k-th photon's bin was
                        // resolved at the first level, (3) every photon below t_lo lies in an earlier bin (so all of
                        // them count) and (4) every photon of the k-th bin or earlier lies below t_hi (so it is either
                        // summed already or in the ring).  Keys are monotone in d2, which makes (3) and (4) exact.
                        const uint32_t bin_lo = (uint32_t)(t_lo * Q.kscale) >> 16, bin_hi = (uint32_t)(t_hi * Q.kscale) >> 16;
                        const uint32_t kbin = prefix >> 16;
                        from_ring = shift == 16 && in_bin <= 64u && n_ring <= (uint32_t)RT_GATHER_RING && bin_lo < kbin && (t_hi >= rq2 || kbin < bin_hi);
                    }
                    if (from_ring) {
                        for (uint32_t base = 0; base < n_ring; base += 64u) {
                            const uint32_t idx = base + (uint32_t)lane;
                            const bool have = idx < n_ring;
                            const float4 ra = have ? L.ring_a[idx] : make_float4(3.0e38f, 0, 0, 0);
                            const float2 rb = have ? L.ring_b[idx] : make_float2(0, 0);
                            const uint32_t kb = (uint32_t)(ra.x * Q.kscale) & bin_mask;
                            const bool take = have && kb < prefix;
                            const bool inb = have && kb == prefix;
                            const unsigned long long mb = ballot64(inb);
                            if (mb) {
                                const uint32_t sbase = lane_u(L.sel_n, 0);
                                if (inb) {
                                    const uint32_t at = sbase + lanes_below(mb);
                                    if (at < 64u) { L.sel_d[at] = ra.x; L.sel_i[at] = idx; }
                                }
                                wave_sync();
                                if (lane == 0) L.sel_n = sbase + (uint32_t)__popcll(mb);
                                wave_sync();
                            }
                            accumulate5(ra.y, ra.z, ra.w, rb.x, __float_as_uint(rb.y), take);
                        }
                    }
                    if (!from_ring) {
                    n_reads += n_sub;
                    // ---- pass 2: sum everything below the bin, collect the bin, select `need` of it ----
                    s_pr = s_pg = s_pb = s_dx = s_dy = s_dz = 0;
                    for_each([&](const Cand &cd, uint32_t s) {
                        const bool ok = cd.d2 < rq2;
                        const uint32_t kb = cand_key(cd, Q) & bin_mask;
                        bool take = ok && kb < prefix;
                        const bool inb = ok && kb == prefix;
                        const unsigned long long mb = ballot64(inb);
                        if (in_bin <= 64u) {
                            if (mb) {
                                uint32_t base = 0;
                                const int leader = __ffsll((long long)mb) - 1;
                                if (lane == leader) { base = L.sel_n; L.sel_n = base + (uint32_t)__popcll(mb); }
                                base = lane_u(base, leader);
                                if (inb) {
                                    const uint32_t at = base + lanes_below(mb);
                                    if (at < 64u) { L.sel_d[at] = cd.d2; L.sel_i[at] = s; }
                                }
                            }
                        } else {
                            // more than 64 photons share all 24 key bits: take the first `need` in scan order
                            const uint32_t rank = tie_taken + lanes_below(mb);
                            if (inb && rank < need) { take = true; tmax = fmaxf(tmax, cd.d2); }
                            tie_taken += (uint32_t)__popcll(mb);
                        }
                        accumulate(cd.pa, cd.pb, take);
                    });
                    }
                    wave_sync();
                    if (in_bin <= 64u) {
                        // exact selection: rank by (d2, list position); take ranks < need
                        const uint32_t n_sel = lane_u(min(L.sel_n, 64u), 0);
                        const bool mine = (uint32_t)lane < n_sel;
                        const float md = mine ? L.sel_d[lane] : 3.0e38f;
                        uint32_t rank = 0;
                        for (uint32_t j = 0; j < n_sel; j++) {
                            const float od = L.sel_d[j];
                            rank += (od < md || (od == md && j < (uint32_t)lane)) ? 1u : 0u;
                        }
                        if (mine && rank < need) {
                            const uint32_t si = L.sel_i[lane];
                            if (from_ring) {
                                const float4 ra = L.ring_a[si];
                                const float2 rb = L.ring_b[si];
                                accumulate5(ra.y, ra.z, ra.w, rb.x, __float_as_uint(rb.y), true);
                            } else
                                accumulate(*(const float4 *)((const char *)G.pm.pa + si), *(const float4 *)((const char *)G.pm.pb + si), true);   // si: byte offset of the slot
                            tmax = md;
                        }
                    }
                    area_d2 = wave_max0(tmax);                        // np.dist2[0] = largest kept distance
                    pred_rk2 = area_d2;
                }
                else if (M > 0) {
                    // at most k inside the full radius: all of them count.  Pass 1 summed those below t_lo and, in the
                    // final round, parked every other one in the ring; if that overflowed, sum them with one more pass
                    if (n_ring <= (uint32_t)RT_GATHER_RING) {
                        for (uint32_t base = 0; base < n_ring; base += 64u) {
                            const uint32_t idx = base + (uint32_t)lane;
                            if (idx < n_ring) {
                                const float4 ra = L.ring_a[idx];
                                const float2 rb = L.ring_b[idx];
                                accumulate5(ra.y, ra.z, ra.w, rb.x, __float_as_uint(rb.y), true);
                            }
                        }
                    } else {
                        n_reads += n_sub;
                        s_pr = s_pg = s_pb = s_dx = s_dy = s_dz = 0;
                        for_each([&](const Cand &cd, uint32_t) { accumulate(cd.pa, cd.pb, cd.d2 < rq2); });
                    }
                }
                }
                // the query is done: its six sums and r_k^2 go to ITS lane; what follows from them (area, normalisation, the
                // weighted add into the sample) is the same scalar arithmetic for every query, so it is done for all the
                // queries a round finished at once, one per lane, after the loop -- not 64 lanes wide per query
                {
                    float t[6];
                    wave_sum6(s_pr, s_pg, s_pb, s_dx, s_dy, s_dz, t);
                    if (lane == q) {
                        f_pr = t[0]; f_pg = t[1]; f_pb = t[2]; f_dx = t[3]; f_dy = t[4]; f_dz = t[5];
                        f_area = M > 0 ? area_d2 : -1.0f;      // dist2[0] >= 0; negative: no photon at all
                        finish = true;
                        pending = false;
                    }
                }
            }
            if (finish) {
                float irr_r = f_pr, irr_g = f_pg, irr_b = f_pb, dx = f_dx, dy = f_dy, dz = f_dz;
                // remember the k-th distance for the next query of this cell (only when more than k qualified: f_area < r2)
                // (or that no more than k were inside the full radius: f_area is then radius^2, or negative without any photon)
                if (G.cell_rk2 && n_leaves > 1) {
                    if (f_area > 0.0f && f_area < r2) G.cell_rk2[cell_index] = f_area;
                    else if (r2cur >= r2) G.cell_rk2[cell_index] = -1.0f;
                }
                if (f_area >= 0.0f) {
                    const float area = (float)M_PI * f_area;               // :326
                    if (area > 0) { const float inv = 1.0f / area; irr_r *= inv; irr_g *= inv; irr_b *= inv; }
                    const float l = sqrtf(dx * dx + dy * dy + dz * dz);    // direction.Normalize() :334
                    dx /= l; dy /= l; dz /= l;
                }
                if (G.mode == 1) {
                    const size_t qq = (size_t)qi;
                    G.out_irr[3 * qq] = irr_r; G.out_irr[3 * qq + 1] = irr_g; G.out_irr[3 * qq + 2] = irr_b;
                    G.out_dir[3 * qq] = dx; G.out_dir[3 * qq + 1] = dy; G.out_dir[3 * qq + 2] = dz;
                } else {
                    // idr_Color += kd * photonrad * max(0, N.(-dir)) (FIN/main.cpp:701-704), times the ray weight
                    const float nx = a.w, ny = b.x, nz = b.y, wr = b.z, wg = b.w, wb = c.x;
                    const uint32_t slot = __float_as_uint(c.y);
                    float theta = nx * (-dx) + ny * (-dy) + nz * (-dz);
                    theta = theta > 0.0f ? theta : 0.0f;
                    float *dst = G.sample_rgb + 3 * (size_t)slot;
                    atomicAdd(dst, (wr * irr_r) * theta);
                    atomicAdd(dst + 1, (wg * irr_g) * theta);
                    atomicAdd(dst + 2, (wb * irr_b) * theta);
                }
                finish = false;
            }
            wave_sync();
        }
    }
    if (G.stats) {                                        // workgroup-uniform; every wave gets here (flush_counters says why it is done this way)
        __shared__ unsigned long long s_acc[4];
        if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
        __syncthreads();
        if (lane == 0 && visited) {
            atomicAdd(&s_acc[0], visited * (unsigned long long)RT_SUB_PHOTONS);
            atomicAdd(&s_acc[1], (unsigned long long)n_rounds);
            atomicAdd(&s_acc[2], (unsigned long long)n_slow);
            atomicAdd(&s_acc[3], (unsigned long long)n_reads * RT_SUB_PHOTONS / 32ull);     // in units of 32 slots = 1 KiB
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const int slot = threadIdx.x == 0 ? ST_PHOTONS_VISITED : threadIdx.x == 1 ? ST_GATHER_ROUNDS : threadIdx.x == 2 ? ST_GATHER_SLOW : ST_GATHER_LEAF_READS;
            const unsigned long long x = s_acc[threadIdx.x];
            if (x) atomicAdd(&G.stats[ST_AT(slot)], x);
        }
    }
    if (G.stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&G.stats[ST_AT(ST_PHOTON_QUERIES)], (unsigned long long)nq);
}

// ------------------------------------------------------------------------------------------------
// K6: per pixel, the tail of RenderPixel (FIN/main.cpp:273-338): hits-only average
// (averageColor :191-199), VariantOverThreshold (:164-189) gate for the second batch, gamma
// (powf(c, 1.0/gamma) :318-320), Color24 pack (cyColor.h:245-246), z of the last hit sample,
// sample-count byte (:312-315), background for all-miss pixels (:326-337).
//   phase 0: after the first batch (samples 0..min-1): either finalise or list the pixel
//   phase 1: after the second batch: finalise listed pixels with all samples
// ------------------------------------------------------------------------------------------------
struct ResolveArgs {
    DevCamera cam; DevTiles tiles;
    uint32_t q0, npix;
    int min_sample, max_sample;
    float threshold; float inv_gamma;
    int phase;
    float bg[3];
    DevScene S;                 // for the background map
    uint8_t *rgb8; float *z; uint8_t *count;
    // packed output (multi-GPU tile exchange): pixel q of this call's tile walk (tile-major, row-major inside
    // the tile) is ONE 8-byte record {r, g, b, z as 4 little-endian bytes, count} at packed + 8*q -- the very
    // buffer a rank contributes to the all-gather, written here coalesced instead of being re-packed afterwards
    uint2 *packed;
    int direct_mode;            // rt_shade_rays: no image, leave samples as they are
};

__device__ __forceinline__ uint8_t float_to_byte(float r)
{
    const float s = r * 255;
    if (!(s == s)) return 0;
    if (s <= -2147483648.0f) return 0;
    if (s >= 2147483647.0f) return 255;
    const int v = (int)s;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

#ifndef RT_RESOLVE_TILE
#define RT_RESOLVE_TILE 8            // samples per pixel staged through LDS at a time
#endif
__global__ __launch_bounds__(256) void k_resolve(DevWork W, ResolveArgs A)
{
    // A pixel's samples are contiguous (max_sample slots of 12 B), so a thread walking its own pixel reads 12 B
    // at a 768-B stride from its neighbours: measured 6.6x the useful bytes from HBM.  In phase 0 the 64 pixels of
    // a wave are consecutive, so the wave stages RT_RESOLVE_TILE samples of each of them through LDS with
    // coalesced reads (one 96-B run per pixel) and every lane then reads its own pixel's samples from LDS
    // (row stride 25 words: conflict-free).  The sums are taken in sample order, as before.
    __shared__ float s_tile[4][64 * (3 * RT_RESOLVE_TILE + 1)];
    float *tile = s_tile[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint32_t npix = A.phase == 1 ? W.counts[CNT_PIXLIST] : A.npix;
    const uint32_t npix_round = (npix + 63u) & ~63u;            // whole waves take part in the staging
    if (blockIdx.x == 0 && threadIdx.x == 0 && W.stats) {
        // the queues of this pass are final by now (same stream): remember how full they got, so that the host
        // can size them from what a scene really produces instead of the 2^bounce worst case
        uint32_t pr = 0;
        for (int l = 1; l < CNT_PHOTONQ; l++) pr = max(pr, W.counts[l]);
        atomicMax(&W.stats[ST_AT(ST_PEAK_RAYS)], (unsigned long long)pr);
        // both query queues are sized from this one figure (the caustic queue gets the photon queue's capacity)
        atomicMax(&W.stats[ST_AT(ST_PEAK_QUERIES)], (unsigned long long)max(W.counts[CNT_PHOTONQ], W.counts[CNT_CAUSTICQ]));
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix_round; i += gridDim.x * blockDim.x) {
        const uint32_t i0 = i - (uint32_t)lane;                  // first pixel of this wave
        const bool staged = A.phase == 0 && i0 + 64u <= npix;    // wave-uniform
        const bool in_range = i < npix;
        const uint32_t ql = in_range ? (A.phase == 1 ? W.pixel_list[i] : i) : 0u;
        int x = 0, y = 0;
        const bool valid = in_range && pixel_of(A.tiles, A.cam, A.q0 + ql, x, y);
        // packed mode: slots of a ragged tile that lie outside the image are part of the exchanged buffer -> zero
        if (A.packed && in_range && !valid && A.phase == 0) A.packed[(size_t)A.q0 + ql] = make_uint2(0u, 0u);
        if (!staged && !valid) continue;
        const size_t index = (size_t)y * A.cam.width + x;
        const float *rgb = W.sample_rgb + 3 * (size_t)ql * A.max_sample;
        const uint8_t *hitf = W.sample_hit + (size_t)ql * A.max_sample;
        const float *zs = W.sample_z + (size_t)ql * A.max_sample;
        const int ns = A.phase == 1 ? A.max_sample : A.min_sample;
        // the hit flags of the pixel's samples (bytes 0 / 1), 64 samples at a time: with rows of a multiple of 16 bytes they are read as
        // 16-byte words and packed into a 64-bit mask (four loads instead of one byte load per sample and pass)
        const bool packed_hits = (A.max_sample & 15) == 0;
        const int nblk = (ns + 63) >> 6;
        auto block_mask = [&](int b) -> unsigned long long {                  // samples [64 b, 64 b + 64) of this pixel, clipped to ns
            const int base = 64 * b, cnt = min(64, ns - base);
            unsigned long long m = 0;
            if (packed_hits) {
                const uint4 *h4 = (const uint4 *)(hitf + base);
                for (int t = 0; 16 * t < cnt; t++) {
                    const uint4 w = h4[t];
                    const uint32_t b16 = (((w.x & 0x01010101u) * 0x01020408u) >> 24) | ((((w.y & 0x01010101u) * 0x01020408u) >> 24) << 4) |
                                         ((((w.z & 0x01010101u) * 0x01020408u) >> 24) << 8) | ((((w.w & 0x01010101u) * 0x01020408u) >> 24) << 12);
                    m |= (unsigned long long)(b16 & 0xFFFFu) << (16 * t);
                }
            } else {
                for (int j = 0; j < cnt; j++) if (hitf[base + j]) m |= 1ull << j;
            }
            if (cnt < 64) m &= (1ull << cnt) - 1ull;
            return m;
        };
        int n = 0, last = -1;
        for (int b = 0; b < nblk; b++) { const unsigned long long m = block_mask(b); n += __popcll(m); if (m) last = 64 * b + 63 - __clzll((long long)m); }
        float hitz = 0;
        if (last >= 0) hitz = zs[last];
        // visit(r, g, b) for the hit samples in order, from LDS tiles (staged) or straight from memory
        auto for_hit_samples = [&](auto &&visit) {
            const float *wave_rgb = W.sample_rgb + 3 * (size_t)i0 * A.max_sample;
            for (int b = 0; b < nblk; b++) {
                const unsigned long long hm = block_mask(b);
                const int base = 64 * b, cnt = min(64, ns - base);
                if (staged) {
                    for (int t0 = 0; t0 < cnt; t0 += RT_RESOLVE_TILE) {
                        const int tn = min(RT_RESOLVE_TILE, cnt - t0);             // samples in this tile
                        const int run = 3 * tn;                                     // floats per pixel
                        __builtin_amdgcn_wave_barrier();
                        for (int k = lane; k < 64 * run; k += 64) {
                            const int pix = k / run, off = k - pix * run;
                            tile[pix * (3 * RT_RESOLVE_TILE + 1) + off] = wave_rgb[(size_t)pix * 3 * A.max_sample + 3 * (base + t0) + off];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const float *mine = tile + lane * (3 * RT_RESOLVE_TILE + 1);
                        for (int j = 0; j < tn; j++) if ((hm >> (t0 + j)) & 1ull) visit(mine[3 * j], mine[3 * j + 1], mine[3 * j + 2]);
                    }
                } else {
                    for (int j = 0; j < cnt; j++) if ((hm >> j) & 1ull) visit(rgb[3 * (base + j)], rgb[3 * (base + j) + 1], rgb[3 * (base + j) + 2]);
                }
            }
        };
        bool over = false;
        if (A.phase == 0 && A.min_sample != A.max_sample && (staged || n > 0)) {
            // VariantOverThreshold over the hit colours of the first batch
            const float ninverse = (float)(1.0 / (n > 0 ? n : 1));
            float sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
            for_hit_samples([&](float r, float g, float b) {
                const float t3[3] = {r, g, b};
                for (int c = 0; c < 3; c++) {
                    sum[c] += t3[c];
                    sq[c] = (float)((double)sq[c] + (double)t3[c] * (double)t3[c]);       // pow(float,int) -> double
                }
            });
            for (int c = 0; c < 3; c++) {
                const float avg = ninverse * sum[c];
                const float var = (float)((double)(sq[c] * ninverse) + (double)avg * (double)avg - (double)(2 * avg * ninverse * sum[c]));
                if (var > A.threshold) over = true;
            }
            over = over && n > 0 && valid;
            if (over) W.pixel_list[atomicAdd(&W.counts[CNT_PIXLIST], 1u)] = ql;
        }
        float c0 = 0, c1 = 0, c2 = 0;
        if (staged || n > 0) {
            const float inv = 1 / (float)(n > 0 ? n : 1);
            for_hit_samples([&](float r, float g, float b) { c0 += r * inv; c1 += g * inv; c2 += b * inv; });
        }
        if (!valid || over) continue;
        float g[3];
        uint8_t cnt_byte = 0;
        float zval = BIGFLOAT;
        if (n > 0) {
            g[0] = powf(c0, A.inv_gamma); g[1] = powf(c1, A.inv_gamma); g[2] = powf(c2, A.inv_gamma);
            cnt_byte = (n <= A.min_sample) ? 0 : 255;
            zval = hitz;
        } else {
            // background.Sample(Point3(x/W, y/H, 0)), FIN/main.cpp:326-328
            const V3 bgc = textured_color(A.S, ld3(A.bg), A.S.bg_map, mk((float)x / A.cam.width, (float)y / A.cam.height, 0));
            g[0] = powf(bgc.x, A.inv_gamma); g[1] = powf(bgc.y, A.inv_gamma); g[2] = powf(bgc.z, A.inv_gamma);
        }
        const uint32_t r8 = float_to_byte(g[0]), g8 = float_to_byte(g[1]), b8 = float_to_byte(g[2]);
        if (A.packed) {
            const uint32_t zb = __float_as_uint(zval);
            A.packed[(size_t)A.q0 + ql] = make_uint2(r8 | (g8 << 8) | (b8 << 16) | (zb << 24), (zb >> 8) | ((uint32_t)cnt_byte << 24));
        } else {
            A.count[index] = cnt_byte;
            A.z[index] = zval;
            A.rgb8[3 * index] = (uint8_t)r8; A.rgb8[3 * index + 1] = (uint8_t)g8; A.rgb8[3 * index + 2] = (uint8_t)b8;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-callable launch wrappers (C++ linkage, used by rt_api.cpp)
// ------------------------------------------------------------------------------------------------
static inline int grid_for(unsigned long long work, int block, int max_blocks)
{
    unsigned long long b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (unsigned long long)max_blocks) b = max_blocks;
    return (int)b;
}

static SlotMap make_slotmap(const DevCamera &cam, const DevTiles &tiles_prepared, uint32_t q0, int max_sample, int mode)
{
    SlotMap m; m.tiles = tiles_prepared; m.width = cam.width; m.height = cam.height; m.q0 = q0; m.max_sample = max_sample; m.mode = mode;
    m.div_ms = fastdiv_make((uint32_t)(max_sample > 0 ? max_sample : 1));
    return m;
}
// k_wavefront serves the FIN, P13 and P12 models when every mesh's BVH fits its traversal stack (RT_BVH_LDS entries per lane in LDS
// + RT_BVH_SPILL per thread in HBM; the host refuses meshes beyond that);
// RT_TRACER=levels forces the per-level structure (A/B)
// test hook: RT_WF_LDS_RAYS=n makes k_wavefront's workgroup rounds use only n entries of their LDS ray stack, so that a small frame
// already sends rays through the global queue, the second pass and the per-level launches
static uint32_t wf_lds_rays()
{
    const char *e = getenv("RT_WF_LDS_RAYS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (uint32_t)v : 0u;
}
static bool wavefront_usable(const DevScene &S, const rt_params &P, bool tex)
{
    static int wavefront = -1;
    if (wavefront < 0) { const char *e = getenv("RT_TRACER"); wavefront = (e && !strcmp(e, "levels")) ? 0 : 1; }
    // P12 (live GI): a hit spawns its hemisphere rays next to the reflection / refraction pair -- one per hit below the first
    // level, hemisphere_sample on it; with up to two of them the LDS stacks hold the tree like FIN's (what does not fit
    // takes the global queue as always).  RT_P12_TRACER=levels: the per-level kernels (A/B)
    static int p12 = -1;
    if (p12 < 0) { const char *e = getenv("RT_P12_TRACER"); p12 = (e && !strcmp(e, "levels")) ? 0 : 1; }
    const bool model_ok = P.shade_model == RT_SHADE_FIN || P.shade_model == RT_SHADE_P13 || (P.shade_model == RT_SHADE_P12 && p12 && P.hemisphere_sample <= 2);
    if (!wavefront || !model_ok) return false;
    // (the traversal stack: the kernel's LDS part plus, when the scene has the spill buffer, RT_BVH_SPILL entries per thread behind it)
    return S.max_bvh_depth <= (tex ? WfCfg<true>::BVH : WfCfg<false>::BVH) + (S.max_bvh_depth > RT_BVH_LDS ? RT_BVH_SPILL : 0);
}

bool rtk_wavefront_usable(const DevScene &S, const rt_params &P)
{
    return wavefront_usable(S, P, S.material_maps != nullptr || S.env_map.texture != RT_MAP_NONE);
}

void rtk_launch_primary(hipStream_t st, const DevScene &S, const DevWork &W, const rt_params &P,
                        const DevRayQueue &qout, uint32_t *qout_count, const DevCamera &cam,
                        const DevTiles &tiles, uint32_t q0, uint32_t npix, int j0, int ns,
                        int max_sample, int mode, const float *rays, int max_blocks)
{
    ShadeCtx C; C.S = S; C.S.bvh_spill = W.bvh_spill; C.W = W; C.P = P; C.qout = qout; C.qout_count = qout_count;
    PrimaryArgs A; A.cam = cam; A.tiles = tiles; A.q0 = q0; A.npix = npix; A.j0 = j0; A.ns = ns;
    A.max_sample = max_sample; A.mode = mode; A.rays = rays; A.lds_rays = wf_lds_rays();
    tiles_prepare(A.tiles);
    A.div_ns = fastdiv_make((uint32_t)(ns > 0 ? ns : 1));
    const int grid = grid_for((unsigned long long)npix * ns, RT_BLOCK, max_blocks);
    const bool tex = S.material_maps != nullptr || S.env_map.texture != RT_MAP_NONE;
    C.lds_a = C.lds_b = C.lds_c = nullptr; C.lds_count = nullptr; C.lds_cap = 0;
    C.sm = make_slotmap(cam, A.tiles, q0, max_sample, mode);
    // default: the whole ray tree in one persistent launch with LDS ray stacks (FIN / P13 models); RT_TRACER=levels: one
    // launch per level of the tree (round 1's structure, kept for the other models and for A/B: DESIGN.md section 3)
    if (wavefront_usable(S, P, tex)) {
        const int wgrid = grid_for((unsigned long long)npix * ns, RT_BLOCK, 256 * (tex ? RT_WF_TEX_WAVES : RT_WF_WAVES));      // resident workgroups per CU (LDS)
        if (P.shade_model == RT_SHADE_FIN) {
            if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_FIN, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
            else hipLaunchKernelGGL((k_wavefront<RT_SHADE_FIN, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        } else if (P.shade_model == RT_SHADE_P12) {
            if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_P12, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
            else hipLaunchKernelGGL((k_wavefront<RT_SHADE_P12, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        } else {
            if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_P13, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
            else hipLaunchKernelGGL((k_wavefront<RT_SHADE_P13, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        }
        return;
    }
#define RT_LAUNCH_PRIMARY(M) do { if (tex) hipLaunchKernelGGL((k_primary<M, true>), dim3(grid), dim3(RT_BLOCK), 0, st, C, A); \
                                  else hipLaunchKernelGGL((k_primary<M, false>), dim3(grid), dim3(RT_BLOCK), 0, st, C, A); } while (0)
    switch (P.shade_model) {
    case RT_SHADE_P13: RT_LAUNCH_PRIMARY(RT_SHADE_P13); break;
    case RT_SHADE_P12: RT_LAUNCH_PRIMARY(RT_SHADE_P12); break;
    case RT_SHADE_P6: RT_LAUNCH_PRIMARY(RT_SHADE_P6); break;
    case RT_SHADE_P3: RT_LAUNCH_PRIMARY(RT_SHADE_P3); break;
    default: RT_LAUNCH_PRIMARY(RT_SHADE_FIN);
    }
#undef RT_LAUNCH_PRIMARY
}

// The rays a k_wavefront pass could not keep on its LDS stacks, traced by a second pass of the same kernel with the
// queue as its source (their whole subtrees stay in LDS; what does not fit THIS time goes on to qout and the per-level
// launches).  Returns false when the model has no wavefront kernel (the caller then starts the level launches at qin).
bool rtk_launch_wavefront_queue(hipStream_t st, const DevScene &S, const DevWork &W, const rt_params &P,
                                const DevRayQueue &qin, const uint32_t *qin_count, const DevRayQueue &qout, uint32_t *qout_count,
                                const DevCamera &cam, const DevTiles &tiles, uint32_t q0, int max_sample, int mode)
{
    static int queue_pass = -1;
    if (queue_pass < 0) { const char *q = getenv("RT_WF_QUEUE_PASS"); queue_pass = (q && q[0] == '0') ? 0 : 1; }
    const bool tex = S.material_maps != nullptr || S.env_map.texture != RT_MAP_NONE;
    if (!queue_pass || !wavefront_usable(S, P, tex)) return false;
    ShadeCtx C; C.S = S; C.S.bvh_spill = W.bvh_spill; C.W = W; C.P = P; C.qout = qout; C.qout_count = qout_count;
    C.lds_a = C.lds_b = C.lds_c = nullptr; C.lds_count = nullptr; C.lds_cap = 0;
    DevTiles tp = tiles;
    tiles_prepare(tp);
    C.sm = make_slotmap(cam, tp, q0, max_sample, mode);
    PrimaryArgs A;
    memset(&A, 0, sizeof A);
    A.mode = 3; A.ns = 1; A.qsrc = qin; A.qsrc_count = qin_count; A.lds_rays = wf_lds_rays();
    A.div_ns = fastdiv_make(1u);
    const int wgrid = 256 * (tex ? RT_WF_TEX_WAVES : RT_WF_WAVES);       // the count is on the device: a full persistent grid, idle workgroups leave at once
    if (P.shade_model == RT_SHADE_FIN) {
        if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_FIN, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        else hipLaunchKernelGGL((k_wavefront<RT_SHADE_FIN, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
    } else if (P.shade_model == RT_SHADE_P12) {
        if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_P12, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        else hipLaunchKernelGGL((k_wavefront<RT_SHADE_P12, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
    } else {
        if (tex) hipLaunchKernelGGL((k_wavefront<RT_SHADE_P13, true>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
        else hipLaunchKernelGGL((k_wavefront<RT_SHADE_P13, false>), dim3(wgrid), dim3(RT_BLOCK), 0, st, C, A);
    }
    return true;
}

void rtk_launch_bounce(hipStream_t st, const DevScene &S, const DevWork &W, const rt_params &P,
                       const DevRayQueue &qin, const DevRayQueue &qout, uint32_t *qout_count,
                       int level, int max_blocks)
{
    ShadeCtx C; C.S = S; C.S.bvh_spill = W.bvh_spill; C.W = W; C.P = P; C.qout = qout; C.qout_count = qout_count;
    C.lds_a = C.lds_b = C.lds_c = nullptr; C.lds_count = nullptr; C.lds_cap = 0;
    memset(&C.sm, 0, sizeof C.sm);
    const bool tex = S.material_maps != nullptr || S.env_map.texture != RT_MAP_NONE;
#define RT_LAUNCH_BOUNCE(M) do { if (tex) hipLaunchKernelGGL((k_bounce<M, true>), dim3(max_blocks), dim3(RT_BLOCK), 0, st, C, qin, level); \
                                 else hipLaunchKernelGGL((k_bounce<M, false>), dim3(max_blocks), dim3(RT_BLOCK), 0, st, C, qin, level); } while (0)
    switch (P.shade_model) {
    case RT_SHADE_P13: RT_LAUNCH_BOUNCE(RT_SHADE_P13); break;
    case RT_SHADE_P12: RT_LAUNCH_BOUNCE(RT_SHADE_P12); break;
    case RT_SHADE_P6: RT_LAUNCH_BOUNCE(RT_SHADE_P6); break;
    case RT_SHADE_P3: break;                               // P3 has no secondary rays
    default: RT_LAUNCH_BOUNCE(RT_SHADE_FIN);
    }
#undef RT_LAUNCH_BOUNCE
}

void rtk_launch_trace(hipStream_t st, const DevScene &S, int model, const float *rays, long long n,
                      uint8_t *hit, float *z, float *p, float *N, int32_t *node, uint8_t *front)
{
    const int grid = grid_for((unsigned long long)n, RT_BLOCK, RT_SPILL_BLOCKS);
    if (model == RT_SHADE_P3) hipLaunchKernelGGL(k_trace<RT_SHADE_P3>, dim3(grid), dim3(RT_BLOCK), 0, st, S, rays, n, hit, z, p, N, node, front);
    else if (model != RT_SHADE_FIN) hipLaunchKernelGGL(k_trace<RT_SHADE_P13>, dim3(grid), dim3(RT_BLOCK), 0, st, S, rays, n, hit, z, p, N, node, front);
    else hipLaunchKernelGGL(k_trace<RT_SHADE_FIN>, dim3(grid), dim3(RT_BLOCK), 0, st, S, rays, n, hit, z, p, N, node, front);
}

void rtk_launch_gather(hipStream_t st, const DevPhotonMap &pm, const float4 *qa, const float4 *qb,
                       const float4 *qc, const uint32_t *count_ptr, uint32_t count_cap, int k,
                       float radius, float *sample_rgb, float *out_irr, float *out_dir, int mode,
                       unsigned long long *stats, int blocks, uint32_t *next_batch, float *cell_rk2)
{
    GatherArgs G; G.pm = pm; G.cell_rk2 = cell_rk2; G.qa = qa; G.qb = qb; G.qc = qc; G.count_ptr = count_ptr; G.count_cap = count_cap;
    G.k = k; G.radius = radius; G.sample_rgb = sample_rgb; G.out_irr = out_irr; G.out_dir = out_dir; G.mode = mode; G.stats = stats; G.next_batch = next_batch;
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64 * RT_GATHER_WAVES), 0, st, G);
}

// Un-interleave an all-gathered frame: rank r contributed its tiles r, r+R, r+2R, ... as `per_rank` packed tiles of
// tile_w*tile_h 8-byte pixel records (see ResolveArgs::packed); one thread per image pixel reads its record (8-byte
// loads, contiguous along a tile row) and writes the three planes of the RenderImage.
__global__ __launch_bounds__(256) void k_unpack_tiles(const uint2 *gathered, int world, int per_rank, int width, int height,
                                                      int tile_w, int tile_h, int tiles_x, uint8_t *rgb8, float *z, uint8_t *count)
{
    const size_t n = (size_t)width * height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
        const int tx = x / tile_w, ty = y / tile_h;
        const int t = ty * tiles_x + tx;
        const int r = t % world, k = t / world;
        const uint2 v = gathered[((size_t)r * per_rank + k) * (size_t)(tile_w * tile_h) + (size_t)(y - ty * tile_h) * tile_w + (x - tx * tile_w)];
        rgb8[3 * i] = (uint8_t)(v.x & 255u); rgb8[3 * i + 1] = (uint8_t)((v.x >> 8) & 255u); rgb8[3 * i + 2] = (uint8_t)((v.x >> 16) & 255u);
        z[i] = __uint_as_float((v.x >> 24) | (v.y << 8));
        count[i] = (uint8_t)(v.y >> 24);
    }
}

void rtk_launch_unpack_tiles(hipStream_t st, const void *gathered, int world, int per_rank, int width, int height, int tile_w, int tile_h,
                             uint8_t *rgb8, float *z, uint8_t *count)
{
    const int tiles_x = (width + tile_w - 1) / tile_w;
    const size_t n = (size_t)width * height;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_unpack_tiles, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, (const uint2 *)gathered, world, per_rank, width, height,
                       tile_w, tile_h, tiles_x, rgb8, z, count);
}

void rtk_launch_resolve(hipStream_t st, const DevScene &S, const DevWork &W, const DevCamera &cam, const DevTiles &tiles,
                        uint32_t q0, uint32_t npix, int min_sample, int max_sample, float threshold,
                        float inv_gamma, int phase, const float bg[3], uint8_t *rgb8, float *z,
                        uint8_t *count, void *packed, int max_blocks)
{
    ResolveArgs A; A.cam = cam; A.tiles = tiles; A.q0 = q0; A.npix = npix; A.min_sample = min_sample;
    A.max_sample = max_sample; A.threshold = threshold; A.inv_gamma = inv_gamma; A.phase = phase;
    A.bg[0] = bg[0]; A.bg[1] = bg[1]; A.bg[2] = bg[2]; A.rgb8 = rgb8; A.z = z; A.count = count; A.packed = (uint2 *)packed; A.direct_mode = 0; A.S = S;
    tiles_prepare(A.tiles);
    const int grid = grid_for(npix, 256, max_blocks);
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, st, W, A);
}
