/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the pinning status).
 *
 * Plain-C restatement of the reference's render path.  Every function follows the reference
 * operation by operation (same float expression order, same thresholds, same quirks) and
 * names the file:line it restates.  FIN = /root/reference/RayTracingFinal/RayTracingFinal,
 * P13 = /root/reference/RayTracingProj13/RayTracingProj13.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, IEEE float).
 *
 * Overload notes (C++ -> C): the reference calls pow()/exp() on float arguments with
 * <math.h>/<cmath> of a C++ standard library in scope, so overload resolution picks the
 * float versions (std::pow(float,float), exp(float)); pow(float,int) promotes to double.
 * Those choices are restated explicitly as powf/expf/pow below.
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BIGFLOAT 1.0e30f   /* FIN/include/scene.h:56 */
/* the reference's min/max are macros (FIN/include/scene.h:48-54): NaN-propagating as written */
#define RMIN(a,b) ((a)<(b)?(a):(b))
#define RMAX(a,b) ((a)>(b)?(a):(b))

static orc_counters g_cnt;
/* FIN's discarded hemisphere loop (FIN/main.cpp:642-693): traced only on request, see shade_fin */
static int g_trace_discarded = 0;
static uint64_t g_discarded_rays = 0;
void orc_set_trace_discarded(int on) { g_trace_discarded = on; g_discarded_rays = 0; }
uint64_t orc_discarded_rays(void) { return g_discarded_rays; }
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void orc_counters_get(orc_counters *out) { *out = g_cnt; }

/* ---- counter RNG for the stochastic effects (replaces libc rand(); same generator and same
 * (sample, node, purpose, index) addressing as the HIP kernels, see rt_kernels.hip) ------------ */
static void philox4x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
#define RNG_LENS 1u
#define RNG_PICK 2u
#define RNG_SHADOW 3u
#define RNG_GLOSSR 4u
#define RNG_GLOSST 5u
#define RNG_GI 6u
static struct { uint32_t seed, sample, node; } g_rng = { 0, 0, 1 };
void orc_set_rng(uint32_t seed, uint32_t sample, uint32_t node) { g_rng.seed = seed; g_rng.sample = sample; g_rng.node = node; }
static void rng2_at(uint32_t seed, uint32_t sample, uint32_t node, uint32_t purpose, uint32_t index, float *u0, float *u1)
{
    uint32_t o[4];
    philox4x32(seed, 0x52544D49u, sample, node, (purpose << 24) | (index & 0xFFFFFFu), 0x5eed5eedu, o);
    *u0 = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
    *u1 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
}

/* ---- test hooks: a scripted rand() stream and a scripted Shadow --------------------------------
 * The reference draws from libc rand() and calls GenLight::Shadow; to compare this restatement with
 * the reference's own PointLight::Illuminate / MtlBlinn::RandomPhotonBounce (compiled in
 * oracle/ref_harness.cpp) draw for draw, a test hands over the RAW rand() values the reference
 * consumed and the values its Shadow double returned; while a script is active every random draw
 * takes the next raw value exactly the way the reference's expression does
 * (`rand() / (float) RAND_MAX`, or the double forms of P13/include/lights.h:73-74) and orc_shadow
 * logs its ray and returns the next scripted value instead of tracing (when shadow values or a log are handed over:
 * a script of rand() values alone -- a whole Shade of RayTracingProj12 under its captured stream -- traces its shadow rays). */
static struct { const int32_t *raw; int n, i; const float *sh; int nsh, ish; float *log; int log_cap, nlog; int on; } g_script;
void orc_script_begin(const int32_t *raw_rand, int n_rand, const float *shadow_values, int n_shadow, float *shadow_log, int log_cap)
{
    g_script.raw = raw_rand; g_script.n = n_rand; g_script.i = 0;
    g_script.sh = shadow_values; g_script.nsh = n_shadow; g_script.ish = 0;
    g_script.log = shadow_log; g_script.log_cap = log_cap; g_script.nlog = 0; g_script.on = 1;
}
int orc_script_end(int *rand_used)
{
    if (rand_used) *rand_used = g_script.i;
    g_script.on = 0;
    return g_script.nlog;
}
#define REF_RAND_MAX 2147483647            /* glibc RAND_MAX, the value the harness ran with */
static int32_t script_raw(void) { return (g_script.i < g_script.n) ? g_script.raw[g_script.i++] : (g_script.i++, 0); }
static float script_uniform(void) { return script_raw() / (float)REF_RAND_MAX; }    /* rand() / (float) RAND_MAX */

static void rng2(uint32_t purpose, uint32_t index, float *u0, float *u1)
{
    if (g_script.on) { *u0 = script_uniform(); *u1 = script_uniform(); return; }
    rng2_at(g_rng.seed, g_rng.sample, g_rng.node, purpose, index, u0, u1);
}
/* the same two draws as numerator/denominator pairs for expressions of the form
 * `M_PI * 2.0 * rand() / (float) RAND_MAX` (evaluated in double, left to right): with the counter
 * generator the numerator is the uniform itself and the denominator 1.0 (an exact division) */
static void rng2_ratio(uint32_t purpose, uint32_t index, double *r0, double *r1, double *den)
{
    if (g_script.on) { *r0 = (double)script_raw(); *r1 = (double)script_raw(); *den = (double)(float)REF_RAND_MAX; return; }
    float u0, u1;
    rng2_at(g_rng.seed, g_rng.sample, g_rng.node, purpose, index, &u0, &u1);
    *r0 = (double)u0; *r1 = (double)u1; *den = 1.0;
}
static uint32_t child_node(uint32_t node, uint32_t kind) { return node * 0x9E3779B1u + kind * 0x85EBCA6Bu + 0x27D4EB2Fu; }

/* ---- cyPoint3f subset (FIN/include/cyPoint.h:259-350) ---------------------------------- */
typedef struct { float x, y, z; } v3;
static v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static v3 v3p(const float *p) { return V3(p[0], p[1], p[2]); }
static v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* Dot = (a*b).Sum() = x+y+z left to right, cyPoint.h:342,303 */
static float vdot(v3 a, v3 b) { v3 r = vmul(a, b); return r.x + r.y + r.z; }
static v3 vcross(v3 a, v3 p) { return V3(a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x); }
static float vlen2(v3 a) { return vdot(a, a); }
static float vlen(v3 a) { return sqrtf(vlen2(a)); }
static v3 vnorm(v3 a) { return vdivs(a, vlen(a)); }            /* *this /= Length() */
static void st3(float *o, v3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }

/* Matrix3 * Point3, column-major (FIN/include/cyMatrix.h:542-546) */
static v3 mmul(const float *d, v3 p)
{
    return V3(p.x * d[0] + p.y * d[3] + p.z * d[6],
              p.x * d[1] + p.y * d[4] + p.z * d[7],
              p.x * d[2] + p.y * d[5] + p.z * d[8]);
}
/* Transformation::TransposeMult (FIN/include/scene.h:254-261): column i dot dir */
static v3 mtmul(const float *d, v3 dir)
{
    return V3(vdot(V3(d[0], d[1], d[2]), dir),
              vdot(V3(d[3], d[4], d[5]), dir),
              vdot(V3(d[6], d[7], d[8]), dir));
}

/* ---- Halton, Color24 -------------------------------------------------------------------- */
/* FIN/include/scene.h:131-140 */
float orc_halton(int index, int base)
{
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (i % base);
        f /= (float)base;
    }
    return r;
}

/* Color24(Color): FloatToByte = Clamp(int(r*255)) (FIN/include/cyColor.h:245-246).
 * int(NaN) is undefined behaviour in the reference; this restatement maps NaN to 0 and
 * saturates out-of-int-range values (documented deviation). */
static uint8_t float_to_byte(float r)
{
    float s = r * 255;
    if (!(s == s)) return 0;
    if (s <= -2147483648.0f) return 0;
    if (s >= 2147483647.0f) return 255;
    int v = (int)s;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
void orc_color24(const float rgb[3], uint8_t out[3])
{
    out[0] = float_to_byte(rgb[0]); out[1] = float_to_byte(rgb[1]); out[2] = float_to_byte(rgb[2]);
}

/* ---- primitives -------------------------------------------------------------------------- */
/* sphere texture coordinate, FIN/include/objects.h:49-51: atan2/asin on floats promote to double */
static void sphere_uvw(v3 p, float uvw[3])
{
    uvw[0] = (float)(0.5 - atan2(p.x, p.y) / (2 * M_PI));
    uvw[1] = (float)(0.5 + asin(p.z) / M_PI);
    uvw[2] = 0;
}

/* Sphere::IntersectRay, FIN/include/objects.h:24-70 (identical in P13/include/objects.h:23-69). */
/* Sphere::IntersectRay of RayTracingProj3 (main.cpp:192-221): no bias, z = min(t1,t2), rejected when
 * negative or not closer; N = p un-normalised; front untouched */
static int sphere_intersect_p3(const float ray[6], orc_hit *hit)
{
    v3 rp = v3p(ray), rd = v3p(ray + 3);
    float a = vdot(rd, rd);
    float c = vdot(rp, rp) - 1;
    float b = 2 * vdot(rp, rd);
    float insqrt = b * b - (4 * a * c);
    if (insqrt >= 0) {
        float t1 = (-b + sqrtf(insqrt)) / (a * 2);
        float t2 = (-b - sqrtf(insqrt)) / (a * 2);
        float prez = hit->z;
        float zz = RMIN(t1, t2);
        if (zz < 0) return 0;
        if (zz >= prez) return 0;
        hit->z = zz;
        v3 p = vadd(vscale(rd, hit->z), rp);
        st3(hit->p, p);
        st3(hit->N, p);
        return 1;
    }
    return 0;
}

int orc_sphere_intersect(int model, const float ray[6], orc_hit *hit)
{
    if (model == RT_SHADE_P3) return sphere_intersect_p3(ray, hit);
    v3 rp = v3p(ray), rd = v3p(ray + 3);
    int behitted = 0;
    float a = vdot(rd, rd);
    float c = vdot(rp, rp) - 1;
    float b = 2 * vdot(rp, rd);
    float insqrt = b * b - (4 * a * c);
    float zero = 0.001f;
    if (insqrt >= zero) {
        float t1 = (-b + sqrtf(insqrt)) / (a * 2);
        float t2 = (-b - sqrtf(insqrt)) / (a * 2);
        float prez = hit->z;
        float min_t = t2;
        if (min_t >= prez) return 0;
        if (t1 > zero && t2 < zero && t1 < prez) {
            hit->z = t1;
            hit->front = 0;
            behitted = 1;
            v3 p = vadd(vscale(rd, hit->z), rp);          /* hitinfo.z*ray.dir+ray.p */
            st3(hit->p, p);
            st3(hit->N, vnorm(p));
            sphere_uvw(p, hit->uvw);
        } else if (t1 > zero && t2 > zero && t2 < prez) {
            behitted = 1;
            hit->z = min_t;
            hit->front = 1;
            v3 p = vadd(vscale(rd, hit->z), rp);
            st3(hit->p, p);
            st3(hit->N, vnorm(p));
            sphere_uvw(p, hit->uvw);
        }
    }
    return behitted;
}

/* Plane::IntersectRay, FIN/include/objects.h:84-111; the P13 variant flips `front`
 * (P13/include/objects.h:83-110: N.d < 0 -> front=false). */
int orc_plane_intersect(int model, const float ray[6], orc_hit *hit)
{
    float zero = 0.001f;
    v3 P = v3p(ray), d = v3p(ray + 3);
    v3 N = V3(0, 0, 1);
    float t = -(P.z / d.z);
    if (t >= zero && t < BIGFLOAT && t < hit->z) {
        v3 Hitp = vadd(P, vscale(d, t));                   /* P+t*d */
        if (Hitp.x >= -1 && Hitp.x <= 1 && Hitp.y >= -1 && Hitp.y <= 1) {
            hit->z = t;
            st3(hit->p, Hitp);
            st3(hit->N, N);
            float nd = vdot(N, d);
            hit->uvw[0] = (Hitp.x + 1) / 2; hit->uvw[1] = (Hitp.y + 1) / 2; hit->uvw[2] = 0;      /* :103 */
            if (model != RT_SHADE_FIN) hit->front = (nd < 0.0f) ? 0 : 1;
            else                       hit->front = (nd <= 0.0f) ? 1 : 0;
            return 1;
        }
    }
    return 0;
}

/* Box::IntersectRay, FIN/scene.cpp:11-65 */
int orc_box_intersect(const float box[6], const float ray[6], float t_max)
{
    g_cnt.box_tests++;
    const float *p = ray, *dir = ray + 3;
    /* IsInside, FIN/include/scene.h:123 */
    int inside = 1;
    for (int i = 0; i < 3; i++) if (box[i] > p[i] || box[3 + i] < p[i]) { inside = 0; break; }
    if (inside) return 1;
    float tenter = -t_max;
    float texit = t_max;
    for (int i = 0; i < 3; i++) {
        if (dir[i] != 0.0f) {
            float t0 = (box[i] - p[i]) / dir[i];
            float t1 = (box[3 + i] - p[i]) / dir[i];
            if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
            tenter = RMAX(t0, tenter);
            texit = RMIN(t1, texit);
        }
    }
    return tenter <= texit && texit <= t_max;
}

static v3 mesh_v(const orc_mesh *m, uint32_t i) { return v3p(m->v + 3 * (size_t)i); }
static v3 mesh_vn(const orc_mesh *m, uint32_t i) { return v3p(m->vn + 3 * (size_t)i); }

/* TriObj::TriangleArea, FIN/include/objects.h:146-157 */
static float triangle_area(int i, v3 A, v3 B, v3 C)
{
    switch (i) {
    case 0: return (B.y - A.y) * (C.z - A.z) - (C.y - A.y) * (B.z - A.z);
    case 1: return (B.x - A.x) * (C.z - A.z) - (C.x - A.x) * (B.z - A.z);
    default: return (B.x - A.x) * (C.y - A.y) - (C.x - A.x) * (B.y - A.y);
    }
}

/* cyTriMesh::GetNormal = Interpolate(faceID, vn, fn, bc), FIN/include/cyTriMesh.h:167,191 */
static v3 mesh_normal(const orc_mesh *m, uint32_t face, v3 bc)
{
    const uint32_t *fn = m->fn + 3 * (size_t)face;
    return vadd(vadd(vscale(mesh_vn(m, fn[0]), bc.x), vscale(mesh_vn(m, fn[1]), bc.y)),
                vscale(mesh_vn(m, fn[2]), bc.z));
}

/* TriObj::IntersectTriangle, FIN/include/objects.h:226-267 (two-sided, bias 1e-3) */
static int tri_intersect_fin(const orc_mesh *m, const float ray[6], orc_hit *hit, uint32_t face)
{
    g_cnt.tri_tests++;
    v3 rp = v3p(ray), rd = v3p(ray + 3);
    float bias = 0.001f;
    const uint32_t *f = m->f + 3 * (size_t)face;
    v3 A = mesh_v(m, f[0]), B = mesh_v(m, f[1]), C = mesh_v(m, f[2]);
    v3 N = vnorm(vcross(vsub(B, A), vsub(C, A)));
    const float dz = vdot(rd, N);
    if (fabsf(dz) < 1e-7f) return 0;
    const float pz = vdot(vsub(rp, A), N);
    const float t = -pz / dz;
    if (t <= bias) return 0;
    if (t < hit->z) {
        int front = (dz <= 0);
        /* CheckHit(hitSide=HIT_FRONT, front) is always true (objects.h:120-122) */
        v3 p = vadd(rp, vscale(rd, t));
        int ignoredAxis;
        const float abs_nx = fabsf(N.x), abs_ny = fabsf(N.y), abs_nz = fabsf(N.z);
        if (abs_nx > abs_ny && abs_nx > abs_nz) ignoredAxis = 0;
        else if (abs_ny > abs_nz) ignoredAxis = 1;
        else ignoredAxis = 2;
        const float s = 1.f / triangle_area(ignoredAxis, A, B, C);
        const float a = triangle_area(ignoredAxis, p, B, C) * s;
        const float b = triangle_area(ignoredAxis, p, C, A) * s;
        const float c = 1.f - a - b;
        if (a < 0 || b < 0 || c < 0) return 0;
        hit->z = t;
        st3(hit->p, p);
        st3(hit->N, mesh_normal(m, face, V3(a, b, c)));    /* NOT normalised */
        hit->front = front;
        return 1;
    }
    return 0;
}

/* TriObj::IntersectTriangle, P13/include/objects.h:148-206 (back-face culled, bias 1e-7) */
static int tri_intersect_p13(const orc_mesh *m, const float ray[6], orc_hit *hit, uint32_t face)
{
    g_cnt.tri_tests++;
    v3 rp = v3p(ray), rd = v3p(ray + 3);
    float bias = 1e-7f;
    const uint32_t *f = m->f + 3 * (size_t)face;
    v3 A = mesh_v(m, f[0]), B = mesh_v(m, f[1]), C = mesh_v(m, f[2]);
    v3 tN = vnorm(vcross(vsub(B, A), vsub(C, A)));
    if (vdot(tN, vsub(rp, A)) < bias) return 0;
    if (vdot(tN, rd) == 0) return 0;
    float t = vdot(tN, vsub(C, rp)) / vdot(tN, rd);
    if (t < bias || t >= hit->z || t >= BIGFLOAT) return 0;
    /* fmax/fabs are the double versions */
    double maxN = fmax(fabs(tN.x), fabs(tN.y));
    float maxNf = (float)maxN;                             /* float maxN = fmax(...) */
    maxNf = (float)fmax(fabs(tN.z), maxNf);
    v3 P = vadd(rp, vscale(rd, t));
    float pa[2], pb[2], pc[2], pp[2];
    if (maxNf == fabs(tN.x)) {
        pa[0] = A.y; pa[1] = A.z; pb[0] = B.y; pb[1] = B.z; pc[0] = C.y; pc[1] = C.z; pp[0] = P.y; pp[1] = P.z;
    } else if (maxNf == fabs(tN.y)) {
        pa[0] = A.x; pa[1] = A.z; pb[0] = B.x; pb[1] = B.z; pc[0] = C.x; pc[1] = C.z; pp[0] = P.x; pp[1] = P.z;
    } else {
        pa[0] = A.x; pa[1] = A.y; pb[0] = B.x; pb[1] = B.y; pc[0] = C.x; pc[1] = C.y; pp[0] = P.x; pp[1] = P.y;
    }
    /* Point2::Cross: x*p.y - y*p.x (cyPoint.h) */
#define CR2(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
    float area_tri = CR2(pa[0] - pc[0], pa[1] - pc[1], pb[0] - pc[0], pb[1] - pc[1]);
    float area_bcp = CR2(pp[0] - pc[0], pp[1] - pc[1], pb[0] - pc[0], pb[1] - pc[1]);
    float area_acp = CR2(pa[0] - pc[0], pa[1] - pc[1], pp[0] - pc[0], pp[1] - pc[1]);
#undef CR2
    float alpha = area_bcp / area_tri;
    float beta = area_acp / area_tri;
    float gamma_ = (float)(1.0 - alpha - beta);
    if (alpha < -bias || beta < -bias || gamma_ < -bias || alpha > 1.0 || beta > 1.0 || gamma_ > 1.0) return 0;
    v3 PN = mesh_normal(m, face, V3(alpha, beta, gamma_));
    hit->front = 1;
    st3(hit->p, vadd(vadd(vscale(A, alpha), vscale(B, beta)), vscale(C, gamma_)));
    st3(hit->N, vnorm(PN));
    hit->z = t;
    /* hInfo.uvw = GetTexCoord(faceID, bc) = Interpolate(faceID, vt, ft, bc), P13/include/objects.h:203,
     * cyTriMesh.h:173,191.  Without texture vertices the reference reads through a null vt; here uvw
     * stays as it was. */
    if (m->vt && m->ft) {
        const uint32_t *ft = m->ft + 3 * (size_t)face;
        v3 uvw = vadd(vadd(vscale(v3p(m->vt + 3 * (size_t)ft[0]), alpha), vscale(v3p(m->vt + 3 * (size_t)ft[1]), beta)),
                      vscale(v3p(m->vt + 3 * (size_t)ft[2]), gamma_));
        st3(hit->uvw, uvw);
    }
    return 1;
}

/* TriObj::TraceBVHNode, FIN/include/objects.h:271-302: depth first, child1 then child2,
 * box test with t_max = BIGFLOAT (no closest-hit culling). */
static int trace_bvh_node(int model, const orc_mesh *m, const float ray[6], orc_hit *hit, uint32_t id)
{
    const rt_bvh_node *n = &m->nodes[id];
    g_cnt.node_visits++;
    int hitted = 0;
    if (!orc_box_intersect(n->box, ray, BIGFLOAT)) return 0;
    if (n->data & 0x80000000u) {                           /* IsLeafNode, cyBVH.h:195 */
        uint32_t count = ((n->data >> 28) & 7u) + 1;       /* ElementCount, cyBVH.h:194 */
        uint32_t off = n->data & 0x0FFFFFFFu;              /* ElementOffset, cyBVH.h:193 */
        for (uint32_t i = 0; i < count; i++) {
            int h = (model != RT_SHADE_FIN) ? tri_intersect_p13(m, ray, hit, m->elements[off + i])
                                            : tri_intersect_fin(m, ray, hit, m->elements[off + i]);
            if (h) hitted = 1;
        }
        return hitted;
    }
    uint32_t c = n->data & 0x7FFFFFFFu;                    /* ChildIndex, cyBVH.h:192 */
    if (trace_bvh_node(model, m, ray, hit, c)) hitted = 1;
    if (trace_bvh_node(model, m, ray, hit, c + 1)) hitted = 1;
    return hitted;
}

int orc_mesh_intersect(int model, const orc_mesh *m, const float ray[6], orc_hit *hit)
{
    if (m->nf <= 0 || m->nnodes < 2) return 0;
    return trace_bvh_node(model, m, ray, hit, 1);          /* GetRootNodeID() == 1 */
}

/* ---- scene graph ------------------------------------------------------------------------- */
/* Node::ToNodeCoords, FIN/include/scene.h:502-508 */
void orc_to_node_coords(const rt_node *n, const float ray[6], float out[6])
{
    v3 p = v3p(ray), d = v3p(ray + 3), pos = v3p(n->pos);
    v3 rp = mmul(n->itm, vsub(p, pos));                    /* TransformTo, scene.h:236 */
    v3 rd = vsub(mmul(n->itm, vsub(vadd(p, d), pos)), rp);
    st3(out, rp); st3(out + 3, rd);
}

/* Node::FromNodeCoords, FIN/include/scene.h:509-513 */
void orc_from_node_coords(const rt_node *n, orc_hit *hit)
{
    v3 p = vadd(mmul(n->tm, v3p(hit->p)), v3p(n->pos));    /* TransformFrom, scene.h:237 */
    v3 N = vnorm(mtmul(n->itm, v3p(hit->N)));              /* VectorTransformFrom + GetNormalized */
    st3(hit->p, p); st3(hit->N, N);
}

/* TraceNode, FIN/main.cpp:108-130 */
static int trace_node(const orc_scene *s, int model, int idx, const float ray[6], orc_hit *hit)
{
    const rt_node *node = &s->nodes[idx];
    float r[6];
    orc_to_node_coords(node, ray, r);
    int h = 0;
    if (node->obj_type != RT_OBJ_NONE) {
        int oh = 0;
        if (node->obj_type == RT_OBJ_SPHERE) oh = orc_sphere_intersect(model, r, hit);
        else if (node->obj_type == RT_OBJ_PLANE) oh = orc_plane_intersect(model, r, hit);
        else if (node->obj_type == RT_OBJ_MESH && node->mesh >= 0 && node->mesh < s->n_meshes)
            oh = orc_mesh_intersect(model, &s->meshes[node->mesh], r, hit);
        if (oh) {
            h = 1;
            hit->node = idx;
            orc_from_node_coords(node, hit);
        }
    }
    for (int c = idx + 1; c < s->n_nodes; c++) {           /* children in document order */
        if (s->nodes[c].parent != idx) continue;
        if (trace_node(s, model, c, r, hit)) {
            orc_from_node_coords(node, hit);
            h = 1;
        }
    }
    return h;
}

static void hit_init(orc_hit *h)
{
    /* HitInfo::Init, FIN/include/scene.h:163 */
    memset(h, 0, sizeof *h);
    h->z = BIGFLOAT; h->node = -1; h->front = 1;
    h->uvw[0] = h->uvw[1] = h->uvw[2] = 0.5f;
}

int orc_trace(const orc_scene *s, int model, const float ray[6], orc_hit *hit)
{
    hit_init(hit);
    if (s->n_nodes <= 0) return 0;
    return trace_node(s, model, 0, ray, hit);
}

/* GenLight::Shadow, FIN/main.cpp:499-513 */
float orc_shadow(const orc_scene *s, int model, const float ray[6], float t_max)
{
    float bias = 1e-14f;
    orc_hit h;
    if (g_script.on && (g_script.nsh > 0 || g_script.log)) {   /* test hook, see orc_script_begin; a script of rand() values alone leaves Shadow tracing */
        if (g_script.log && g_script.nlog < g_script.log_cap) { memcpy(g_script.log + 7 * g_script.nlog, ray, 24); g_script.log[7 * g_script.nlog + 6] = t_max; }
        g_script.nlog++;
        return g_script.nsh > 0 ? g_script.sh[(g_script.ish++) % g_script.nsh] : 1.0f;
    }
    g_cnt.rays_shadow++;
    if (orc_trace(s, model, ray, &h)) {
        if (h.z > bias && h.z < t_max) return 0.0f;
    }
    return 1.0f;
}

/* Light::Direction: Ambient (0,0,0) lights.h:35; Direct: direction lights.h:51;
 * Point: (p-position).GetNormalized() lights.h:159 */
static v3 light_direction(const rt_light *l, v3 p)
{
    if (l->type == RT_LIGHT_POINT) return vnorm(vsub(p, v3p(l->position)));
    if (l->type == RT_LIGHT_DIRECT) return v3p(l->direction);
    return V3(0, 0, 0);
}

void orc_light_direction(const rt_light *l, const float p[3], float out[3]) { st3(out, light_direction(l, v3p(p))); }

/* Light::Illuminate.
 *   Ambient: intensity (FIN/include/lights.h:34)
 *   Direct : Shadow(Ray(p,-direction)) * intensity, t_max = BIGFLOAT (lights.h:50)
 *   Point  : FIN/include/lights.h:67-131 -- 4 samples on a disc of radius `size`; with size==0
 *            every sample ray is position-p exactly (xv1.Length() == 0), the mean is 0 or 1 and
 *            the 16-sample refinement never runs.  size>0 draws rand() (statistical only).
 *            P13 variant (P13/include/lights.h:65-91) also collapses to position-p at size 0;
 *            RayTracingProj12's (include/lights.h:66-89) is P13's without the division by distance^2. */
void orc_illuminate(const orc_scene *s, const rt_params *P, const rt_light *l,
                    const float p_[3], const float N_[3], float out[3])
{
    (void)N_;
    int model = P->shade_model;
    v3 p = v3p(p_);
    v3 I = v3p(l->intensity);
    if (l->type == RT_LIGHT_AMBIENT) { st3(out, I); return; }
    if (l->type == RT_LIGHT_DIRECT) {
        float ray[6];
        st3(ray, p); st3(ray + 3, vneg(v3p(l->direction)));
        st3(out, vscale(I, orc_shadow(s, model, ray, BIGFLOAT)));   /* Shadow(...) * intensity */
        return;
    }
    v3 position = v3p(l->position);
    if (model == RT_SHADE_P6 || model == RT_SHADE_P3) {
        /* PointLight::Illuminate of P6/P3 (include/lights.h:61): Shadow(Ray(p,position-p),1) * intensity */
        float ray[6];
        st3(ray, p); st3(ray + 3, vsub(position, p));
        st3(out, vscale(I, orc_shadow(s, model, ray, 1)));
        return;
    }
    float size = l->size;
    int ns = P->shadow_samples > 0 ? P->shadow_samples : 4;
    const uint32_t li = (uint32_t)(l - s->lights);
    float shadow = 0.0f;
    float u0, u1;
    if (model != RT_SHADE_FIN) {
        float shadow_coef = 0.0f;
        for (int i = 0; i < ns; i++) {
            double r0, r1, den;
            rng2_ratio(RNG_SHADOW, li * 64 + (uint32_t)i, &r0, &r1, &den);
            float r = orc_halton(i, 2);
            r = sqrtf(r) * size;
            float theta = (float)(M_PI * 2.0 * r0 / den);              /* M_PI * 2.0 * rand()/ (float) RAND_MAX */
            float gam = (float)(M_PI * r1 / den);                      /* M_PI * rand()/ (float) RAND_MAX */
            float dx = r * sinf(gam) * cosf(theta);
            float dy = r * sinf(gam) * sinf(theta);
            float dz = r * cosf(gam);
            v3 newlightPos = vadd(V3(dx, dy, dz), position);
            float ray[6];
            st3(ray, p); st3(ray + 3, vsub(newlightPos, p));
            shadow_coef += orc_shadow(s, model, ray, 1);
        }
        v3 avg_shadow = vdivs(vscale(I, shadow_coef), (float)ns);   /* intensity*coef/SAMPLES */
        if (model == RT_SHADE_P12) { st3(out, avg_shadow); return; }   /* RayTracingProj12/include/lights.h:86-88: no fall-off yet */
        float distance = vlen2(vsub(p, position));
        st3(out, vdivs(avg_shadow, distance));
        return;
    }
    v3 dir = vsub(position, p);
    v3 xAxis = V3(1, 0, 0), yAxis = V3(0, 1, 0), v1;
    if (vdot(dir, xAxis) > 0.8) v1 = vcross(yAxis, dir);
    else v1 = vcross(xAxis, dir);
    v3 v2 = vcross(v1, dir);
    v2 = vnorm(v2);
    v1 = vnorm(v1);
    for (int i = 0; i < ns; i++) {
        rng2(RNG_SHADOW, li * 64 + (uint32_t)i, &u0, &u1);
        float rRadius = sqrtf(u0) * size;
        float rAngle = (float)((double)u1 * (2.0 * M_PI));
        float xv = rRadius * cosf(rAngle);                             /* cos(float) -> float overload */
        float yv = rRadius * sinf(rAngle);
        v3 xv1 = vscale(v1, xv), yv2 = vscale(v2, yv);
        /* (position + xv1.Length() + yv2.Length()) - p : scalars added to all 3 coords */
        float lx = vlen(xv1), ly = vlen(yv2);
        v3 sd = vsub(V3(position.x + lx + ly, position.y + lx + ly, position.z + lx + ly), p);
        float ray[6];
        st3(ray, p); st3(ray + 3, sd);
        shadow += orc_shadow(s, model, ray, 1);
    }
    shadow /= (float)ns;
    if (shadow != 0.0 && shadow != 1.0) {
        const int nmax = 16;                               /* MAX_SHADOW_SAMPLES */
        shadow = 0.0f;
        for (int i = 0; i < nmax; i++) {
            rng2(RNG_SHADOW, li * 64 + 32 + (uint32_t)i, &u0, &u1);
            float rRadius = sqrtf(u0) * size;
            float rAngle = (float)((double)u1 * (2.0 * M_PI));
            float xv = rRadius * cosf(rAngle);
            float yv = rRadius * sinf(rAngle);
            v3 xv1 = vscale(v1, -xv), yv2 = vscale(v2, -yv);
            float lx = vlen(xv1), ly = vlen(yv2);
            v3 sd = vsub(V3(position.x + lx + ly, position.y + lx + ly, position.z + lx + ly), p);
            float ray[6];
            st3(ray, p); st3(ray + 3, sd);
            shadow += orc_shadow(s, model, ray, 1);
        }
        shadow /= (float)nmax;
    }
    /* intensity * shadow / (p - position).LengthSquared() */
    st3(out, vdivs(vscale(I, shadow), vlen2(vsub(p, position))));
}

/* ---- textures ------------------------------------------------------------------------------- */
/* Texture::TileClamp, FIN/include/scene.h:356-366 */
static v3 tile_clamp(v3 uvw)
{
    v3 u;
    u.x = uvw.x - (int)uvw.x; u.y = uvw.y - (int)uvw.y; u.z = uvw.z - (int)uvw.z;
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
/* TextureFile::Sample (FIN/texture.cpp:95-121), TextureChecker::Sample (:125-133) */
void orc_texture_sample(const rt_texture *t, const uint8_t *texels, const float uvw_[3], float rgb[3])
{
    v3 u = tile_clamp(v3p(uvw_));
    if (t->type == RT_TEX_CHECKER) {
        const float *c = (u.x <= 0.5f) ? (u.y <= 0.5f ? t->color1 : t->color2) : (u.y <= 0.5f ? t->color2 : t->color1);
        rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
        return;
    }
    int width = t->width, height = t->height;
    if (width + height == 0) { rgb[0] = rgb[1] = rgb[2] = 0; return; }
    const uint8_t *data = texels + t->texel_offset;
    float x = width * u.x;
    float y = height * u.y;
    int ix = (int)x;
    int iy = (int)y;
    float fx = x - ix;
    float fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1;
    if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1;
    if (iyp >= height) iyp -= height;
#define TEXEL(X, Y) V3(data[3 * ((Y) * width + (X))] / 255.0f, data[3 * ((Y) * width + (X)) + 1] / 255.0f, data[3 * ((Y) * width + (X)) + 2] / 255.0f)
    v3 r = vadd(vadd(vadd(vscale(TEXEL(ix, iy), (1 - fx) * (1 - fy)), vscale(TEXEL(ixp, iy), fx * (1 - fy))),
                     vscale(TEXEL(ix, iyp), (1 - fx) * fy)), vscale(TEXEL(ixp, iyp), fx * fy));
#undef TEXEL
    st3(rgb, r);
}
/* TextureMap: Transformation::TransformTo(uvw) = itm * (uvw - pos), scene.h:236,383 */
void orc_texmap_transform(const rt_texmap *m, const float uvw[3], float out[3])
{
    st3(out, mmul(m->itm, vsub(v3p(uvw), v3p(m->pos))));
}
/* the coordinate TexturedColor::SampleEnvironment looks up, scene.h:426-432 */
void orc_environment_coord(const float dir[3], float uvw[3])
{
    float z = asinf(-dir[2]) / (float)M_PI + 0.5f;
    float x = dir[0] / (fabsf(dir[0]) + fabsf(dir[1]));
    float y = dir[1] / (fabsf(dir[0]) + fabsf(dir[1]));
    /* Point3(0.5,0.5,0) + z*(x*Point3(0.5,0.5,0) + y*Point3(-0.5,0.5,0)) */
    v3 a = vadd(vscale(V3(0.5f, 0.5f, 0), x), vscale(V3(-0.5f, 0.5f, 0), y));
    st3(uvw, vadd(V3(0.5f, 0.5f, 0.0f), vscale(a, z)));
}
/* TexturedColor::Sample(uvw[,duvw]) with duvw == 0 (scene.h:331-334,383-392,422-423) */
void orc_textured_color(const orc_scene *s, const float color[3], const rt_texmap *map, const float uvw[3], float out[3])
{
    if (!map || map->texture == RT_MAP_NONE) { out[0] = color[0]; out[1] = color[1]; out[2] = color[2]; return; }
    float t[3] = {0, 0, 0};
    if (map->texture >= 0 && map->texture < s->n_textures) {
        float u[3];
        orc_texmap_transform(map, uvw, u);
        orc_texture_sample(&s->textures[map->texture], s->texels, u, t);
    }
    out[0] = color[0] * t[0]; out[1] = color[1] * t[1]; out[2] = color[2] * t[2];
}
static v3 environment_color(const orc_scene *s, v3 dir)
{
    float uvw[3], c[3];
    orc_environment_coord((const float *)&dir, uvw);
    orc_textured_color(s, s->env, s->env_map, uvw, c);
    return v3p(c);
}
static void material_colors(const orc_scene *s, const orc_hit *h, const rt_blinn *m, v3 *kd, v3 *ks)
{
    int mi = s->nodes[h->node].material;
    float c[3];
    orc_textured_color(s, m->diffuse, s->material_maps ? &s->material_maps[2 * mi] : 0, h->uvw, c); *kd = v3p(c);
    orc_textured_color(s, m->specular, s->material_maps ? &s->material_maps[2 * mi + 1] : 0, h->uvw, c); *ks = v3p(c);
}

/* Attenuation, FIN/include/materials.h:60-66 (exp(float) -> expf, see header note) */
static v3 attenuation(v3 absorption, float l)
{
    return V3(expf(-absorption.x * l), expf(-absorption.y * l), expf(-absorption.z * l));
}

static float gray3(v3 c);
static float clampf(float v, float lo, float hi);
static void shade_fin(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h, int bounce, float out[3]);
static void shade_p13(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h, int bounce, int specount, float out[3]);
static void shade_p6(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h, int bounce, float out[3]);
static void shade_p3(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h, float out[3]);

void orc_shade(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *h, int bounce, float out[3])
{
    if (P->shade_model == RT_SHADE_P13 || P->shade_model == RT_SHADE_P12) shade_p13(s, P, ray, h, bounce, 0, out);   /* Shade(..., bouncelimit, 0), P13/main.cpp:286 */
    else if (P->shade_model == RT_SHADE_P6) shade_p6(s, P, ray, h, bounce, out);
    else if (P->shade_model == RT_SHADE_P3) shade_p3(s, P, ray, h, out);
    else shade_fin(s, P, ray, h, bounce, out);
}

static const rt_blinn *hit_material(const orc_scene *s, const orc_hit *h)
{
    static const rt_blinn none = {{0}};
    int m = s->nodes[h->node].material;
    if (m < 0 || m >= s->n_materials) return &none;
    return &s->materials[m];
}

/* MtlBlinn::Shade, FIN/main.cpp:516-708 */
static void shade_fin(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *hInfo, int bounceCount, float out[3])
{
    const rt_blinn *mtl = hit_material(s, hInfo);
    v3 color = v3p(mtl->emission);                                         /* :517 */
    const v3 p = v3p(hInfo->p);
    v3 N = vnorm(v3p(hInfo->N));                                           /* :521-522 */
    v3 direction = vnorm(vneg(v3p(ray + 3)));                              /* :523-524 */
    v3 kd, ks;
    material_colors(s, hInfo, mtl, &kd, &ks);                              /* diffuse/specular .Sample(uvw, duvw) :531-532 */
    float gloss = mtl->glossiness;
    v3 reflection = v3p(mtl->reflection), refraction = v3p(mtl->refraction);
    float ior = mtl->ior;
    v3 absorption = v3p(mtl->absorption);

    const float coef = s->n_lights == 0 ? 1.0f : 1.0f / s->n_lights;       /* :545 */
    for (int li = 0; li < s->n_lights; li++) {
        const rt_light *light = &s->lights[li];
        float Il[3];
        orc_illuminate(s, P, light, hInfo->p, (const float *)&N, Il);
        v3 intensity = vscale(v3p(Il), coef);                              /* :551 */
        if (hInfo->front) {
            if (light->type != RT_LIGHT_AMBIENT) {
                v3 L = vscale(light_direction(light, p), (float)(-1));     /* :556 */
                L = vnorm(L);
                v3 H = vnorm(vadd(L, direction));
                float cosNL = RMAX(0.f, vdot(N, L));
                float cosNH = RMAX(0.f, vdot(N, H));
                v3 diffuse = vscale(vmul(kd, intensity), cosNL);           /* :563 */
                v3 specular = vscale(vscale(vmul(ks, intensity), powf(cosNH, gloss)), cosNL);   /* :564 */
                color = vadd(color, vadd(diffuse, specular));              /* :566 */
            } else {
                orc_illuminate(s, P, light, hInfo->p, (const float *)&N, Il);
                intensity = v3p(Il);                                       /* :568 */
                color = vadd(color, vmul(kd, intensity));                  /* :569 */
            }
        }
    }

    /* reflection / refraction set-up, :577-610 */
    float ein = 1, eout = ior;
    if (!hInfo->front) { ein = ior; eout = 1; }
    float eta = ein / eout;
    v3 Y = vdot(N, direction) > 0.f ? N : vneg(N);
    const v3 Z = vcross(direction, Y);
    v3 X = vnorm(vcross(Y, Z));
    float cosI = vdot(N, direction);
    float sinI = sqrtf(1 - cosI * cosI);
    float sinO = RMAX(0.f, RMIN(1.f, sinI * eta));
    float cosO = sqrtf(1.f - sinO * sinO);
    v3 tDir = vsub(vscale(vneg(X), sinO), vscale(Y, cosO));                /* -X*sinO - Y*cosO */
    v3 rDir = vsub(vscale(vscale(N, 2.f), vdot(N, direction)), direction); /* 2.f*N*(N.d) - d */
    const float C0 = (eta - 1.f) * (eta - 1.f) / ((eta + 1.f) * (eta + 1.f));
    float rC = C0 + (1.f - C0) * powf(1.f - fabsf(cosI), 5.f);
    const float tC = 1.f - rC;
    const int totReflection = (eta * sinI) > 1.001f;                       /* materials.h:20 */
    const v3 tK = totReflection ? V3(0.f, 0.f, 0.f) : vscale(refraction, tC);
    const v3 rK = totReflection ? vadd(reflection, refraction) : vadd(reflection, vscale(refraction, rC));
    const float thr = 0.001f;                                              /* materials.h:21-22 */

    if (bounceCount > 0 && (rK.x > thr || rK.y > thr || rK.z > thr)) {     /* :613-623 */
        float r[6];
        st3(r, p); st3(r + 3, vnorm(rDir));
        orc_hit hh;
        g_cnt.rays_reflect++;
        if (orc_trace(s, P->shade_model, r, &hh)) {
            v3 K = vmul(rK, hh.front ? V3(1.f, 1.f, 1.f) : attenuation(absorption, hh.z));
            float c[3];
            const uint32_t me = g_rng.node;
            g_rng.node = child_node(me, 1u);
            shade_fin(s, P, r, &hh, bounceCount - 1, c);
            g_rng.node = me;
            color = vadd(color, vmul(K, v3p(c)));
        }
    }
    if (bounceCount > 0 && (tK.x > thr || tK.y > thr || tK.z > thr)) {     /* :625-638 */
        float r[6];
        st3(r, p); st3(r + 3, vnorm(tDir));
        orc_hit hh;
        g_cnt.rays_refract++;
        if (orc_trace(s, P->shade_model, r, &hh)) {
            v3 K = vmul(tK, hh.front ? V3(1.f, 1.f, 1.f) : attenuation(absorption, hh.z));
            float c[3];
            const uint32_t me = g_rng.node;
            g_rng.node = child_node(me, 2u);
            shade_fin(s, P, r, &hh, bounceCount - 1, c);
            g_rng.node = me;
            color = vadd(color, vmul(K, v3p(c)));
        } else {
            color = vadd(color, vmul(tK, environment_color(s, vnorm(tDir))));  /* environment.SampleEnvironment(tRay.dir) */
        }
    }
    v3 idr = V3(0, 0, 0);
    if (bounceCount == P->bounce) {
        /* :642-693 -- HEMISPHERE_SAMPLE rays are traced and shaded, but the result is assigned
         * to a shadowing inner variable (:676) and the outer idrColor (:668) stays 0, so the
         * block adds exactly 0.  Not traced by default (SURVEY.md section 0 finding 2); with
         * orc_set_trace_discarded(1) the rays ARE traced and shaded the way the reference spends
         * its time on them (bench.py's cpu_baseline), and still thrown away. */
        if (g_trace_discarded) {
            v3 newz = v3p(hInfo->N);
            v3 newx = vdot(newz, V3(1, 0, 0)) < 0.4 ? vcross(newz, V3(1, 0, 0)) : vcross(newz, V3(0, 0, 1));
            newx = vnorm(newx);
            v3 newy = vcross(newz, newx);
            int Nofsample = P->hemisphere_sample;
            const uint32_t me = g_rng.node;
            for (int i = 0; i < Nofsample; i++) {
                float u0, u1;
                rng2(RNG_GI, (uint32_t)i, &u0, &u1);
                float phi = (float)(2 * M_PI * (double)u0);
                float cosphi = cosf(phi);
                float ysquare = u1;
                float sintheta = sqrtf(ysquare), costheta = sqrtf(1 - ysquare);
                v3 hemis_dir = vnorm(vadd(vadd(vscale(newx, sintheta * cosphi), vscale(newy, sintheta * sinf(phi))), vscale(newz, costheta)));
                if (vdot(hemis_dir, newz) < 0.0) continue;
                float r[6], c[3];
                orc_hit hh;
                st3(r, p); st3(r + 3, vnorm(hemis_dir));
                g_discarded_rays++;
                if (orc_trace(s, P->shade_model, r, &hh)) {
                    g_rng.node = child_node(me, 64u + (uint32_t)i);
                    shade_fin(s, P, r, &hh, bounceCount - 1, c);         /* Shade(..., bounceCount-1, 1): result dropped */
                    g_rng.node = me;
                }
            }
        }
    } else {
        /* :695-705 */
        float irr[3], dir[3];
        if (s->n_photons > 0) {
            orc_estimate_irradiance(s->photons, s->n_photons, P->knn_k, P->knn_radius,
                                    hInfo->p, (const float *)&N, irr, dir);
            float theta = vdot(N, vneg(v3p(dir)));
            theta = (theta > 0.0 ? theta : 0.0f);
            idr = vadd(idr, vscale(vmul(kd, v3p(irr)), theta));            /* kd * photonrad * theta */
        }
    }
    color = vadd(color, idr);
    st3(out, color);
}

/* MtlBlinn::Shade, P13/main.cpp:485-756 (glossiness jitter needs rand(): only the
 * deterministic reflectionGlossiness == refractionGlossiness == 0 case is restated). */
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   /* P13/main.cpp:98-106 */
static void shade_p13(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *hInfo, int bounceCount, int specount, float out[3])
{
    const rt_blinn *m = hit_material(s, hInfo);
    const int p12 = P->shade_model == RT_SHADE_P12;
    v3 ra_color = V3(0, 0, 0), re_color = V3(0, 0, 0), re_ra_color;
    v3 ambient_color = V3(0, 0, 0), diffuse_color = V3(0, 0, 0), cau_Color = V3(0, 0, 0);
    v3 N = v3p(hInfo->N), Pp = v3p(hInfo->p);
    v3 Kd, Ks;
    material_colors(s, hInfo, m, &Kd, &Ks);
    float alpha = m->glossiness;
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light *l = &s->lights[i];
        float Il[3];
        if (l->type == RT_LIGHT_AMBIENT) {
            orc_illuminate(s, P, l, hInfo->p, hInfo->N, Il);
            ambient_color = vadd(ambient_color, vmul(v3p(Il), Kd));                   /* :510 */
        } else {
            /* the caustic lookup the reference keeps in a comment (P13/main.cpp:518-533), live when rt_params.caustic_k > 0:
             * causticmap.EstimateIrradiance<k>(causticrad, dirc, radius, hInfo.p, &N, ...); cau_Color += Kd*causticrad*theta */
            if (P->caustic_k > 0 && s->n_caustic > 0 && gray3(v3p(m->diffuse)) > 0 && specount > 2) {
                float irr[3], dirc[3];
                orc_estimate_irradiance(s->caustic, s->n_caustic, P->caustic_k, P->caustic_radius, hInfo->p, hInfo->N, irr, dirc);
                float theta = vdot(N, vneg(v3p(dirc)));
                theta = (theta > 0.0 ? theta : 0.0f);
                cau_Color = vadd(cau_Color, vscale(vmul(Kd, v3p(irr)), theta));
            }
            specount++;
            orc_illuminate(s, P, l, hInfo->p, hInfo->N, Il);
            v3 I_i = v3p(Il);
            v3 L = vscale(light_direction(l, Pp), (float)-1);
            if (!p12) L = vnorm(L);                                                   /* P13 adds L.Normalize() (:540) */
            v3 V = vnorm(vneg(v3p(ray + 3)));
            v3 H = vnorm(vadd(L, V));
            v3 kse = vadd(vscale(Ks, powf(vdot(N, H), alpha)), Kd);                   /* :547 */
            float theta = vdot(N, L);
            diffuse_color = vadd(diffuse_color, vmul(vscale(I_i, (theta > 0.0 ? theta : 0.0f)), kse));   /* :551 */
        }
    }
    const uint32_t me = g_rng.node;
    v3 all = vadd(ambient_color, vadd(diffuse_color, cau_Color));                     /* :622 all = ambient + (diffuse + idr + cau), idr = 0 */
    if (p12) {
        /* RayTracingProj12 main.cpp:393-448: cosine-weighted hemisphere rays, HEMISPHERE_SAMPLE at
         * the primary hit and 1 below it; all = ambient + ((diffuse/pi) + idr)*Kd */
        v3 idr = V3(0, 0, 0);
        if (bounceCount > 0) {
            v3 newz = v3p(hInfo->N);
            v3 newx = vdot(newz, V3(1, 0, 0)) < 0.4 ? vcross(newz, V3(1, 0, 0)) : vcross(newz, V3(0, 0, 1));
            newx = vnorm(newx);
            v3 newy = vcross(newz, newx);
            int Nof = (bounceCount == P->bounce) ? P->hemisphere_sample : 1;
            for (int i = 0; i < Nof; i++) {
                float u0, u1;
                rng2(RNG_GI, (uint32_t)i, &u0, &u1);
                float phi = (float)(2 * M_PI * u0);
                float cosphi = cosf(phi);
                float ysquare = u1;
                float sintheta = sqrtf(ysquare);
                float costheta = sqrtf(1 - ysquare);
                v3 hd = vadd(vadd(vscale(newx, sintheta * cosphi), vscale(newy, sintheta * sinf(phi))), vscale(newz, costheta));
                hd = vnorm(hd);
                float dotN_wi = vdot(hd, newz);
                float r[6];
                st3(r, Pp); st3(r + 3, vnorm(hd));
                orc_hit hh;
                v3 ic;
                g_cnt.rays_refract++;                      /* counted with the secondary rays */
                if (orc_trace(s, P->shade_model, r, &hh)) {
                    float c[3];
                    g_rng.node = child_node(me, 3u + (uint32_t)i);
                    shade_p13(s, P, r, &hh, bounceCount - 1, specount, c);
                    g_rng.node = me;
                    ic = v3p(c);
                } else ic = environment_color(s, v3p(r + 3));
                ic = vscale(ic, dotN_wi);
                idr = vadd(idr, vdivs(vscale(ic, 1.0f), (float)Nof));
            }
        }
        all = vadd(ambient_color, vmul(vadd(vdivs(diffuse_color, (float)M_PI), idr), Kd));
    }
    v3 V = vneg(vnorm(v3p(ray + 3)));                                                 /* :632 */
    if (bounceCount > 0) {                                                            /* :633-663 */
        v3 newN = N;
        if (m->reflection_glossiness) {                                               /* :635-647 */
            v3 newx = vcross(newN, V3(1, 0, 0));
            v3 newy = vcross(newN, newx);
            float u0, u1;
            rng2(RNG_GLOSSR, 0, &u0, &u1);
            float r = sqrtf(u0) * m->reflection_glossiness;
            float theta = (float)(M_PI * 2.0 * (double)u1);
            float dx = r * cosf(theta), dy = r * sinf(theta);
            newN = vnorm(vadd(newN, vadd(vscale(newx, dx), vscale(newy, dy))));
        }
        N = newN;
        float costheta = clampf(vdot(N, V), -1.0f, 1.0f);
        v3 R = vsub(vscale(N, 2 * costheta), V);
        float r[6];
        st3(r, Pp); st3(r + 3, vnorm(R));
        orc_hit hh;
        g_cnt.rays_reflect++;
        if (orc_trace(s, P->shade_model, r, &hh)) {
            float c[3];
            g_rng.node = child_node(me, 1u);
            shade_p13(s, P, r, &hh, bounceCount - 1, specount, c);
            g_rng.node = me;
            re_color = v3p(c);
        } else re_color = environment_color(s, v3p(r + 3));
    }
    all = vadd(all, vmul(re_color, v3p(m->reflection)));                              /* :665 */
    if (bounceCount > 0) {                                                            /* :671-751 */
        N = v3p(hInfo->N);
        if (m->refraction_glossiness) {                                               /* :673-686 */
            v3 newx = vcross(N, V3(1, 0, 0));
            v3 newy = vcross(N, newx);
            float u0, u1;
            rng2(RNG_GLOSST, 0, &u0, &u1);
            float r = sqrtf(u0) * m->refraction_glossiness;
            float theta = (float)(M_PI * 2.0 * (double)u1);
            float dx = r * cosf(theta), dy = r * sinf(theta);
            N = vnorm(vadd(N, vadd(vscale(newx, dx), vscale(newy, dy))));
        }
        float R0 = 0.0f, re_ratio = 0.0f, ra_ratio = 0.0f;
        V = vnorm(V);
        float costheta1 = fabsf(vdot(V, N));
        float sintheta1 = sqrtf(RMAX(0.0f, 1 - (costheta1 * costheta1)));
        float n1 = 1.0, n2 = 1.0;
        if (hInfo->front) n2 = m->ior;
        else { n1 = m->ior; N = vneg(N); }
        float ratio_n = n1 / n2;
        float sintheta2 = ratio_n * sintheta1;
        orc_hit hh;
        hit_init(&hh);
        float absorb = 1.0;
        if (sintheta2 <= 1.0) {
            float costheta2 = sqrtf(RMAX(0.0f, 1 - (sintheta2 * sintheta2)));
            v3 S = vcross(N, vcross(N, V));
            N = vnorm(N);
            S = vnorm(S);
            v3 T = vadd(vscale(vneg(N), costheta2), vscale(S, sintheta2));
            float r[6];
            st3(r, Pp); st3(r + 3, T);
            g_cnt.rays_refract++;
            if (orc_trace(s, P->shade_model, r, &hh)) {
                float c[3];
                g_rng.node = child_node(me, 2u);
                shade_p13(s, P, r, &hh, bounceCount - 1, specount, c);
                g_rng.node = me;
                ra_color = v3p(c);
            } else ra_color = environment_color(s, v3p(r + 3));
            absorb = expf(-m->absorption[0] * hh.z);                                  /* :728 */
            R0 = (n1 - n2) / (n1 + n2);
            R0 = R0 * R0;
            double tmp = 1.0 - costheta1;
            re_ratio = (float)(R0 + (1.0 - R0) * pow(tmp, 5.0));
            ra_ratio = (float)(1.0 - re_ratio);
        } else re_ratio = 1.0f;
        re_ra_color = re_color;
        /* refraction * (ra_ratio*absorb*ra_color + re_ratio*re_ra_color) */
        all = vadd(all, vmul(v3p(m->refraction),
                             vadd(vscale(ra_color, ra_ratio * absorb), vscale(re_ra_color, re_ratio))));
    }
    st3(out, all);
}

/* MtlBlinn::Shade of RayTracingProj3, main.cpp:152-190: ambient + Blinn with V = camera.pos - p
 * (every P3 ray starts at the camera, so camera.pos is the ray origin) */
static void shade_p3(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *hInfo, float out[3])
{
    const rt_blinn *m = hit_material(s, hInfo);
    v3 ambient = V3(0, 0, 0), diffuse = V3(0, 0, 0);
    v3 N = v3p(hInfo->N), Pp = v3p(hInfo->p);
    v3 Kd = v3p(m->diffuse), Ks = v3p(m->specular);
    float alpha = m->glossiness;
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light *l = &s->lights[i];
        float Il[3];
        orc_illuminate(s, P, l, hInfo->p, hInfo->N, Il);
        if (l->type == RT_LIGHT_AMBIENT) ambient = vadd(ambient, vmul(v3p(Il), Kd));
        else {
            v3 L = vscale(light_direction(l, Pp), (float)-1);
            v3 V = vnorm(vsub(v3p(ray), Pp));                             /* camera.pos - hInfo.p */
            v3 LpV = vadd(L, V);
            v3 H = vnorm(vdivs(LpV, vlen(LpV)));                          /* LpV/LpV.Length(); H.Normalize() */
            v3 kse = vadd(vscale(Ks, powf(vdot(N, H), alpha)), Kd);
            float theta = vdot(N, L);
            diffuse = vadd(diffuse, vmul(vscale(v3p(Il), (theta > 0 ? theta : 0)), kse));
        }
    }
    st3(out, vadd(ambient, diffuse));
}

/* MtlBlinn::Shade of RayTracingProj6, main.cpp:175-340 */
static void shade_p6(const orc_scene *s, const rt_params *P, const float ray[6], const orc_hit *hInfo, int bounceCount, float out[3])
{
    const rt_blinn *m = hit_material(s, hInfo);
    v3 ra_color = V3(0, 0, 0), re_color = V3(0, 0, 0), re_ra_color;
    v3 ambient = V3(0, 0, 0), diffuse = V3(0, 0, 0);
    v3 N = v3p(hInfo->N), Pp = v3p(hInfo->p);
    v3 Kd = v3p(m->diffuse), Ks = v3p(m->specular);
    v3 reflection = v3p(m->reflection), refraction = v3p(m->refraction);
    float alpha = m->glossiness;
    for (int i = 0; i < s->n_lights; i++) {
        const rt_light *l = &s->lights[i];
        float Il[3];
        orc_illuminate(s, P, l, hInfo->p, hInfo->N, Il);
        if (l->type == RT_LIGHT_AMBIENT) ambient = vadd(ambient, vmul(v3p(Il), Kd));            /* :199 */
        else {
            v3 L = vscale(light_direction(l, Pp), (float)-1);
            v3 V = vnorm(vneg(v3p(ray + 3)));
            v3 H = vnorm(vadd(L, V));
            v3 kse = vadd(vscale(Ks, powf(vdot(N, H), alpha)), Kd);                             /* :210 */
            float theta = vdot(N, L);
            diffuse = vadd(diffuse, vmul(vscale(v3p(Il), (theta > 0.0 ? theta : 0.0f)), kse));  /* :214 */
        }
    }
    v3 all = vadd(ambient, diffuse);
    v3 V = vneg(vnorm(v3p(ray + 3)));                                                            /* :223 */
    const int has_re = gray3(reflection) > 0;
    if (has_re && bounceCount > 0) {                                                             /* :225-238 */
        float costheta = clampf(vdot(N, V), -1.0f, 1.0f);
        v3 R = vnorm(vsub(vscale(N, 2 * costheta), V));
        float r[6];
        st3(r, Pp); st3(r + 3, R);
        orc_hit hh;
        g_cnt.rays_reflect++;
        if (orc_trace(s, P->shade_model, r, &hh)) { float c[3]; shade_p6(s, P, r, &hh, bounceCount - 1, c); re_color = v3p(c); }
    }
    all = vadd(all, vmul(re_color, reflection));                                                 /* :240 */
    if (gray3(refraction) > 0 && bounceCount > 0) {                                              /* :246-335 */
        float R0 = 0.0f, re_ratio = 0.0f, ra_ratio = 0.0f;
        V = vnorm(V);
        float costheta1 = fabsf(vdot(V, N));
        float sintheta1 = sqrtf(RMAX(0.0f, 1 - (costheta1 * costheta1)));
        float n1 = 1.0, n2 = 1.0;
        if (hInfo->front) n2 = m->ior;
        else { n1 = m->ior; N = vneg(v3p(hInfo->N)); }
        float ratio_n = n1 / n2;
        float sintheta2 = ratio_n * sintheta1;
        float absorb = 0.0;
        if (sintheta2 <= 1.0) {
            float costheta2 = sqrtf(RMAX(0.0f, 1 - (sintheta2 * sintheta2)));
            v3 S = vcross(N, vcross(N, V));
            N = vnorm(N);
            S = vnorm(S);
            v3 T = vadd(vscale(vneg(N), costheta2), vscale(S, sintheta2));
            float r[6];
            st3(r, Pp); st3(r + 3, T);
            orc_hit hh;
            g_cnt.rays_refract++;
            if (orc_trace(s, P->shade_model, r, &hh)) {
                float c[3];
                shade_p6(s, P, r, &hh, bounceCount - 1, c);
                ra_color = v3p(c);
                absorb = expf(-m->absorption[0] * hh.z);                                         /* :296 */
                R0 = (n1 - n2) / (n1 + n2);
                R0 = R0 * R0;
                double tmp = 1.0 - costheta1;
                re_ratio = (float)(R0 + (1.0 - R0) * pow(tmp, 5.0));
                ra_ratio = (float)(1.0 - re_ratio);
            }
        } else re_ratio = 1.0f;
        re_ra_color = re_color;
        if (has_re && bounceCount > 0) re_ra_color = re_color;
        else if (re_ratio > 0.0 && bounceCount > 0) {                                            /* :314-329 */
            float costheta = clampf(vdot(N, V), -1.0f, 1.0f);
            v3 R = vsub(vscale(N, 2 * costheta), V);                                             /* not normalised */
            float r[6];
            st3(r, Pp); st3(r + 3, R);
            orc_hit hh;
            g_cnt.rays_reflect++;
            if (orc_trace(s, P->shade_model, r, &hh)) { float c[3]; shade_p6(s, P, r, &hh, bounceCount - 1, c); re_ra_color = v3p(c); }
        }
        all = vadd(all, vmul(refraction, vadd(vscale(ra_color, ra_ratio * absorb), vscale(re_ra_color, re_ratio))));   /* :332 */
    }
    st3(out, all);
}

/* ---- photon map -------------------------------------------------------------------------- */
/* Photon::SetPower / SetDirection, FIN/include/cyPhotonMap.h:139-156; AddPhoton :184-192
 * (a fresh Photon's planeAndDirZ is uninitialised in the reference; 0 here). */
void orc_photon_pack(const float pos[3], const float dir[3], const float pw[3], rt_photon *out)
{
    memset(out, 0, sizeof *out);
    out->position[0] = pos[0]; out->position[1] = pos[1]; out->position[2] = pos[2];
    out->dir_x = (int16_t)(dir[0] * 0x7FFF);
    out->dir_y = (int16_t)(dir[1] * 0x7FFF);
    if (dir[2] > 0) out->plane_and_dirz &= 0x7;
    else out->plane_and_dirz = 0x8 | (out->plane_and_dirz & 0x7);
    float power = pw[0];
    if (power < pw[1]) power = pw[1];
    if (power < pw[2]) power = pw[2];
    out->power = power;
    float c[3] = { pw[0] / power, pw[1] / power, pw[2] / power };
    orc_color24(c, out->color);
}

/* Photon::GetDirection, FIN/include/cyPhotonMap.h:158-180 -- including the reference's
 * `dirX*dirX + dirY-dirY` typo (:162): z is derived from x alone. */
void orc_photon_direction(const rt_photon *p, float dir[3])
{
    int dirX = p->dir_x, dirY = p->dir_y;
    dir[0] = (float)dirX / (float)0x7FFF;
    dir[1] = (float)dirY / (float)0x7FFF;
    int dirXY2 = dirX * dirX + dirY - dirY;
    if (dirXY2 > 0x3FFF0001) dirXY2 = 0x3FFF0001;
    int dirZ2 = 0x3FFF0001 - dirXY2;
    int dirZ = 0;
    int place = 0x40000000;
    int remainder = dirZ2;
    while (place > remainder) place = place >> 2;
    while (place) {
        if (remainder >= dirZ + place) {
            remainder = remainder - dirZ - place;
            dirZ = dirZ + (place << 1);
        }
        dirZ = dirZ >> 1;
        place = place >> 2;
    }
    dir[2] = (float)dirZ / (float)0x7FFF;
    if (p->plane_and_dirz & 0x8) dir[2] = -dir[2];
}

/* Photon::GetPower, cyPhotonMap.h:58: color.ToColor()*power, ToColor = c/255.0f */
void orc_photon_power(const rt_photon *p, float rgb[3])
{
    rgb[0] = (p->color[0] / 255.0f) * p->power;
    rgb[1] = (p->color[1] / 255.0f) * p->power;
    rgb[2] = (p->color[2] / 255.0f) * p->power;
}

/* PhotonMap::BalanceSegment, FIN/include/cyPhotonMap.h:222-284 */
static void balance_segment(rt_photon *ph, rt_photon *bal, v3 boxMin, v3 boxMax,
                            uint32_t index, uint32_t start, uint32_t end)
{
    uint32_t median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;
    int axis = 2;
    v3 boxDif = vsub(boxMax, boxMin);
    if (boxDif.x > boxDif.y) { if (boxDif.x > boxDif.z) axis = 0; }
    else if (boxDif.y > boxDif.z) axis = 1;
    uint32_t left = start, right = end;
#define SWAP(i, j) do { rt_photon t_ = ph[i]; ph[i] = ph[j]; ph[j] = t_; } while (0)
    while (right > left) {
        const float v = ph[right].position[axis];
        uint32_t i = left - 1;
        uint32_t j = right;
        while (ph[++i].position[axis] < v);
        while (ph[--j].position[axis] > v && j > left);
        while (i < j) {
            SWAP(i, j);
            while (ph[++i].position[axis] < v);
            while (ph[--j].position[axis] > v && j > left);
        }
        SWAP(i, right);
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
#undef SWAP
    bal[index] = ph[median];
    bal[index].plane_and_dirz = (uint8_t)((bal[index].plane_and_dirz & 0x8) | (axis & 0x3));   /* SetPlane :63 */
    if (median > start) {
        if (start < median - 1) {
            v3 tBoxMax = boxMax;
            ((float *)&tBoxMax)[axis] = bal[index].position[axis];
            balance_segment(ph, bal, boxMin, tBoxMax, 2 * index, start, median - 1);
        } else bal[2 * index] = ph[start];
    }
    if (median < end) {
        if (median + 1 < end) {
            v3 tBoxMin = boxMin;
            ((float *)&tBoxMin)[axis] = bal[index].position[axis];
            balance_segment(ph, bal, tBoxMin, boxMax, 2 * index + 1, median + 1, end);
        } else bal[2 * index + 1] = ph[end];
    }
}

/* PhotonMap::PrepareForIrradianceEstimation, cyPhotonMap.h:196-218.  The bounding box loop
 * starts from photons[0] (the unused slot) exactly like the reference. */
void orc_photon_balance(rt_photon *in, uint32_t n, rt_photon *out)
{
    v3 boxMin = v3p(in[0].position), boxMax = boxMin;
    for (uint32_t i = 1; i <= n; i++) {
        const float *q = in[i].position;
        if (boxMin.x > q[0]) boxMin.x = q[0];
        if (boxMax.x < q[0]) boxMax.x = q[0];
        if (boxMin.y > q[1]) boxMin.y = q[1];
        if (boxMax.y < q[1]) boxMax.y = q[1];
        if (boxMin.z > q[2]) boxMin.z = q[2];
        if (boxMax.z < q[2]) boxMax.z = q[2];
    }
    memset(out, 0, sizeof(rt_photon) * ((size_t)n + 1));   /* std::vector value-initialises */
    if (n >= 1) balance_segment(in, out, boxMin, boxMax, 1, 1, n);
}

typedef struct {
    v3 pos; const v3 *normal; int maxPhotons; int found;
    float *dist2; rt_photon *photon;
    const rt_photon *photons; int half;
} nearest_t;

/* PhotonMap::LocatePhotons, cyPhotonMap.h:365-440 (normScale == 0 since ellipticity == 1) */
static void locate_photons(nearest_t *np, const int index)
{
    const rt_photon *p = &np->photons[index];
    g_cnt.photons_visited++;
    int axis = p->plane_and_dirz & 0x3;
    if (index < np->half) {
        float dist = ((const float *)&np->pos)[axis] - p->position[axis];
        if (dist > 0) {
            locate_photons(np, 2 * index + 1);
            if (dist * dist < np->dist2[0]) locate_photons(np, 2 * index);
        } else {
            locate_photons(np, 2 * index);
            if (dist * dist < np->dist2[0]) locate_photons(np, 2 * index + 1);
        }
    }
    v3 dif = vsub(v3p(p->position), np->pos);
    float dist2 = vlen2(dif);
    if (dist2 < np->dist2[0]) {
        if (np->normal) {
            float d[3];
            orc_photon_direction(p, d);
            if (vdot(v3p(d), *np->normal) >= 0) return;
        }
        if (np->found < np->maxPhotons) {
            np->found++;
            np->dist2[np->found] = dist2;
            np->photon[np->found] = *p;
            if (np->found == np->maxPhotons) {
                int half_found = np->found >> 1;
                for (int k = half_found; k >= 1; k--) {
                    int parent = k;
                    rt_photon tp = np->photon[k];
                    float td2 = np->dist2[k];
                    while (parent <= half_found) {
                        int j = parent + parent;
                        if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
                        if (td2 >= np->dist2[j]) break;
                        np->dist2[parent] = np->dist2[j];
                        np->photon[parent] = np->photon[j];
                        parent = j;
                    }
                    np->photon[parent] = tp;
                    np->dist2[parent] = td2;
                }
            }
        } else {
            int parent = 1;
            int j = 2;
            while (j <= np->found) {
                if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
                if (dist2 > np->dist2[j]) break;
                np->dist2[parent] = np->dist2[j];
                np->photon[parent] = np->photon[j];
                parent = j;
                j <<= 1;
            }
            np->photon[parent] = *p;
            np->dist2[parent] = dist2;
            np->dist2[0] = np->dist2[1];
        }
    }
}

/* PhotonMap::EstimateIrradiance<k>(irr, dir, radius, pos, &normal, 1, CONSTANT),
 * cyPhotonMap.h:288-336.  halfStoredPhotons = (size-1)/2 - 1 (:217). */
void orc_estimate_irradiance(const rt_photon *photons, uint32_t n, int k, float radius,
                             const float pos[3], const float normal[3], float irr[3], float dir[3])
{
    irr[0] = irr[1] = irr[2] = 0; dir[0] = dir[1] = dir[2] = 0;
    g_cnt.photon_queries++;
    if (n == 0 || k <= 0) return;
    float *d2 = (float *)malloc(sizeof(float) * ((size_t)k + 1));
    rt_photon *fp = (rt_photon *)malloc(sizeof(rt_photon) * ((size_t)k + 1));
    v3 nrm = normal ? v3p(normal) : V3(0, 0, 0);
    nearest_t np;
    np.pos = v3p(pos); np.normal = normal ? &nrm : 0; np.maxPhotons = k; np.found = 0;
    np.dist2 = d2; np.photon = fp; np.photons = photons; np.half = (int)(n / 2) - 1;
    np.dist2[0] = radius * radius;
    locate_photons(&np, 1);
    v3 I = V3(0, 0, 0), D = V3(0, 0, 0);
    for (int i = 1; i <= np.found; i++) {
        float pw[3], dd[3];
        orc_photon_power(&np.photon[i], pw);
        float filter = 1;
        I = vadd(I, vscale(v3p(pw), filter));                                 /* irrad += filter*power */
        orc_photon_direction(&np.photon[i], dd);
        D = vadd(D, vscale(v3p(dd), filter * np.photon[i].power));            /* dir*(filter*GetMaxPower) */
    }
    if (np.found > 0) {
        float area = (float)M_PI * np.dist2[0];
        if (area > 0) { const float one_over_area = 1.0f / area; I = vscale(I, one_over_area); }
        D = vnorm(D);
    }
    st3(irr, I); st3(dir, D);
    free(d2); free(fp);
}

/* ---- cyBVH build -------------------------------------------------------------------------- */
typedef struct tnode { struct tnode *c1, *c2; float box[6]; uint32_t count, offset; } tnode;

static void ebounds(const float *v, const uint32_t *f, uint32_t i, float box[6])
{   /* BVHTriMesh::GetElementBounds, FIN/include/cyBVH.h:356-368 */
    const uint32_t *fi = f + 3 * (size_t)i;
    const float *p = v + 3 * (size_t)fi[0];
    box[0] = box[3] = p[0]; box[1] = box[4] = p[1]; box[2] = box[5] = p[2];
    for (int j = 1; j < 3; j++) {
        p = v + 3 * (size_t)fi[j];
        for (int k = 0; k < 3; k++) {
            if (box[k] > p[k]) box[k] = p[k];
            if (box[k + 3] < p[k]) box[k + 3] = p[k];
        }
    }
}
static float ecenter(const float *v, const uint32_t *f, uint32_t i, int dim)
{   /* GetElementCenter, cyBVH.h:371-375 */
    const uint32_t *fi = f + 3 * (size_t)i;
    return (v[3 * (size_t)fi[0] + dim] + v[3 * (size_t)fi[1] + dim] + v[3 * (size_t)fi[2] + dim]) / 3.0f;
}
static void box_init(float b[6]) { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }
static void box_add(float b[6], const float o[6])
{ for (int i = 0; i < 3; i++) { if (b[i] > o[i]) b[i] = o[i]; if (b[i + 3] < o[i + 3]) b[i + 3] = o[i + 3]; } }

/* MeanSplit, cyBVH.h:295-328 */
static uint32_t mean_split(const float *v, const uint32_t *f, uint32_t elementCount, uint32_t *ne,
                           const float *box, uint32_t maxPer)
{
    if (elementCount <= maxPer) return 0;
    float d[3] = { box[3] - box[0], box[4] - box[1], box[5] - box[2] };
    uint32_t sd[3];
    sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
    sd[1] = (sd[0] + 1) % 3;
    sd[2] = (sd[0] + 2) % 3;
    if (d[sd[1]] < d[sd[2]]) { uint32_t t = sd[1]; sd[1] = sd[2]; sd[2] = t; }
    uint32_t child1 = 0;
    for (int s = 0; s < 3; s++) {
        uint32_t splitDim = sd[s];
        float splitPos = 0.5f * (box[splitDim] + box[splitDim + 3]);
        uint32_t i = 0, j = elementCount;
        while (i < j) {
            float center = ecenter(v, f, ne[i], (int)splitDim);
            if (center <= splitPos) i++;
            else { j--; uint32_t t = ne[i]; ne[i] = ne[j]; ne[j] = t; }
        }
        if (i < elementCount && i > 0) { child1 = i; break; }
    }
    return child1;
}

/* SplitTempNode, cyBVH.h:242-278 (CY_BVH_MAX_ELEMENT_COUNT = 8) */
static void split_temp(const float *v, const uint32_t *f, uint32_t *elements, tnode *t, uint32_t maxPer)
{
    uint32_t *ne = &elements[t->offset];
    uint32_t c1 = mean_split(v, f, t->count, ne, t->box, maxPer);
    if (c1 == 0 || c1 >= t->count) {
        if (t->count > 8) c1 = t->count / 2;
        else return;
    }
    float b1[6], b2[6], eb[6];
    box_init(b1); box_init(b2);
    for (uint32_t i = 0; i < c1; i++) { ebounds(v, f, ne[i], eb); box_add(b1, eb); }
    for (uint32_t i = c1; i < t->count; i++) { ebounds(v, f, ne[i], eb); box_add(b2, eb); }
    t->c1 = (tnode *)calloc(1, sizeof(tnode));
    t->c2 = (tnode *)calloc(1, sizeof(tnode));
    t->c1->count = c1; t->c1->offset = t->offset; memcpy(t->c1->box, b1, sizeof b1);
    t->c2->count = t->count - c1; t->c2->offset = t->offset + c1; memcpy(t->c2->box, b2, sizeof b2);
    split_temp(v, f, elements, t->c1, maxPer);
    split_temp(v, f, elements, t->c2, maxPer);
}
static uint32_t tcount(const tnode *t) { return 1 + (t->c1 ? tcount(t->c1) : 0) + (t->c2 ? tcount(t->c2) : 0); }
/* ConvertTempData, cyBVH.h:281-291 */
static uint32_t convert_temp(rt_bvh_node *nodes, uint32_t id, const tnode *t, uint32_t childIndex)
{
    memcpy(nodes[id].box, t->box, sizeof t->box);
    if (!t->c1) {
        nodes[id].data = (t->offset & 0x0FFFFFFFu) | ((t->count - 1) << 28) | 0x80000000u;
        return childIndex;
    }
    nodes[id].data = childIndex & 0x7FFFFFFFu;
    uint32_t nci = convert_temp(nodes, childIndex, t->c1, childIndex + 2);
    return convert_temp(nodes, childIndex + 1, t->c2, nci);
}
static void tfree(tnode *t) { if (!t) return; tfree(t->c1); tfree(t->c2); free(t); }

/* BVH::Build, cyBVH.h:122-142 */
int orc_bvh_build(const float *v, const uint32_t *f, int32_t nf, int32_t max_per_leaf,
                  rt_bvh_node *nodes_out, uint32_t *elements_out)
{
    if (nf <= 0) return 0;
    uint32_t maxPer = (uint32_t)max_per_leaf;
    if (maxPer > 8) maxPer = 8;
    for (int32_t i = 0; i < nf; i++) elements_out[i] = (uint32_t)i;
    float box[6], eb[6];
    box_init(box);
    for (int32_t i = 0; i < nf; i++) { ebounds(v, f, (uint32_t)i, eb); box_add(box, eb); }
    tnode *root = (tnode *)calloc(1, sizeof(tnode));
    root->count = (uint32_t)nf; root->offset = 0; memcpy(root->box, box, sizeof box);
    split_temp(v, f, elements_out, root, maxPer);
    uint32_t n = tcount(root);
    memset(&nodes_out[0], 0, sizeof(rt_bvh_node));
    convert_temp(nodes_out, 1, root, 2);
    tfree(root);
    return (int)n + 1;
}

/* ---- RenderPixel -------------------------------------------------------------------------- */
typedef struct { v3 b; float u, v; float m[9]; } cam_setup;

/* camera set-up in RenderPixel, FIN/main.cpp:205-224 */
static void camera_setup(const rt_camera *cam, cam_setup *cs)
{
    float theta = cam->fov;
    float l = cam->focaldist;
    float h = (float)(2 * l * tan(theta / 2 * (M_PI / 180)));
    float w = h * (float)cam->width / cam->height;
    v3 b = V3(-w / 2, h / 2, -l);
    float u = w / cam->width;
    float v = -h / cam->height;
    float du = u / 2, dv = v / 2;
    b.x += du; b.y += dv;
    v3 up = v3p(cam->up);
    v3 z_new = vscale(v3p(cam->dir), (float)-1);
    v3 x_new = vcross(up, z_new);
    up = vnorm(up); z_new = vnorm(z_new); x_new = vnorm(x_new);
    cs->b = b; cs->u = u; cs->v = v;
    cs->m[0] = x_new.x; cs->m[1] = x_new.y; cs->m[2] = x_new.z;      /* Matrix3(x,y,z) = columns */
    cs->m[3] = up.x; cs->m[4] = up.y; cs->m[5] = up.z;
    cs->m[6] = z_new.x; cs->m[7] = z_new.y; cs->m[8] = z_new.z;
}

/* generateSample (FIN/main.cpp:147-162) + ray build (:288-292); with camera.dof != 0 the per-pixel
 * lens table (:246-262) and the per-sample pick (:284) draw from the counter RNG */
static void primary_ray_ms(const rt_camera *cam, const cam_setup *cs, int x, int y, int j, int max_sample, uint32_t seed, float ray[6])
{
    v3 tmp = vadd(V3(x * cs->u, y * cs->v, 0), cs->b);                 /* :235-236 */
    float sx = orc_halton(j, 2) * cs->u;
    float sy = cs->v * orc_halton(j, 3);
    sx += tmp.x; sy += tmp.y;
    v3 sample = V3(sx, sy, tmp.z);
    v3 d_campos = V3(0, 0, 0);
    if (cam->dof != 0) {
        const uint32_t pixel_id = (uint32_t)y * (uint32_t)cam->width + (uint32_t)x;
        const uint32_t sample_id = pixel_id * (uint32_t)max_sample + (uint32_t)j;
        float u0, u1;
        rng2_at(seed, sample_id, 0, RNG_PICK, 0, &u0, &u1);
        int pick = (int)(u0 * 64.0f);
        if (pick > 63) pick = 63;
        rng2_at(seed, pixel_id, 0, RNG_LENS, (uint32_t)(pick + 1), &u0, &u1);
        float r = orc_halton(pick + 1, 2);
        r = sqrtf(r) * cam->dof;
        float theta = (float)(M_PI * 2.0 * (double)u0);
        d_campos = mmul(cs->m, V3(r * cosf(theta), r * sinf(theta), 0));
    }
    v3 o = vadd(v3p(cam->pos), d_campos);
    v3 dir = mmul(cs->m, sample);
    dir = vsub(dir, d_campos);
    dir = vnorm(dir);
    st3(ray, o); st3(ray + 3, dir);
}

static void primary_ray(const rt_camera *cam, const cam_setup *cs, int x, int y, int j, float ray[6])
{
    primary_ray_ms(cam, cs, x, y, j, 1, 0, ray);          /* dof == 0 callers */
}

void orc_primary_ray(const rt_camera *cam, int x, int y, int j, float ray[6])
{
    cam_setup cs;
    camera_setup(cam, &cs);
    primary_ray(cam, &cs, x, y, j, ray);
}

/* VariantOverThreshold, FIN/main.cpp:164-189 (pow(float,int) -> double) */
static int variant_over_threshold(const float *list, int n, float threshold)
{
    float ninverse = (float)(1.0 / n);
    float sum[3] = { 0, 0, 0 }, sq[3] = { 0, 0, 0 };
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            float t = list[3 * i + c];
            sum[c] += t;
            sq[c] = (float)((double)sq[c] + pow((double)t, 2));
        }
    int over = 0;
    for (int c = 0; c < 3; c++) {
        float avg = ninverse * sum[c];
        float var = (float)((double)(sq[c] * ninverse) + pow((double)avg, 2) - (double)(2 * avg * ninverse * sum[c]));
        if (var > threshold) over = 1;
    }
    return over;
}

int orc_pixel_samples(const orc_scene *s, const rt_camera *cam, const rt_params *P,
                      int x, int y, int j0, int j1, float *rgb, uint8_t *hitmask, float *z)
{
    cam_setup cs;
    camera_setup(cam, &cs);
    int nh = 0;
    for (int j = j0; j < j1; j++) {
        float ray[6];
        primary_ray_ms(cam, &cs, x, y, j, P->max_sample, P->seed, ray);
        orc_set_rng(P->seed, ((uint32_t)y * (uint32_t)cam->width + (uint32_t)x) * (uint32_t)P->max_sample + (uint32_t)j, 1);
        orc_hit h;
        g_cnt.rays_primary++;
        hitmask[j - j0] = 0;
        rgb[3 * (j - j0)] = rgb[3 * (j - j0) + 1] = rgb[3 * (j - j0) + 2] = 0;
        if (orc_trace(s, P->shade_model, ray, &h)) {
            orc_shade(s, P, ray, &h, P->bounce, &rgb[3 * (j - j0)]);
            hitmask[j - j0] = 1;
            if (z) *z = h.z;
            nh++;
        }
    }
    return nh;
}

/* RenderPixel, FIN/main.cpp:202-344 over a pixel rectangle */
void orc_render(const orc_scene *s, const rt_camera *cam, const rt_params *P,
                int x0, int y0, int x1, int y1, uint8_t *rgb8, float *zbuf, uint8_t *count)
{
    cam_setup cs;
    camera_setup(cam, &cs);
    int has_children = 0;
    for (int i = 1; i < s->n_nodes; i++) if (s->nodes[i].parent == 0) has_children = 1;
    int maxs = P->max_sample > P->min_sample ? P->max_sample : P->min_sample;
    float *colorlist = (float *)malloc(sizeof(float) * 3 * (size_t)(maxs > 0 ? maxs : 1));
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
        int index = y * cam->width + x;
        if (!has_children) continue;                                   /* :266 */
        int hit = 0, ncol = 0;
        float hitz = 0;
        int s_start = 0, s_end = P->min_sample;
        while (s_start == 0 || (variant_over_threshold(colorlist, ncol, P->threshold) && s_start != P->max_sample)) {
            for (int k = s_start; k < s_end; k++) {
                float ray[6];
                primary_ray_ms(cam, &cs, x, y, k, P->max_sample, P->seed, ray);
                orc_set_rng(P->seed, ((uint32_t)y * (uint32_t)cam->width + (uint32_t)x) * (uint32_t)P->max_sample + (uint32_t)k, 1);
                orc_hit h;
                g_cnt.rays_primary++;
                if (orc_trace(s, P->shade_model, ray, &h)) {
                    hit = 1;
                    orc_shade(s, P, ray, &h, P->bounce, &colorlist[3 * ncol]);
                    ncol++;
                    hitz = h.z;
                }
            }
            s_start = s_end;
            s_end = P->max_sample;
            if (!hit) break;
            if (s_start >= s_end && s_start != P->max_sample) break;   /* guard: min > max */
        }
        float g[3];
        if (hit) {
            /* averageColor, :191-199 */
            float n = 1 / (float)ncol;
            v3 c = V3(0, 0, 0);
            for (int i = 0; i < ncol; i++) c = vadd(c, vscale(v3p(&colorlist[3 * i]), n));
            count[index] = (ncol <= P->min_sample) ? 0 : 255;          /* :312-315 */
            float ig = (float)(1.0 / (double)P->gamma);                /* powf(x, 1.0/gamma) */
            g[0] = powf(c.x, ig); g[1] = powf(c.y, ig); g[2] = powf(c.z, ig);
            orc_color24(g, &rgb8[3 * (size_t)index]);
            zbuf[index] = hitz;
        } else {
            float ig = (float)(1.0 / (double)P->gamma);
            float uvw[3] = { (float)x / cam->width, (float)y / cam->height, 0 }, bgc[3];      /* :326-328 */
            orc_textured_color(s, s->bg, s->bg_map, uvw, bgc);
            g[0] = powf(bgc[0], ig); g[1] = powf(bgc[1], ig); g[2] = powf(bgc[2], ig);
            orc_color24(g, &rgb8[3 * (size_t)index]);
            zbuf[index] = BIGFLOAT;
            count[index] = 0;
        }
    }
    free(colorlist);
}

/* ---- photon pass ----------------------------------------------------------------------------- */
/* The reference draws from libc rand(); this build replaces it by a counter-based generator
 * (Philox-4x32-10, key = (seed, 'RTMI'), counter = (attempt lo, attempt hi, block, 0)), the same
 * one the HIP photon kernel uses, so that both produce the same photons. */
typedef struct { uint32_t key0, key1, c0, c1, blk, o[4]; int used; } philox_t;
static void philox_refill(philox_t *g)
{
    uint32_t x0 = g->c0, x1 = g->c1, x2 = g->blk, x3 = 0, k0 = g->key0, k1 = g->key1;
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    g->o[0] = x0; g->o[1] = x1; g->o[2] = x2; g->o[3] = x3;
    g->blk++; g->used = 0;
}
static float philox_next(philox_t *g)     /* stands in for rand() / (float) RAND_MAX */
{
    if (g_script.on) return script_uniform();               /* test hook, see orc_script_begin */
    if (g->used >= 4) philox_refill(g);
    return (float)(g->o[g->used++] >> 8) * (1.0f / 16777216.0f);
}
static float gray3(v3 c) { return (c.x + c.y + c.z) / 3.0f; }      /* Color::Gray, cyColor.h */

/* MtlBlinn::RandomPhotonBounce, FIN/include/materials.h:99-256 (all branches, incl. the glossy ones).
 * sqrtf(1-cosI^2) is clamped at 0 like in the Shade restatement. */
static int random_photon_bounce(const rt_blinn *m, const orc_hit *h, float ray[6], v3 *c, philox_t *rng)
{
    const v3 V = vneg(v3p(ray + 3));
    const v3 N = v3p(h->N);
    const float NV = vdot(N, V);
    const v3 Y = NV > 0.f ? N : vneg(N);
    float ein = 1, eout = m->ior;
    if (!h->front) { ein = m->ior; eout = 1; }
    float eta = ein / eout;
    const v3 Z = vcross(V, Y);
    v3 X = vnorm(vcross(Y, Z));
    float cosI = NV;
    float sinI = sqrtf(fmaxf(0.0f, 1 - cosI * cosI));
    float sinO = RMAX(0.f, RMIN(1.f, sinI * eta));
    float cosO = sqrtf(1.f - sinO * sinO);
    v3 tDir = vsub(vscale(vneg(X), sinO), vscale(Y, cosO));
    v3 rDir = vsub(vscale(vscale(N, 2.f), NV), V);
    const float C0 = (eta - 1.f) * (eta - 1.f) / ((eta + 1.f) * (eta + 1.f));
    float rC = C0 + (1.f - C0) * powf(1.f - fabsf(cosI), 5.f);
    const float tC = 1.f - rC;
    const int tot = (eta * sinI) > 1.001f;
    const v3 tK = v3p(m->refraction), rK = v3p(m->reflection);
    const v3 sRefr = tot ? V3(0, 0, 0) : vscale(tK, tC);
    const v3 sRefl = tot ? vadd(rK, tK) : vadd(rK, vscale(tK, rC));
    const v3 sDiff = v3p(m->diffuse), sSpec = v3p(m->specular);
    float random = philox_next(rng);
    float diffuseProb = gray3(sDiff), refractionProb = gray3(sRefr), reflectionProb = gray3(sRefl), absorptionProb = gray3(v3p(m->absorption));
    float total = diffuseProb + reflectionProb + refractionProb + absorptionProb;
    diffuseProb /= total; refractionProb /= total; reflectionProb /= total;
    const float rcp = 1.f / total;
    const float select = random * total;
    const float luma = 0.00001f;
    int selected; float scale = 1.f;
    if (select <= refractionProb && refractionProb > luma) { selected = 0; scale = refractionProb * rcp; }
    else if (select > refractionProb && select <= refractionProb + reflectionProb && reflectionProb > luma) { selected = 1; scale = reflectionProb * rcp; }
    else if (select > refractionProb + reflectionProb && select < refractionProb + reflectionProb + diffuseProb && diffuseProb > luma) { selected = 2; scale = diffuseProb * rcp; }
    else selected = 3;
    v3 dir, BxDF;
    if (selected == 0) {
        if (m->refraction_glossiness > 0.f) {              /* :183-190: SampleHemisphere (:40-48), in ITS frame, used as is */
            float u1 = philox_next(rng), u2 = philox_next(rng);
            const float r = sqrtf(1.0f - u1 * u1);
            const float phi = (float)(2 * M_PI * (double)u2);
            dir = V3(cosf(phi) * r, sinf(phi) * r, u1);
            const v3 L = vnorm(dir);
            const v3 H = vnorm(vadd(V, L));
            const float cosVH = RMAX(0.f, vdot(V, H));
            BxDF = vscale(sRefr, powf(cosVH, m->refraction_glossiness));
        } else { dir = tDir; BxDF = sRefr; }
    } else if (selected == 1) {
        if (m->reflection_glossiness > 0.f) {              /* :200-207: CosineSampleHemisphere (:27-38) */
            float u1 = philox_next(rng), u2 = philox_next(rng);
            const float r = sqrtf(u1);
            const float theta = (float)(2 * M_PI * (double)u2);
            dir = V3(r * cosf(theta), r * sinf(theta), sqrtf(fmaxf(0.0f, 1 - u1)));
            const v3 L = vnorm(dir);
            const v3 H = vnorm(vadd(V, L));
            const float cosNH = RMAX(0.f, vdot(N, H));
            BxDF = vscale(sRefl, powf(cosNH, m->reflection_glossiness));
        } else { dir = rDir; BxDF = sRefl; }
    } else if (selected == 2) {
        if (!h->front) return 0;
        v3 Nt = vdot(N, V3(1, 0, 0)) < 0.4f ? vcross(N, V3(1, 0, 0)) : vcross(N, V3(0, 0, 1));   /* createCoordinateSystem :50-59 */
        Nt = vnorm(Nt);
        const v3 Nb = vcross(N, Nt);
        float theta = (float)((double)philox_next(rng) * M_PI_2);
        float phi = (float)((double)philox_next(rng) * (2.0 * M_PI));
        dir = vadd(vadd(vscale(vscale(Nt, cosf(phi)), sinf(theta)), vscale(vscale(Nb, sinf(phi)), sinf(theta))), vscale(N, cosf(theta)));
        const v3 L = vnorm(dir);
        const v3 H = vnorm(vadd(V, L));
        const float cosNH = RMAX(0.f, vdot(N, H));
        BxDF = vadd(sDiff, vscale(sSpec, powf(cosNH, m->glossiness)));
    } else return 0;
    st3(ray, v3p(h->p)); st3(ray + 3, vnorm(dir));
    *c = vdivs(vmul(*c, BxDF), 1.f * scale);
    if (!h->front) *c = vmul(*c, attenuation(v3p(m->absorption), h->z));
    return 1;
}

/* public faces of the three materials.h helpers, for the tests that compare them with the reference's own
 * (tests/golden/pbounce.npz): RandomPhotonBounce draws from the active script (orc_script_begin) */
int orc_random_photon_bounce(const rt_blinn *m, const orc_hit *h, float ray[6], float c[3])
{
    philox_t rng;
    memset(&rng, 0, sizeof rng);
    rng.used = 4;
    v3 cc = v3p(c);
    const int r = random_photon_bounce(m, h, ray, &cc, &rng);
    st3(c, cc);
    return r;
}
void orc_attenuation(const float absorption[3], float l, float out[3]) { st3(out, attenuation(v3p(absorption), l)); }
/* createCoordinateSystem, FIN/include/materials.h:50-59 */
void orc_coordinate_system(const float N_[3], float Nt_[3], float Nb_[3])
{
    const v3 N = v3p(N_);
    v3 Nt = vdot(N, V3(1, 0, 0)) < 0.4f ? vcross(N, V3(1, 0, 0)) : vcross(N, V3(0, 0, 1));
    Nt = vnorm(Nt);
    st3(Nt_, Nt); st3(Nb_, vcross(N, Nt));
}

/* RenderImage::ComputeZBufferImage, FIN/include/scene.h:591-613 */
void orc_zbuffer_image(const float *zbuffer, int width, int height, uint8_t *zbufferImg)
{
    int size = width * height;
    float zmin = BIGFLOAT, zmax = 0;
    for (int i = 0; i < size; i++) {
        if (zbuffer[i] == BIGFLOAT) continue;
        if (zmin > zbuffer[i]) zmin = zbuffer[i];
        if (zmax < zbuffer[i]) zmax = zbuffer[i];
    }
    for (int i = 0; i < size; i++) {
        if (zbuffer[i] == BIGFLOAT) zbufferImg[i] = 0;
        else {
            float f = (zmax - zbuffer[i]) / (zmax - zmin);
            int c = (int)(f * 255);
            if (c < 0) c = 0;
            if (c > 255) c = 255;
            zbufferImg[i] = (uint8_t)c;
        }
    }
}
/* RenderImage::ComputeSampleCountImage, FIN/include/scene.h:615-637; returns smax */
int orc_sample_count_image(const uint8_t *sampleCount, int width, int height, uint8_t *sampleCountImg)
{
    int size = width * height;
    uint8_t smin = 255, smax = 0;
    for (int i = 0; i < size; i++) {
        if (smin > sampleCount[i]) smin = sampleCount[i];
        if (smax < sampleCount[i]) smax = sampleCount[i];
    }
    if (smax == smin) {
        for (int i = 0; i < size; i++) sampleCountImg[i] = 0;
    } else {
        for (int i = 0; i < size; i++) {
            int c = (255 * (sampleCount[i] - smin)) / (smax - smin);
            if (c < 0) c = 0;
            if (c > 255) c = 255;
            sampleCountImg[i] = (uint8_t)c;
        }
    }
    return smax;
}

/* The caustic loop of generatePhotonMap (P13/main.cpp:383-398) + CausticTracing (:431-457), with FIN's
 * RandomPhotonBounce and the counter RNG of the photon pass: every diffuse hit is counted, only those behind more
 * than one specular hit are stored; out is 1-based, returns the number stored. */
uint32_t orc_caustic_pass(const orc_scene *s, uint32_t seed, uint32_t max_diffuse_hits, int max_bounce,
                          rt_photon *out, uint64_t *attempts_out)
{
    int npl = 0;
    for (int l = 0; l < s->n_lights; l++) if (s->lights[l].type == RT_LIGHT_POINT) npl++;
    uint32_t n = 0;
    uint64_t attempt = 0, counted = 0;
    memset(&out[0], 0, sizeof(rt_photon));
    for (; npl > 0 && counted < max_diffuse_hits; attempt++) {
        philox_t rng;
        rng.key0 = seed; rng.key1 = 0x52544D49u; rng.c0 = (uint32_t)attempt; rng.c1 = (uint32_t)(attempt >> 32); rng.blk = 0; rng.used = 4;
        int pick = (int)(philox_next(&rng) * (float)npl);
        if (pick >= npl) pick = npl - 1;
        const rt_light *L = 0;
        for (int l = 0; l < s->n_lights; l++) if (s->lights[l].type == RT_LIGHT_POINT) { if (pick == 0) { L = &s->lights[l]; break; } pick--; }
        v3 c = v3p(L->intensity);
        const v3 position = v3p(L->position);
        const float x = 2 * philox_next(&rng) - 1, y = 2 * philox_next(&rng) - 1, z = 2 * philox_next(&rng) - 1;
        float ray[6];
        st3(ray, position);
        st3(ray + 3, vnorm(vsub(vadd(V3(x, y, z), position), position)));
        orc_hit h;
        if (!orc_trace(s, RT_SHADE_FIN, ray, &h)) continue;
        const rt_blinn *m = hit_material(s, &h);
        int hitspec = gray3(v3p(m->diffuse)) > 0 ? 0 : 1;                      /* :391-396 */
        int bounce = max_bounce;
        uint32_t stored = 0;
        while (bounce > 0 && random_photon_bounce(m, &h, ray, &c, &rng)) {       /* CausticTracing */
            orc_hit nh;
            if (!orc_trace(s, RT_SHADE_FIN, ray, &nh)) break;
            m = hit_material(s, &nh);
            if (gray3(v3p(m->diffuse)) > 0) {
                if (hitspec > 1 && stored < 8) {
                    float pw[3] = { c.x, c.y, c.z };
                    orc_photon_pack(nh.p, ray + 3, pw, &out[++n]);
                    stored++;
                }
                counted++;                                                      /* savedPhoton++ (:448) */
            } else hitspec++;
            bounce--;
            h = nh;
        }
    }
    if (n > 0) {
        const float scale = (float)(1.0 * 4 * M_PI / n);
        for (uint32_t i = 1; i <= n; i++) out[i].power *= scale;
    }
    if (attempts_out) *attempts_out = attempt;
    return n;
}

/* generatePhotonMap (FIN/main.cpp:350-396) + PhotonTracing (:439-459) + RandomPhoton (:489-497);
 * out is 1-based; returns the number of photons (>= max_photons, overshoot <= 7). */
uint32_t orc_photon_pass(const orc_scene *s, uint32_t seed, uint32_t max_photons, int max_bounce,
                         rt_photon *out, uint64_t *attempts_out)
{
    int npl = 0;
    for (int l = 0; l < s->n_lights; l++) if (s->lights[l].type == RT_LIGHT_POINT) npl++;
    uint32_t n = 0;
    uint64_t attempt = 0;
    memset(&out[0], 0, sizeof(rt_photon));
    for (; npl > 0 && n < max_photons; attempt++) {
        philox_t rng;
        rng.key0 = seed; rng.key1 = 0x52544D49u; rng.c0 = (uint32_t)attempt; rng.c1 = (uint32_t)(attempt >> 32); rng.blk = 0; rng.used = 4;
        int pick = (int)(philox_next(&rng) * (float)npl);
        if (pick >= npl) pick = npl - 1;
        const rt_light *L = 0;
        for (int l = 0; l < s->n_lights; l++) if (s->lights[l].type == RT_LIGHT_POINT) { if (pick == 0) { L = &s->lights[l]; break; } pick--; }
        v3 c = v3p(L->intensity);
        const v3 position = v3p(L->position);
        const float x = 2 * philox_next(&rng) - 1, y = 2 * philox_next(&rng) - 1, z = 2 * philox_next(&rng) - 1;
        float ray[6];
        st3(ray, position);
        st3(ray + 3, vnorm(vsub(vadd(V3(x, y, z), position), position)));     /* Direction(p+position) */
        orc_hit h;
        if (!orc_trace(s, RT_SHADE_FIN, ray, &h)) continue;
        const rt_blinn *m = hit_material(s, &h);
        if (!(gray3(v3p(m->diffuse)) > 0)) continue;                          /* IsPhotonSurface */
        int bounce = max_bounce;
        uint32_t stored = 0;
        while (bounce > 0 && random_photon_bounce(m, &h, ray, &c, &rng)) {
            orc_hit nh;
            if (!orc_trace(s, RT_SHADE_FIN, ray, &nh)) break;
            m = hit_material(s, &nh);
            if (gray3(v3p(m->diffuse)) > 0 && stored < 8) {
                float pw[3] = { c.x, c.y, c.z };
                orc_photon_pack(nh.p, ray + 3, pw, &out[++n]);
                stored++;
            }
            bounce--;
            h = nh;
        }
    }
    const float scale = (float)(1.0 * 4 * M_PI / n);
    for (uint32_t i = 1; i <= n; i++) out[i].power *= scale;
    if (attempts_out) *attempts_out = attempt;
    return n;
}
