// rt_photon_build.hip -- the photon set-up on the GPU ("emit, bounce, pack, kd-balance"), gfx950.
//
//   photon_compact   what generatePhotonMap's while() does with the attempts of a k_photon_trace batch (FIN/main.cpp:
//                    361-395): attempts are consumed in order while fewer than MAX photons are stored; the stored photons of
//                    the consumed attempts are packed into the reference's 24-byte Photon (AddPhoton: SetDirection /
//                    SetPower, cyPhotonMap.h:139-156,184-192), in attempt order; ScalePhotonPowers (:396) afterwards.
//   photon_structure the gather structure k_gather walks (rt_dev.h), from 24-byte photons resident in HBM: recursive median
//                    splits along the widest axis of each segment's tight box down to sub-leaves of <= RT_SUB_PHOTONS
//                    photons -- one stable radix sort of (segment id, coordinate) keys per level, all segments of a level
//                    at once --, tight boxes of every node of the tree, decoded 32-byte slots, density grid.
//
// k_gather returns the exact k nearest photons whatever the partition, so this build only has to be a GOOD partition
// (balanced, tight boxes), not the reference's left-balanced heap: the heap order is only needed to know which photons
// LocatePhotons can never reach (cyPhotonMap.h:217,371) and is found on the host for those few (rt_api.cpp).
// Sort and scan are rocPRIM's (hipcub front end); everything else is written here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "rt_dev.h"

#define PB_BLOCK 256

// ---- photon_compact ---------------------------------------------------------------------------------------------
struct CompactState { unsigned long long attempts, counted; uint32_t stored, pad; };

__global__ __launch_bounds__(PB_BLOCK) void k_split_counts(const uint32_t *count, uint32_t n, int mode, uint32_t *stored, uint32_t *counted)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = count[i];
    stored[i] = c & 0xFFFFu;
    counted[i] = mode == 0 ? (c & 0xFFFFu) : (c >> 16);
}

// Photon::SetDirection / SetPower via AddPhoton (FIN/include/cyPhotonMap.h:139-156,184-192)
__device__ __forceinline__ void pack_photon(const float *rec, rt_photon &o)
{
    o.position[0] = rec[0]; o.position[1] = rec[1]; o.position[2] = rec[2];
    o.dir_x = (int16_t)(rec[3] * 0x7FFF);
    o.dir_y = (int16_t)(rec[4] * 0x7FFF);
    o.plane_and_dirz = rec[5] > 0 ? 0 : 0x8;
    float power = rec[6];
    if (power < rec[7]) power = rec[7];
    if (power < rec[8]) power = rec[8];
    o.power = power;
    for (int c = 0; c < 3; c++) {
        const float s = (rec[6 + c] / power) * 255;
        const int v = (s == s) ? (int)s : 0;
        o.color[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// one thread per attempt of the batch; ex_* = exclusive prefix sums over the batch
__global__ __launch_bounds__(PB_BLOCK) void k_compact(const float *recs, const uint32_t *stored, const uint32_t *counted, const uint32_t *ex_stored,
                                                      const uint32_t *ex_counted, uint32_t n_attempts, unsigned long long max_count,
                                                      CompactState *st, rt_photon *out, uint32_t out_cap)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_attempts) return;
    const unsigned long long before = st->counted;            // state of the previous batches (read-only in this launch)
    const uint32_t n0 = st->stored;
    // the reference checks the count between attempts (:361): attempt a runs iff fewer than max were counted before it
    const bool consumed = before + ex_counted[a] < max_count;
    if (!consumed) return;
    const uint32_t ns = stored[a];
    for (uint32_t j = 0; j < ns; j++) {
        const uint32_t at = n0 + ex_stored[a] + j + 1u;       // 1-based
        if (at < out_cap) pack_photon(recs + ((size_t)a * 8 + j) * 9, out[at]);
    }
    const bool last = a + 1 == n_attempts || !(before + ex_counted[a + 1] < max_count);
    if (last) {
        // the only writer; the other threads of this launch read the fields above -- written to the shadow slot st[1]
        st[1].attempts = st->attempts + a + 1;
        st[1].counted = before + ex_counted[a] + counted[a];
        st[1].stored = n0 + ex_stored[a] + ns;
    }
}

__global__ void k_commit_state(CompactState *st) { st[0] = st[1]; }

__global__ __launch_bounds__(PB_BLOCK) void k_scale_powers(rt_photon *ph, uint32_t n, float scale)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ph[i + 1].power *= scale;
}

// scratch = 4 arrays of n_attempts uint32 + the scan's temporary storage (rtk_photon_compact_scratch bytes)
size_t rtk_photon_compact_scratch(uint32_t n_attempts)
{
    size_t tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_attempts);
    return (size_t)n_attempts * 16 + ((tmp + 255) & ~(size_t)255) + 256;
}
void rtk_photon_compact(hipStream_t st, const float *recs, const uint32_t *count, uint32_t n_attempts, int mode, unsigned long long max_count,
                        void *state_dev, rt_photon *out, uint32_t out_cap, void *scratch, size_t scratch_bytes)
{
    uint32_t *stored = (uint32_t *)scratch, *counted = stored + n_attempts, *ex_s = counted + n_attempts, *ex_c = ex_s + n_attempts;
    void *tmp = (char *)scratch + (size_t)n_attempts * 16;
    size_t tmp_bytes = scratch_bytes - (size_t)n_attempts * 16;
    const int grid = (int)((n_attempts + PB_BLOCK - 1) / PB_BLOCK);
    hipLaunchKernelGGL(k_split_counts, dim3(grid), dim3(PB_BLOCK), 0, st, count, n_attempts, mode, stored, counted);
    (void)hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, stored, ex_s, (int)n_attempts, st);
    (void)hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counted, ex_c, (int)n_attempts, st);
    CompactState *S = (CompactState *)state_dev;
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(PB_BLOCK), 0, st, recs, stored, counted, ex_s, ex_c, n_attempts, max_count, S, out, out_cap);
    hipLaunchKernelGGL(k_commit_state, dim3(1), dim3(1), 0, st, S);
}
void rtk_photon_scale(hipStream_t st, rt_photon *ph, uint32_t n, float scale)
{
    if (n) hipLaunchKernelGGL(k_scale_powers, dim3((n + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, ph, n, scale);
}

// out[i] = in[i + (number of skipped indices <= ...)]: the photons LocatePhotons can reach (at most 8 indices are skipped)
struct SkipList { uint32_t n; uint32_t idx[8]; };      // ascending, 0-based positions in `in`
__global__ __launch_bounds__(PB_BLOCK) void k_copy_skipping(const rt_photon *in, uint32_t n_in, SkipList skip, rt_photon *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    uint32_t before = 0;
    for (uint32_t k = 0; k < skip.n; k++) { if (skip.idx[k] == i) return; if (skip.idx[k] < i) before++; }
    out[i - before] = in[i];
}
void rtk_photon_copy_skipping(hipStream_t st, const rt_photon *in, uint32_t n_in, const uint32_t *skip, uint32_t n_skip, rt_photon *out)
{
    SkipList S; S.n = n_skip > 8 ? 8 : n_skip;
    for (uint32_t k = 0; k < S.n; k++) S.idx[k] = skip[k];
    if (n_in) hipLaunchKernelGGL(k_copy_skipping, dim3((n_in + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, in, n_in, S, out);
}

// ---- photon_unreachable ---------------------------------------------------------------------------------------------
// Which photons of an UNBALANCED array PhotonMap::BalanceSegment (FIN/include/cyPhotonMap.h:222-284) would put into the heap slots
// LocatePhotons never visits (slots [first, last] = the last three or four: it descends only while index < halfStoredPhotons,
// :217,:371).  The content of a segment when BalanceSegment partitions it is a SET that depends only on the partitions of its
// ancestors -- the `median - start` smallest keys go left, the larger ones right -- as long as the key of the median is unique
// in its segment (with equal keys the reference's swap sequence decides; that case is reported and left to the host's exact
// replay, rt::UnreachablePhotons).  So only the root paths of those slots are followed: per level and live segment a radix
// select (three histogram passes over 11 + 11 + 10 key bits) of the median's key among the photons labelled with the segment,
// then one pass that relabels them with the child segments.  Segment sizes, median ranks and which children matter are a
// closed form of n and come precomputed from the host; boxes, split axes and split values are found here.
#define UR_MAX_LEVELS 40
#define UR_MAX_SEGS 4
#define UR_BINS 2048
struct UrSeg {
    uint32_t index, count, m;            // heap slot of the segment's median, photons in it, 0-based rank of the median key (host)
    int32_t left_next, right_next;       // position of the child segment in the next level's list, -1: not followed (host)
    uint32_t left_single, right_single, median_target;   // a one-photon child / the median itself lands in a wanted slot (host)
    float box[6];                        // bmin, bmax handed down by the parent (device)
};
struct UrState { uint32_t n_out, tie; uint32_t out[14]; };
__device__ __forceinline__ uint32_t ur_key(float f) { if (f == 0.0f) f = 0.0f; return ((__float_as_uint(f) & 0x80000000u) ? ~__float_as_uint(f) : (__float_as_uint(f) | 0x80000000u)); }
__device__ __forceinline__ int ur_axis(const float *box)      // BalanceSegment's choice, :234-240
{
    const float dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
    int axis = 2;
    if (dx > dy) { if (dx > dz) axis = 0; }
    else if (dy > dz) axis = 1;
    return axis;
}
// the bin of `hist` (nbins counters) that holds 0-based rank r, and r's rank inside that bin; whole block, nbins <= 8 * blockDim
__device__ uint32_t ur_pick(const uint32_t *hist, uint32_t nbins, uint32_t &r, uint32_t *s_scan)
{
    const uint32_t t = threadIdx.x, per = (nbins + blockDim.x - 1) / blockDim.x;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t b = t * per + k; if (b < nbins) mine += hist[b]; }
    s_scan[t] = mine;
    __syncthreads();
    __shared__ uint32_t s_bin, s_rank;
    if (t == 0) {
        uint32_t cum = 0, owner = 0;
        for (uint32_t i = 0; i < blockDim.x; i++) { if (cum + s_scan[i] > r) { owner = i; break; } cum += s_scan[i]; owner = i; }
        uint32_t b = owner * per;
        for (; b < nbins && b < (owner + 1) * per; b++) { const uint32_t h = hist[b]; if (cum + h > r) break; cum += h; }
        s_bin = b < nbins ? b : nbins - 1; s_rank = r - cum;
    }
    __syncthreads();
    r = s_rank;
    const uint32_t bin = s_bin;
    __syncthreads();
    return bin;
}
// prefix (the key bits fixed so far) and remaining rank after `passes` finished histogram passes of this segment
__device__ void ur_prefix(const uint32_t *hists, const UrSeg &S, int passes, uint32_t &prefix, uint32_t &rank, uint32_t *s_scan)
{
    prefix = 0; rank = S.m;
    if (passes >= 1) prefix = ur_pick(hists, UR_BINS, rank, s_scan);
    if (passes >= 2) prefix = (prefix << 11) | ur_pick(hists + UR_BINS, UR_BINS, rank, s_scan);
    if (passes >= 3) prefix = (prefix << 10) | ur_pick(hists + 2 * UR_BINS, 1024, rank, s_scan);
}
__global__ __launch_bounds__(PB_BLOCK) void k_ur_bbox(const rt_photon *ph0, uint32_t n, uint32_t *ord)
{
    // the reference's box loop covers photons[0], the unused all-zero slot, too (cyPhotonMap.h:201-210)
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; a++) { const uint32_t k = ur_key(ph0[i].position[a]); lo[a] = min(lo[a], k); hi[a] = max(hi[a], k); }
    __shared__ uint32_t s_lo[3], s_hi[3];
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0xFFFFFFFFu; s_hi[threadIdx.x] = 0u; }
    __syncthreads();
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) { lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], off)); hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_lo[a], lo[a]); atomicMax(&s_hi[a], hi[a]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&ord[threadIdx.x], s_lo[threadIdx.x]); atomicMax(&ord[3 + threadIdx.x], s_hi[threadIdx.x]); }
}
// (position, raw index) records of photons [0, n]: what the host's replay of BalanceSegment works on (16 bytes instead of the 24-byte photon)
__global__ __launch_bounds__(PB_BLOCK) void k_ur_pack(const rt_photon *ph0, uint32_t n, float4 *recs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) recs[i] = make_float4(ph0[i].position[0], ph0[i].position[1], ph0[i].position[2], __uint_as_float(i));
}
void rtk_photon_pack_positions(hipStream_t st, const rt_photon *ph0, uint32_t n, void *recs16)
{
    hipLaunchKernelGGL(k_ur_pack, dim3((n + 1 + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, ph0, n, (float4 *)recs16);
}
__global__ void k_ur_root(const uint32_t *ord, UrSeg *root)
{
    for (int a = 0; a < 6; a++) { const uint32_t u = ord[a]; root->box[a] = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
}
// one histogram pass (pass 0, 1, 2) of every live segment of a level: grid = (blocks, segments)
__global__ __launch_bounds__(PB_BLOCK) void k_ur_hist(const rt_photon *ph0, const uint32_t *label, uint32_t n, const UrSeg *segs, uint32_t *hists, int pass)
{
    __shared__ uint32_t s_hist[UR_BINS];
    __shared__ uint32_t s_scan[PB_BLOCK];
    const UrSeg S = segs[blockIdx.y];
    uint32_t *H = hists + (size_t)blockIdx.y * 3 * UR_BINS;
    uint32_t prefix, rank;
    ur_prefix(H, S, pass, prefix, rank, s_scan);
    for (uint32_t b = threadIdx.x; b < UR_BINS; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    const int axis = ur_axis(S.box);
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t mask = pass == 2 ? 1023u : 2047u;
    for (uint32_t i = 1 + blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        if (label[i] != S.index) continue;
        const uint32_t k = ur_key(ph0[i].position[axis]);
        if (pass > 0 && (k >> (shift + (pass == 1 ? 11 : 10))) != prefix) continue;
        atomicAdd(&s_hist[(k >> shift) & mask], 1u);
    }
    __syncthreads();
    uint32_t *G = H + (size_t)pass * UR_BINS;
    for (uint32_t b = threadIdx.x; b < UR_BINS; b += blockDim.x) if (s_hist[b]) atomicAdd(&G[b], s_hist[b]);
}
// relabel the photons of every live segment with its children; hand the children their boxes; record wanted photons
__global__ __launch_bounds__(PB_BLOCK) void k_ur_apply(const rt_photon *ph0, uint32_t *label, uint32_t n, const UrSeg *segs, UrSeg *next, const uint32_t *hists,
                                                       UrState *state)
{
    __shared__ uint32_t s_scan[PB_BLOCK];
    const UrSeg S = segs[blockIdx.y];
    const uint32_t *H = hists + (size_t)blockIdx.y * 3 * UR_BINS;
    uint32_t vkey, rank;
    ur_prefix(H, S, 3, vkey, rank, s_scan);
    const bool unique = H[2 * UR_BINS + (vkey & 1023u)] == 1u;          // the median's key occurs once in the segment
    const int axis = ur_axis(S.box);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!unique) atomicExch(&state->tie, 1u);
        const float v = __uint_as_float((vkey & 0x80000000u) ? (vkey & 0x7FFFFFFFu) : ~vkey);
        if (S.left_next >= 0) { UrSeg &L = next[S.left_next]; for (int a = 0; a < 6; a++) L.box[a] = S.box[a]; L.box[3 + axis] = v; }     // tmax[axis] = split (:262)
        if (S.right_next >= 0) { UrSeg &R = next[S.right_next]; for (int a = 0; a < 6; a++) R.box[a] = S.box[a]; R.box[axis] = v; }        // tmin[axis] = split (:271)
    }
    for (uint32_t i = 1 + blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        if (label[i] != S.index) continue;
        const uint32_t k = ur_key(ph0[i].position[axis]);
        bool want = false;
        uint32_t nl = 0;
        if (k < vkey) { if (S.left_next >= 0) nl = 2u * S.index; want = S.left_single != 0; }
        else if (k > vkey) { if (S.right_next >= 0) nl = 2u * S.index + 1u; want = S.right_single != 0; }
        else want = S.median_target != 0;
        label[i] = nl;
        if (want) { const uint32_t at = atomicAdd(&state->n_out, 1u); if (at < 14u) state->out[at] = i; }
    }
}
__global__ __launch_bounds__(PB_BLOCK) void k_ur_fill(uint32_t *p, uint32_t n, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

size_t rtk_photon_unreachable_scratch(uint32_t n)
{
    return (((size_t)n + 1) * 4 + 255 & ~(size_t)255) + (size_t)UR_MAX_LEVELS * UR_MAX_SEGS * (sizeof(UrSeg) + 3 * UR_BINS * 4) + 1024;
}
// ph0: the photons as generated, 1-based ([0] all zero), on the device.  Heap slots [first, last] are wanted.  result (host):
// [0] = number of photons found, [1] = 1 when a median's key was not unique (the caller must use the host's exact replay),
// [2..] = their 1-based raw indices (unsorted).  Returns after the stream has been synchronised.
hipError_t rtk_photon_unreachable(hipStream_t st, const rt_photon *ph0, uint32_t n, uint32_t first, uint32_t last, void *scratch, size_t scratch_bytes,
                                  uint32_t result[16])
{
    memset(result, 0, 16 * sizeof(uint32_t));
    if (n == 0 || first > last || first > n) return hipSuccess;
    if (scratch_bytes < rtk_photon_unreachable_scratch(n)) return hipErrorInvalidValue;
    // ---- the schedule: segment sizes and median ranks along the root paths of the wanted slots (closed form of n) ----
    auto holds = [&](uint32_t index) { for (uint32_t t = first; t <= last; t++) { uint32_t a = t; while (a > index) a >>= 1; if (a == index) return true; } return false; };
    auto wanted = [&](uint32_t slot) { return slot >= first && slot <= last; };
    std::vector<std::vector<UrSeg>> sched(1);
    { UrSeg r; memset(&r, 0, sizeof r); r.index = 1; r.count = n; r.left_next = r.right_next = -1; sched[0].push_back(r); }
    for (size_t L = 0; L < sched.size(); L++) {
        std::vector<UrSeg> nxt;
        for (UrSeg &S : sched[L]) {
            const uint32_t start = 1, end = S.count;          // ranks are relative to the segment
            if (S.count == 1) { S.m = 0; S.median_target = wanted(S.index); continue; }      // (never scheduled: a one-photon child is a *_single)
            uint32_t median = 1;
            while (4 * median <= end - start + 1) median += median;
            if (3 * median <= end - start + 1) { median += median; median += start - 1; }
            else median = end - median + 1;
            S.m = median - start;
            S.median_target = wanted(S.index);
            const uint32_t nleft = median - start, nright = end - median;
            if (nleft == 1) S.left_single = wanted(2 * S.index);
            else if (nleft >= 2 && holds(2 * S.index)) { UrSeg c; memset(&c, 0, sizeof c); c.index = 2 * S.index; c.count = nleft; c.left_next = c.right_next = -1; S.left_next = (int32_t)nxt.size(); nxt.push_back(c); }
            if (nright == 1) S.right_single = wanted(2 * S.index + 1);
            else if (nright >= 2 && holds(2 * S.index + 1)) { UrSeg c; memset(&c, 0, sizeof c); c.index = 2 * S.index + 1; c.count = nright; c.left_next = c.right_next = -1; S.right_next = (int32_t)nxt.size(); nxt.push_back(c); }
        }
        if (!nxt.empty()) sched.push_back(nxt);
        if (sched.size() > UR_MAX_LEVELS || nxt.size() > UR_MAX_SEGS) return hipErrorInvalidValue;
    }
    // ---- device buffers inside `scratch` ----
    char *base = (char *)scratch;
    uint32_t *label = (uint32_t *)base;
    base += (((size_t)n + 1) * 4 + 255) & ~(size_t)255;
    UrSeg *d_sched = (UrSeg *)base;
    base += (size_t)UR_MAX_LEVELS * UR_MAX_SEGS * sizeof(UrSeg);
    uint32_t *d_hists = (uint32_t *)base;
    base += (size_t)UR_MAX_LEVELS * UR_MAX_SEGS * 3 * UR_BINS * 4;
    UrState *d_state = (UrState *)base;
    uint32_t *d_ord = (uint32_t *)(base + 256);
    std::vector<UrSeg> flat((size_t)UR_MAX_LEVELS * UR_MAX_SEGS);
    memset(flat.data(), 0, flat.size() * sizeof(UrSeg));
    for (size_t L = 0; L < sched.size(); L++) for (size_t k = 0; k < sched[L].size(); k++) flat[L * UR_MAX_SEGS + k] = sched[L][k];
    hipError_t e;
    if ((e = hipMemcpyAsync(d_sched, flat.data(), flat.size() * sizeof(UrSeg), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_hists, 0, sched.size() * UR_MAX_SEGS * 3 * UR_BINS * 4, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_state, 0, sizeof(UrState), st)) != hipSuccess) return e;
    const uint32_t ord_init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    if ((e = hipMemcpyAsync(d_ord, ord_init, sizeof ord_init, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    const int grid = (int)std::min<size_t>(((size_t)n + PB_BLOCK) / PB_BLOCK, 1024);
    hipLaunchKernelGGL(k_ur_fill, dim3((n + 1 + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, label, n + 1, 1u);
    hipLaunchKernelGGL(k_ur_bbox, dim3(grid), dim3(PB_BLOCK), 0, st, ph0, n, d_ord);
    hipLaunchKernelGGL(k_ur_root, dim3(1), dim3(1), 0, st, d_ord, d_sched);
    for (size_t L = 0; L < sched.size(); L++) {
        const UrSeg *segs = d_sched + L * UR_MAX_SEGS;
        uint32_t *hists = d_hists + L * UR_MAX_SEGS * 3 * UR_BINS;
        const dim3 g(grid, (unsigned)sched[L].size());
        for (int pass = 0; pass < 3; pass++) hipLaunchKernelGGL(k_ur_hist, g, dim3(PB_BLOCK), 0, st, ph0, label, n, segs, hists, pass);
        hipLaunchKernelGGL(k_ur_apply, g, dim3(PB_BLOCK), 0, st, ph0, label, n, segs, d_sched + (L + 1) * UR_MAX_SEGS, hists, d_state);
        if (L < 6 || (L & 3) == 3) {
            // a median that is not unique ends the search: found out early (maps whose photons sit on axis-aligned walls tie within
            // the first levels -- the wanted slots lie on the path of the LARGEST keys, where a wall's photons share one coordinate)
            uint32_t tie = 0;
            if ((e = hipMemcpyAsync(&tie, &d_state->tie, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            if (tie) { result[1] = 1; return hipSuccess; }
        }
    }
    UrState h;
    if ((e = hipMemcpyAsync(&h, d_state, sizeof h, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    result[0] = h.n_out; result[1] = h.tie;
    for (uint32_t k = 0; k < 14 && k < h.n_out; k++) result[2 + k] = h.out[k];
    return hipSuccess;
}

// ---- photon_structure -----------------------------------------------------------------------------------------------
// order-preserving map float -> uint32 (for atomicMin / atomicMax and for radix keys)
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
// (written without a select: `(u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u` feeding a float subtraction crashes this compiler's instruction selection)
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float(u ^ ((uint32_t)((int32_t)(~u) >> 31) | 0x80000000u)); }

// Segment j of level L (0 <= j < 2^L) of the recursion  [lo, hi) -> [lo, mid), [mid, hi)  with mid = lo + (hi - lo + 1) / 2:
// which one holds sorted position i, and where it starts / ends
__device__ __forceinline__ uint32_t segment_of(uint32_t i, uint32_t n, int L, uint32_t &lo, uint32_t &hi)
{
    lo = 0; hi = n;
    uint32_t j = 0;
    for (int l = 0; l < L; l++) {
        const uint32_t mid = lo + (hi - lo + 1u) / 2u;
        if (i < mid) { hi = mid; j = 2u * j; } else { lo = mid; j = 2u * j + 1u; }
    }
    return j;
}

__global__ __launch_bounds__(PB_BLOCK) void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(PB_BLOCK) void k_fill_boxes(uint32_t *boxu, size_t n_nodes)
{
    const uint32_t vmin = f2ord(3.0e38f), vmax = f2ord(-3.0e38f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * 6; i += (size_t)gridDim.x * blockDim.x) boxu[i] = (i % 6) < 3 ? vmin : vmax;
}
__global__ __launch_bounds__(PB_BLOCK) void k_iota(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// boxes of the nodes of level L (heap index 2^L + j), as ordered uints: boxu[node][0..2] = min, [3..5] = max.
// Top levels (segments of thousands of photons): a workgroup reduces a contiguous chunk of the sorted order -- a thread keeps
// a running box while the segment id stays the same, waves and then the workgroup combine through shuffles / LDS when they
// sit in one segment (the rule) -- and only then touches the node's words with atomics: a few thousand per level instead
// of one per wave (100 000 same-word atomics on the root's box took 0.36 ms per level).
#define PB_TOP_BLOCKS 1024
// the value of the first lane whose v differs from `none` (or `none` when there is no such lane), in every lane
__device__ __forceinline__ uint32_t __reduce_first_valid(uint32_t v, uint32_t none)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(v != none);
    if (m == 0) return none;
    return (uint32_t)__shfl((int)v, __ffsll((long long)m) - 1);
}
__device__ __forceinline__ void box_atomics(uint32_t *boxu, int L, uint32_t j, const uint32_t mn[3], const uint32_t mx[3])
{
    uint32_t *b = boxu + 6 * (size_t)((1u << L) + j);
    for (int a = 0; a < 3; a++) { atomicMin(b + a, mn[a]); atomicMax(b + 3 + a, mx[a]); }
}
__global__ __launch_bounds__(PB_BLOCK) void k_level_boxes(const rt_photon *ph, const uint32_t *perm, uint32_t n, int L, uint32_t *boxu)
{
    __shared__ uint32_t s_box[PB_BLOCK / 64][6];
    __shared__ uint32_t s_seg[PB_BLOCK / 64];
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t c0 = blockIdx.x * per, c1 = min(n, c0 + per);
    const uint32_t EMPTY = 0xFFFFFFFFu;
    uint32_t cur = EMPTY, mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = f2ord(3.0e38f); mx[a] = f2ord(-3.0e38f); }
    for (uint32_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
        uint32_t lo, hi;
        const uint32_t j = segment_of(i, n, L, lo, hi);
        if (j != cur) {
            if (cur != EMPTY) box_atomics(boxu, L, cur, mn, mx);
            cur = j;
            for (int a = 0; a < 3; a++) { mn[a] = f2ord(3.0e38f); mx[a] = f2ord(-3.0e38f); }
        }
        const float *p = ph[perm[i]].position;
        for (int a = 0; a < 3; a++) { const uint32_t v = f2ord(p[a]); mn[a] = min(mn[a], v); mx[a] = max(mx[a], v); }
    }
    // wave: one segment in every lane that has one -> shuffles, else every lane for itself
    const uint32_t any = (uint32_t)__builtin_amdgcn_readfirstlane((int)__reduce_first_valid(cur, EMPTY));
    const bool wave_uniform = __all(cur == any || cur == EMPTY);
    const int w = threadIdx.x >> 6;
    if (wave_uniform) {
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off)); mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off)); }
        if ((threadIdx.x & 63) == 0) { s_seg[w] = any; for (int a = 0; a < 3; a++) { s_box[w][a] = mn[a]; s_box[w][3 + a] = mx[a]; } }
    } else {
        if (cur != EMPTY) box_atomics(boxu, L, cur, mn, mx);
        if ((threadIdx.x & 63) == 0) s_seg[w] = EMPTY;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the workgroup: waves of one segment are merged before the atomics
        for (int k = 0; k < PB_BLOCK / 64; k++) {
            if (s_seg[k] == EMPTY) continue;
            uint32_t bmn[3] = {s_box[k][0], s_box[k][1], s_box[k][2]}, bmx[3] = {s_box[k][3], s_box[k][4], s_box[k][5]};
            for (int m = k + 1; m < PB_BLOCK / 64; m++)
                if (s_seg[m] == s_seg[k]) { for (int a = 0; a < 3; a++) { bmn[a] = min(bmn[a], s_box[m][a]); bmx[a] = max(bmx[a], s_box[m][3 + a]); } s_seg[m] = EMPTY; }
            box_atomics(boxu, L, s_seg[k], bmn, bmx);
        }
    }
}

// The same for the levels whose segments are no longer than a few thousand photons: ONE WAVE per segment walks it and writes
// the box -- no atomics (at the last level 65 536 segments of <= 16 photons: six million same-word atomics took 0.4 ms per level)
__device__ __forceinline__ void segment_bounds(uint32_t j, uint32_t n, int L, uint32_t &lo, uint32_t &hi)
{
    lo = 0; hi = n;
    for (int l = L - 1; l >= 0; l--) {
        const uint32_t mid = lo + (hi - lo + 1u) / 2u;
        if ((j >> l) & 1u) lo = mid; else hi = mid;
    }
}
__global__ __launch_bounds__(PB_BLOCK) void k_level_boxes_wave(const rt_photon *ph, const uint32_t *perm, uint32_t n, int L, uint32_t *boxu)
{
    const uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (j >= (1u << L)) return;
    uint32_t lo, hi;
    segment_bounds(j, n, L, lo, hi);
    uint32_t mn[3] = {f2ord(3.0e38f), f2ord(3.0e38f), f2ord(3.0e38f)}, mx[3] = {f2ord(-3.0e38f), f2ord(-3.0e38f), f2ord(-3.0e38f)};
    for (uint32_t i = lo + lane; i < hi; i += 64u) {
        const float *p = ph[perm[i]].position;
        for (int a = 0; a < 3; a++) { const uint32_t v = f2ord(p[a]); mn[a] = min(mn[a], v); mx[a] = max(mx[a], v); }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off)); mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off)); }
    if (lane == 0) {
        uint32_t *b = boxu + 6 * (size_t)((1u << L) + j);
        for (int a = 0; a < 3; a++) { b[a] = mn[a]; b[3 + a] = mx[a]; }
    }
}

// key of sorted position i for the split of level L: (segment id, coordinate along the segment's widest axis)
__global__ __launch_bounds__(PB_BLOCK) void k_level_keys(const rt_photon *ph, const uint32_t *perm, uint32_t n, int L, const uint32_t *boxu,
                                                         unsigned long long *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo, hi;
    const uint32_t j = segment_of(i, n, L, lo, hi);
    const uint32_t *b = boxu + 6 * (size_t)((1u << L) + j);
    const float ex = ord2f(b[3]) - ord2f(b[0]), ey = ord2f(b[4]) - ord2f(b[1]), ez = ord2f(b[5]) - ord2f(b[2]);
    int axis = 0;
    if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
    const float *pp = ph[perm[i]].position;
    const float px = pp[0], py = pp[1], pz = pp[2];
    const float c = axis == 0 ? px : (axis == 1 ? py : pz);     // (a dynamically indexed member array crashes this compiler's instruction selection)
    ((uint2 *)keys)[i] = make_uint2(f2ord(c), j);              // little-endian: low word = coordinate, high word = segment id
}

// Photon::GetDirection (FIN/include/cyPhotonMap.h:158-180), including the reference's `dirX*dirX + dirY-dirY` (:162): z from x alone
__device__ __forceinline__ void photon_direction(const rt_photon &p, float d[3])
{
    const int dirX = p.dir_x, dirY = p.dir_y;
    d[0] = (float)dirX / (float)0x7FFF;
    d[1] = (float)dirY / (float)0x7FFF;
    int dirXY2 = dirX * dirX + dirY - dirY;
    if (dirXY2 > 0x3FFF0001) dirXY2 = 0x3FFF0001;
    const int dirZ2 = 0x3FFF0001 - dirXY2;
    int dirZ = 0, place = 0x40000000, remainder = dirZ2;
    while (place > remainder) place >>= 2;
    while (place) {
        if (remainder >= dirZ + place) { remainder -= dirZ + place; dirZ += place << 1; }
        dirZ >>= 1;
        place >>= 2;
    }
    d[2] = (float)dirZ / (float)0x7FFF;
    if (p.plane_and_dirz & 0x8) d[2] = -d[2];
}

// sorted position i -> slot (i - lo) of sub-leaf j: the 32-byte record k_gather reads (pa = position, dir.x; pb = dir.yz,
// GetMaxPower, colour bytes: the device forms byte / 255.0f * power itself, GetPower :58)
__global__ __launch_bounds__(PB_BLOCK) void k_scatter_slots(const rt_photon *ph, const uint32_t *perm, uint32_t n, int D, float4 *pa, float4 *pb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo, hi;
    const uint32_t j = segment_of(i, n, D, lo, hi);
    const rt_photon p = ph[perm[i]];
    float d[3];
    photon_direction(p, d);
    const size_t at = (size_t)j * RT_SUB_PHOTONS + (i - lo);
    pa[at] = make_float4(p.position[0], p.position[1], p.position[2], d[0]);
    const uint32_t cb = (uint32_t)p.color[0] | ((uint32_t)p.color[1] << 8) | ((uint32_t)p.color[2] << 16);
    pb[at] = make_float4(d[1], d[2], p.power, __uint_as_float(cb));
}

__global__ __launch_bounds__(PB_BLOCK) void k_fill_slots(float4 *pa, float4 *pb, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        pa[i] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
        pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// boxu -> the device form of a box: two aligned 16-byte words (lo.xyz, 0), (hi.xyz, 0); node 0 is unused
__global__ __launch_bounds__(PB_BLOCK) void k_boxes_to_float(const uint32_t *boxu, uint32_t n_nodes, float4 *box4)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t *b = boxu + 6 * (size_t)i;
    box4[2 * (size_t)i] = make_float4(ord2f(b[0]), ord2f(b[1]), ord2f(b[2]), 0.f);
    box4[2 * (size_t)i + 1] = make_float4(ord2f(b[3]), ord2f(b[4]), ord2f(b[5]), 0.f);
}

__global__ __launch_bounds__(PB_BLOCK) void k_density_grid(const rt_photon *ph, uint32_t n, float gx, float gy, float gz, float cell, int dx, int dy, int dz, uint32_t *grid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = ph[i].position;
    const int x = min(dx - 1, max(0, (int)((p[0] - gx) / cell))), y = min(dy - 1, max(0, (int)((p[1] - gy) / cell))), z = min(dz - 1, max(0, (int)((p[2] - gz) / cell)));
    atomicAdd(&grid[((size_t)z * dy + y) * dx + x], 1u);
}

// Scratch of a build over n photons with n_sub sub-leaves: perm x2, keys x2, boxu, sort temp.
size_t rtk_photon_structure_scratch(uint32_t n, uint32_t n_sub)
{
    size_t tmp = 0;
    hipcub::DoubleBuffer<unsigned long long> k(nullptr, nullptr);
    hipcub::DoubleBuffer<uint32_t> v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k, v, (int)n, 0, 64);
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    return np * (2 * 4 + 2 * 8) + (size_t)2 * n_sub * 24 + ((tmp + 255) & ~(size_t)255) + 1024;
}

// Builds pa / pb ((n_sub + 1) * RT_SUB_PHOTONS slots each), box4 (2 * 2 * n_sub float4: heap node i at [2i, 2i+1]; the
// tree over the leaves is its head, the sub-leaf boxes the nodes [n_sub, 2 n_sub)) and the density grid (64^3 counters
// provided; dims / origin / cell come back in grid_out after a stream synchronisation inside this call).
struct PhotonGridOut { float min[3]; float cell; int dim[3]; };
hipError_t rtk_photon_structure(hipStream_t st, const rt_photon *ph, uint32_t n, uint32_t n_sub, float4 *pa, float4 *pb, float4 *box4,
                                uint32_t *grid, PhotonGridOut *grid_out, void *scratch, size_t scratch_bytes)
{
    int D = 0;
    while ((1u << D) < n_sub) D++;
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    char *sp = (char *)scratch;
    uint32_t *perm0 = (uint32_t *)sp; sp += np * 4;
    uint32_t *perm1 = (uint32_t *)sp; sp += np * 4;
    unsigned long long *keys0 = (unsigned long long *)sp; sp += np * 8;
    unsigned long long *keys1 = (unsigned long long *)sp; sp += np * 8;
    uint32_t *boxu = (uint32_t *)sp; sp += (size_t)2 * n_sub * 24;
    void *tmp = sp;
    size_t tmp_bytes = scratch_bytes - (size_t)(sp - (char *)scratch);
    const int grid_n = (int)((n + PB_BLOCK - 1) / PB_BLOCK);
    // empty boxes: min = +3e38, max = -3e38 (what the gather's box test treats as "nothing here"; rt_api.cpp's host build did the same)
    hipLaunchKernelGGL(k_fill_boxes, dim3(256), dim3(PB_BLOCK), 0, st, boxu, (size_t)2 * n_sub);
    hipLaunchKernelGGL(k_iota, dim3(grid_n), dim3(PB_BLOCK), 0, st, perm0, n);
    hipLaunchKernelGGL(k_fill_slots, dim3(512), dim3(PB_BLOCK), 0, st, pa, pb, ((size_t)n_sub + 1) * RT_SUB_PHOTONS);
    hipcub::DoubleBuffer<unsigned long long> kb(keys0, keys1);
    hipcub::DoubleBuffer<uint32_t> vb(perm0, perm1);
    for (int L = 0; L <= D; L++) {
        if (L <= 8) hipLaunchKernelGGL(k_level_boxes, dim3(grid_n < PB_TOP_BLOCKS ? grid_n : PB_TOP_BLOCKS), dim3(PB_BLOCK), 0, st, ph, vb.Current(), n, L, boxu);
        else hipLaunchKernelGGL(k_level_boxes_wave, dim3(((1u << L) * 64u + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, ph, vb.Current(), n, L, boxu);
        if (L == D) break;
        hipLaunchKernelGGL(k_level_keys, dim3(grid_n), dim3(PB_BLOCK), 0, st, ph, vb.Current(), n, L, boxu, kb.Current());
        // stable: equal coordinates keep their order; only the bits in use are sorted (L bits of segment id above 32 of key)
        const hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kb, vb, (int)n, 0, 32 + L, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_scatter_slots, dim3(grid_n), dim3(PB_BLOCK), 0, st, ph, vb.Current(), n, D, pa, pb);
    hipLaunchKernelGGL(k_boxes_to_float, dim3((2 * n_sub + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, boxu, 2 * n_sub, box4);
    // density grid over the photons' bounding box (the root's box), at most 64 cells along the longest axis
    uint32_t root[6];
    hipError_t e = hipMemcpyAsync(root, boxu + 6, sizeof root, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    float rb[6];
    for (int a = 0; a < 6; a++) { const uint32_t u = root[a]; const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; memcpy(&rb[a], &b, 4); }
    float ext = fmaxf(fmaxf(rb[3] - rb[0], rb[4] - rb[1]), rb[5] - rb[2]);
    if (!(ext > 0)) ext = 1.0f;
    const float cell = ext / 64.0f;
    for (int a = 0; a < 3; a++) {
        int d = (int)floorf((rb[3 + a] - rb[a]) / cell) + 1;
        grid_out->dim[a] = d < 1 ? 1 : (d > 64 ? 64 : d);
        grid_out->min[a] = rb[a];
    }
    grid_out->cell = cell;
    const size_t cells = (size_t)grid_out->dim[0] * grid_out->dim[1] * grid_out->dim[2];
    hipLaunchKernelGGL(k_fill_u32, dim3(256), dim3(PB_BLOCK), 0, st, grid, cells, 0u);
    hipLaunchKernelGGL(k_density_grid, dim3(grid_n), dim3(PB_BLOCK), 0, st, ph, n, rb[0], rb[1], rb[2], cell, grid_out->dim[0], grid_out->dim[1], grid_out->dim[2], grid);
    return hipGetLastError();
}

// ---- per-cell start node of the gather's tree walk (DevPhotonMap::cell_start) -------------------------------------------------
__device__ __forceinline__ float boxbox_dist2(const float4 *b, const float lo[3], const float hi[3])
{
    const float4 blo = b[0], bhi = b[1];
    const float gx = fmaxf(fmaxf(blo.x - hi[0], lo[0] - bhi.x), 0.0f);
    const float gy = fmaxf(fmaxf(blo.y - hi[1], lo[1] - bhi.y), 0.0f);
    const float gz = fmaxf(fmaxf(blo.z - hi[2], lo[2] - bhi.z), 0.0f);
    return gx * gx + gy * gy + gz * gz;
}
__global__ __launch_bounds__(PB_BLOCK) void k_cell_start(const float4 *tbox, uint32_t n_leaves, float gx, float gy, float gz, float cell, int dx, int dy, int dz,
                                                          float radius, uint32_t *start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)(dx * dy * dz)) return;
    const int x = (int)(i % (uint32_t)dx), y = (int)((i / (uint32_t)dx) % (uint32_t)dy), z = (int)(i / (uint32_t)(dx * dy));
    // the cell, a little inflated: the query's cell index is computed in float from its position
    const float pad = 1e-3f * cell;
    const float lo[3] = {gx + x * cell - pad, gy + y * cell - pad, gz + z * cell - pad};
    const float hi[3] = {gx + (x + 1) * cell + pad, gy + (y + 1) * cell + pad, gz + (z + 1) * cell + pad};
    const float r2 = radius * radius * 1.0001f;
    uint32_t node = 1;
    for (;;) {
        if (!(4u * node < 2u * n_leaves && 2u * node < n_leaves)) break;       // no grandchildren below this node
        uint32_t hit = 0, which = 0;
        for (uint32_t g = 0; g < 4; g++)
            if (boxbox_dist2(tbox + 2 * (size_t)(4u * node + g), lo, hi) < r2) { hit++; which = g; }
        if (hit != 1) break;
        const uint32_t next = 4u * node + which;
        if (next >= n_leaves) break;                                           // a leaf: the walk lists leaves from their grandparent
        node = next;
    }
    start[i] = node;
}
void rtk_photon_cell_start(hipStream_t st, const float4 *tbox, uint32_t n_leaves, const float grid_min[3], float cell, const int dim[3], float radius, uint32_t *start)
{
    const uint32_t cells = (uint32_t)(dim[0] * dim[1] * dim[2]);
    hipLaunchKernelGGL(k_cell_start, dim3((cells + PB_BLOCK - 1) / PB_BLOCK), dim3(PB_BLOCK), 0, st, tbox, n_leaves, grid_min[0], grid_min[1], grid_min[2], cell,
                       dim[0], dim[1], dim[2], radius, start);
}
