// tools_peaks.hip -- the ceilings bench.py's roofline quotes, measured on the GPU it runs on (gfx950 / MI355X).
//   build:  hipcc -O3 --offload-arch=gfx950 tools_peaks.hip -o raytracing_folder_amd/lib/tools_peaks
//   run:    raytracing_folder_amd/lib/tools_peaks > profiles/r04_peaks.json        (under gpurun; ~5 s)
// (i)   wave64 v_fma_f32 issue rate per SIMD at 1..8 waves per SIMD              -> the VALU peak
// (ii)  16-byte-per-lane loads of a 16 KiB window that stays in the CU's vector L1 -> the L1 peak, bytes per clock and CU
// (iii) k_gather's access shape: every 16-lane group reads one 256-byte sub-leaf from each of two 17 MB arrays (32 B per
//       photon slot), two steps in flight, 5 workgroups of 4 waves per CU -- sub-leaf ids uniformly random over the table
//       (every read an L2 / Infinity Cache access) and random inside a 32 KiB window per workgroup (mostly L1 hits)
// Rates are wall-clock (HIP events around 3 launches after a warm-up): work of the whole launch / its time.  The shader clock each
// kernel ran at is reported beside them: s_memtime ticks / s_memrealtime ticks (100 MHz) over the median wave's loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fma(float *out, int iters, unsigned long long *cyc)
{
    float a[16];
    for (int j = 0; j < 16; j++) a[j] = (float)(threadIdx.x + j);
    const float b = 1.0000001f, c = 1.0e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int j = 0; j < 16; j++) s += a[j];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

// every lane loads 16 bytes, a wave 1 KiB contiguous; 8 independent loads per iteration from a window of `window_vec4` float4
__global__ __launch_bounds__(256) void k_l1(const float4 *buf, uint32_t window_vec4, int iters, float *out, unsigned long long *cyc)
{
    const uint32_t mask = window_vec4 - 1u;
    uint32_t at = threadIdx.x;
    float s = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = buf[(at + 256u * k) & mask];
        at += 2048u;
#pragma unroll
        for (int k = 0; k < 8; k++) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

// k_gather's pass-1 shape: lane l of a wave reads slot (l & 15) of sub-leaf id[l >> 4] from pa and from pb (256-byte runs)
__global__ __launch_bounds__(256) void k_gshape(const float4 *pa, const float4 *pb, uint32_t n_sub, uint32_t window_sub, int iters,
                                                float *out, unsigned long long *cyc)
{
    const uint32_t lane = threadIdx.x & 63, grp = lane >> 4, sl = lane & 15;
    uint32_t rng = (blockIdx.x * 256u + (threadIdx.x & ~15u)) * 2654435761u + 12345u;      // one stream per 16-lane group
    const uint32_t span = window_sub ? window_sub : n_sub;
    const uint32_t base = window_sub ? __umulhi(blockIdx.x * 2654435761u, n_sub - window_sub) : 0u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return base + __umulhi(rng, span); };
    float s = 0;
    (void)grp;
    typedef float f4 __attribute__((ext_vector_type(4)));
    // explicit dwordx4 loads and counted waits (the compiler splits a float4 load whose components are used at different times)
    auto ld = [&](f4 &a, f4 &b, uint32_t id) {
        const float4 *qa = pa + (id * 16u + sl), *qb = pb + (id * 16u + sl);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=&v"(a), "=&v"(b) : "v"(qa), "v"(qb) : "memory");
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f4 a0, b0, a1, b1;
    ld(a0, b0, next());
    for (int i = 0; i < iters; i++) {
        ld(a1, b1, next());
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(a0), "+v"(b0) :: "memory");
        s += (a0.x * b0.y + a0.w) + (a0.y * b0.x + a0.z) + (b0.z + b0.w);
        ld(a0, b0, next());
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(a1), "+v"(b1) :: "memory");
        s += (a1.x * b1.y + a1.w) + (a1.y * b1.x + a1.z) + (b1.z + b1.w);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(b0) :: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) out[0] = s + a0.x + b0.x;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

// shader clock (GHz) of the median wave: its s_memtime ticks over its s_memrealtime ticks (100 MHz)
static double median_clock_ghz(unsigned long long *d_cyc, int n)
{
    std::vector<unsigned long long> h(2 * (size_t)n);
    CK(hipMemcpy(h.data(), d_cyc, 2 * (size_t)n * 8, hipMemcpyDeviceToHost));
    std::vector<double> g(n);
    for (int i = 0; i < n; i++) g[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0;
    std::sort(g.begin(), g.end());
    return g[n / 2];
}
template <class L> static double timed_ms(L &&launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / 3.0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 2 * 8 * 4 * cus * 8));
    printf("{\"what\": \"tools_peaks.hip: measured ceilings for bench.py's roofline\", \"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d,\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    // (i) VALU issue
    printf(" \"valu_fma\": [");
    double best_valu = 0;
    for (int w = 1; w <= 8; w++) {
        const int iters = 4000, blocks = cus * w;
        const double ms = timed_ms([&]() { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); });
        const double ghz = median_clock_ghz(cyc, blocks * 4), inst = (double)iters * 128.0;
        const double ginst = (double)blocks * 4 * inst / (ms * 1e-3) / 1e9;
        best_valu = std::max(best_valu, ginst);
        printf("%s{\"waves_per_simd\": %d, \"chip_Gwave_inst_per_s\": %.1f, \"shader_clock_ghz\": %.3f, \"wave_inst_per_clk_per_simd\": %.4f}",
               w > 1 ? ", " : "", w, ginst, ghz, ginst / (cus * 4 * ghz));
    }
    printf("],\n \"valu_peak_Gwave_inst_per_s\": %.1f,\n", best_valu);
    // (ii) L1-resident 16-byte loads
    const uint32_t n_sub = 69632;                                          // 17 MiB per array: the bench frame's photon structure (34 MB)
    float4 *pa, *pb;
    CK(hipMalloc(&pa, (size_t)n_sub * 256)); CK(hipMalloc(&pb, (size_t)n_sub * 256));
    CK(hipMemset(pa, 0, (size_t)n_sub * 256)); CK(hipMemset(pb, 0, (size_t)n_sub * 256));
    printf(" \"l1_resident_loads_16B_per_lane\": [");
    double best_l1 = 0;
    for (int w = 1; w <= 8; w++) {
        const int iters = 2000, blocks = cus * w;
        const double ms = timed_ms([&]() { hipLaunchKernelGGL(k_l1, dim3(blocks), dim3(256), 0, 0, pa, 1024u, iters, out, cyc); });
        const double ghz = median_clock_ghz(cyc, blocks * 4), bytes_wave = (double)iters * 8 * 1024.0;
        const double gbs = (double)blocks * 4 * bytes_wave / (ms * 1e-3) / 1e9;
        best_l1 = std::max(best_l1, gbs);
        printf("%s{\"waves_per_simd\": %d, \"chip_GBps\": %.0f, \"shader_clock_ghz\": %.3f, \"bytes_per_clk_per_cu\": %.2f}", w > 1 ? ", " : "", w, gbs, ghz, gbs / (cus * ghz));
    }
    printf("],\n \"l1_peak_GBps\": %.0f,\n", best_l1);
    // (iii) the gather's access shape at its occupancy (5 workgroups of 4 waves per CU)
    printf(" \"gather_shape_5_waves_per_simd\": {");
    const char *names[3] = {"random_over_34MB", "window_32KiB_per_workgroup", "window_256KiB_per_workgroup"};
    const uint32_t windows[3] = {0u, 64u, 512u};
    for (int v = 0; v < 3; v++) {
        const int iters = 1500, blocks = cus * 5;
        const double ms = timed_ms([&]() { hipLaunchKernelGGL(k_gshape, dim3(blocks), dim3(256), 0, 0, pa, pb, n_sub, windows[v], iters, out, cyc); });
        const double ghz = median_clock_ghz(cyc, blocks * 4), bytes_wave = (double)(2 * iters + 1) * 2048.0;
        const double gbs = (double)blocks * 4 * bytes_wave / (ms * 1e-3) / 1e9;
        printf("%s\"%s\": {\"chip_GBps\": %.0f, \"shader_clock_ghz\": %.3f, \"bytes_per_clk_per_cu\": %.2f, \"ms\": %.3f}", v ? ", " : "", names[v], gbs, ghz, gbs / (cus * ghz), ms);
    }
    printf("}\n}\n");
    return 0;
}
