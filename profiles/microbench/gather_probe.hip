// What the chip delivers for SATURATED random record gathers -- the access pattern of a code-scanning stage 1
// (TPC/search/filter_pids.cpp:36-47 reads `codes[offset .. offset+doclen)` of every candidate passage: a contiguous
// run of doclen x 4 bytes at a random place of the code array).  Asked by the round-5 review: is "random 512-byte
// reads at 1.07 TB/s" a DRAM floor or a lack of requests in flight / of locality?
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
//
// Every group of REC/16 lanes reads one record of REC bytes (16 bytes per lane, the record's lanes contiguous), DEPTH
// records per lane in flight before the first is consumed.  The record index is a hash of (workgroup, wave, iteration):
// no index array is read.  Two placements of the records:
//   table  : uniformly random over the whole table (size TAB) -- every workgroup anywhere at any time;
//   window : uniformly random inside a window of WIN bytes that slides over the table with the iteration count, all
//            workgroups at the same place at the same time -- what a lock-step sweep of all queries over passage space
//            would see (a 1 M-passage index is touched 8-15 times per 256-query batch: see DESIGN 4.3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int LANES_PER_REC, int DEPTH, bool NT>
__global__ __launch_bounds__(512) void gather_kernel(const u32x4* __restrict__ tab, uint32_t nrec /* records in the region */,
                                                     uint32_t win_rec /* records per window (0: whole table) */, int iters,
                                                     uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LANES_PER_REC, l = lane % LANES_PER_REC;
    constexpr int RPW = 64 / LANES_PER_REC;   // records per wave and load instruction
    uint32_t acc = 0;
    const uint32_t wid = (blockIdx.x * 8 + wave) * RPW + sub;
    for (int it = 0; it < iters; it++) {
        u32x4 v[DEPTH];
        uint32_t base = 0, span = nrec;
        if (win_rec) {   // the window's start advances with the iteration: iters windows tile the table
            span = win_rec;
            base = (uint32_t)(((uint64_t)it * (nrec - win_rec)) / (iters > 1 ? iters - 1 : 1));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t r = base + mix(wid * 2654435761u + (uint32_t)(it * DEPTH + d) * 40503u + 12345u) % span;
            const u32x4* p = tab + (size_t)r * LANES_PER_REC + l;
            v[d] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LPR, int DEPTH, bool NT>
static double run(const u32x4* tab, size_t tab_bytes, size_t win_bytes, int wgs, int iters, uint32_t* sink) {
    const uint32_t rec_bytes = LPR * 16;
    const uint32_t nrec = (uint32_t)(tab_bytes / rec_bytes), win_rec = (uint32_t)(win_bytes / rec_bytes);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<LPR, DEPTH, NT>), dim3(wgs), dim3(512), 0, 0, tab, nrec, win_rec, iters, sink);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_kernel<LPR, DEPTH, NT>), dim3(wgs), dim3(512), 0, 0, tab, nrec, win_rec, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * 512 * 16.0 * DEPTH * iters;
    return bytes / (ms * 1e-3) / 1e12;
}

template <int LPR>
static void sweep(const u32x4* tab, size_t tab_bytes, uint32_t* sink, int wgs_per_cu) {
    const int wgs = 256 * wgs_per_cu;
    // the same number of bytes per configuration: ~8 GB
    auto iters_for = [&](int depth) { return (int)(8e9 / ((double)wgs * 512 * 16 * depth)); };
    printf("record %4d B, %d workgroups of 8 waves per CU\n", LPR * 16, wgs_per_cu);
    const size_t wins[] = {0, (size_t)256 << 20, (size_t)64 << 20, (size_t)16 << 20, (size_t)4 << 20};
    for (size_t w : wins) {
        if (w >= tab_bytes) continue;
        const double a = run<LPR, 2, false>(tab, tab_bytes, w, wgs, iters_for(2), sink);
        const double b = run<LPR, 4, false>(tab, tab_bytes, w, wgs, iters_for(4), sink);
        const double c = run<LPR, 8, false>(tab, tab_bytes, w, wgs, iters_for(8), sink);
        const double d = run<LPR, 8, true>(tab, tab_bytes, w, wgs, iters_for(8), sink);
        if (w) printf("  window %4zu MB sliding over %5zu MB : ", w >> 20, tab_bytes >> 20);
        else printf("  whole table %5zu MB               : ", tab_bytes >> 20);
        printf("depth 2 %6.2f  depth 4 %6.2f  depth 8 %6.2f  depth 8 nt %6.2f  TB/s\n", a, b, c, d);
    }
}

int main(int argc, char** argv) {
    const size_t tab_mb = argc > 1 ? (size_t)atol(argv[1]) : 512;   // the 1 M x 128 code array is 512 MB
    const size_t tab_bytes = tab_mb << 20;
    u32x4* tab;
    uint32_t* sink;
    CK(hipMalloc(reinterpret_cast<void**>(&tab), tab_bytes));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CK(hipMemset(tab, 1, tab_bytes));
    CK(hipDeviceSynchronize());
    for (int wpc : {2, 4}) {
        sweep<32>(tab, tab_bytes, sink, wpc);   // 512-byte records: 128 codes
        sweep<16>(tab, tab_bytes, sink, wpc);   // 256 bytes: ~57 distinct codes of a passage of the built index
        sweep<8>(tab, tab_bytes, sink, wpc);    // 128 bytes: one cache line
    }
    // a larger table: nothing of it stays in the 256 MB Infinity Cache
    if (argc <= 2) {
        const size_t big = (size_t)4096 << 20;
        u32x4* tb;
        CK(hipMalloc(reinterpret_cast<void**>(&tb), big));
        CK(hipMemset(tb, 1, big));
        CK(hipDeviceSynchronize());
        sweep<32>(tb, big, sink, 4);
        sweep<16>(tb, big, sink, 4);
        CK(hipFree(tb));
    }
    return 0;
}
