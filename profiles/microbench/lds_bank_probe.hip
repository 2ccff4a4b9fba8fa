// How the LDS of gfx950 serves a wave's 64 four-byte reads by ADDRESS PATTERN: the question behind the dense stage-1 forms' mask
// probes (64 random words of a K-bit mask per instruction; profiles/r06/s1_exact_ablation.txt: at lane-linear addresses the exact
// form runs 30 % faster).
//   hipcc --offload-arch=gfx950 -O3 -o lds_bank_probe lds_bank_probe.hip && ./lds_bank_probe
// One workgroup of WAVES waves per CU, every wave issues N ds_read_b32 back to back (8 in flight) at addresses
// base[lane] + (i * step) mod table; the table of patterns below sets base[lane].  Reported: LDS cycles per instruction and CU
// (wall clock of the launch x clock / instructions per CU) -- with 16 waves the pipe is saturated and latency is hidden.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void lds_kernel(const uint32_t* __restrict__ pat, int iters, uint32_t* sink, int words) {
    extern __shared__ uint32_t tab[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) tab[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t a = pat[lane];            // dword index of this lane's first read
    uint32_t acc = 0;
    const uint32_t mask = (uint32_t)words - 1u;
    for (int it = 0; it < iters; it++) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = tab[(a + 64u * (uint32_t)u * 0u + (uint32_t)(it * 8 + u) * 64u) & mask];   // (+64 dwords: the bank pattern of the lanes is kept)
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static double run(const std::vector<uint32_t>& pat, int waves, int words) {
    uint32_t *dp, *sink;
    CK(hipMalloc(reinterpret_cast<void**>(&dp), 64 * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CK(hipMemcpy(dp, pat.data(), 64 * 4, hipMemcpyHostToDevice));
    const int iters = 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, words * 4));
    hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(64 * waves), words * 4, 0, dp, iters, sink, words);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(64 * waves), words * 4, 0, dp, iters, sink, words);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(dp)); CK(hipFree(sink));
    return (double)ms * 1e-3 / ((double)iters * 8 * waves);   // seconds per instruction and CU
}

int main() {
    const int words = 16384;   // 64 KB
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const double clk = clk_khz * 1e3;
    printf("ds_read_b32, 64 lanes, 16 waves per CU; cycles per instruction at the reported clock of %.2f GHz\n", clk * 1e-9);
    std::mt19937 rng(7);
    struct P { const char* name; std::vector<uint32_t> pat; };
    std::vector<P> ps;
    auto mk = [&](const char* n, auto f) { P p; p.name = n; p.pat.resize(64); for (int l = 0; l < 64; l++) p.pat[l] = f(l); ps.push_back(p); };
    mk("lane (linear)", [](int l) { return (uint32_t)l; });
    mk("2 x lane", [](int l) { return (uint32_t)(2 * l); });
    mk("4 x lane", [](int l) { return (uint32_t)(4 * l); });
    mk("8 x lane", [](int l) { return (uint32_t)(8 * l); });
    mk("16 x lane", [](int l) { return (uint32_t)(16 * l); });
    mk("32 x lane", [](int l) { return (uint32_t)(32 * l); });
    mk("64 x lane (one bank)", [](int l) { return (uint32_t)(64 * l); });
    mk("all lanes one word", [](int) { return 5u; });
    mk("lane + 64 x perm (own bank of 64, rows scattered)", [&](int l) { return (uint32_t)(l + 64 * ((l * 37 + 11) % 61)); });
    mk("(lane % 32) + 32 x perm (own bank of 32 per half)", [&](int l) { return (uint32_t)((l % 32) + 32 * ((l * 37 + 11) % 61)); });
    mk("random", [&](int) { return (uint32_t)(rng() % 16384); });
    mk("random (another)", [&](int) { return (uint32_t)(rng() % 16384); });
    mk("pairs of lanes share a bank of 64 (2-way)", [&](int l) { return (uint32_t)((l / 2) + 64 * ((l * 37 + 11) % 61)); });
    mk("four lanes share a bank of 64 (4-way)", [&](int l) { return (uint32_t)((l / 4) + 64 * ((l * 37 + 11) % 61)); });
    mk("lanes l and l + 32 share a bank of 64", [&](int l) { return (uint32_t)((l % 32) + 64 * ((l * 37 + 11) % 61)); });
    mk("lanes l and l + 16 share a bank of 64", [&](int l) { return (uint32_t)((l % 16) + 16 * ((l / 32) % 2) * 0 + 64 * ((l * 37 + 11) % 61) + 16 * (l / 32)); });
    for (auto& p : ps) {
        const double s = run(p.pat, 16, words);
        printf("  %-58s %6.2f cycles\n", p.name, s * clk);
    }
    return 0;
}
