// Micro-benchmarks behind the stage-2 design choice: what the MI355X memory system delivers for the
// three ways a query's 131072 survivor-token score rows can be produced.
//   gather   : random 256-byte fp16 centroid rows (one per token) from a table of S bytes -- the current stage-2 kernel's access
//   stream   : sequential 8 KB tiles of the 33.5 MB table, every wave walking the SAME sequence (query-stationary dense walk:
//              the table is shared through L2 by the waves of an XCD)
//   atomics  : 128-byte row-wise atomic max into an accumulator array of S bytes (centroid-stationary dense pass that scatters
//              token rows into per-(query, passage) maxima), workgroup scope and agent scope
//   lds      : the same scatter into a 128 KB LDS accumulator (one query per CU)
// Build: hipcc --offload-arch=gfx950 -O3 -o s2_design_probe s2_design_probe.hip ; run: ./s2_design_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

// each wave: ITER tiles of 32 random rows (256 B each); lane L fetches 16-byte piece L%16 of rows 4g + L/16, g = 0..7
template <int DEPTH>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, uint32_t nrows_mask, int iters, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) / 64 * 2654435761u + 12345u;   // per-wave stream
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it += DEPTH) {
        uint4 v[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int g = 0; g < 8; g++) {
                uint32_t s2 = seed + (uint32_t)(it + d) * 97u + (uint32_t)(4 * g + (lane >> 4)) * 7919u;
                const uint32_t row = lcg(s2) >> 8 & nrows_mask;
                v[d][g] = table[(size_t)row * 16 + (lane & 15)];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int g = 0; g < 8; g++) { acc.x ^= v[d][g].x; acc.y += v[d][g].y; acc.z ^= v[d][g].z; acc.w += v[d][g].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// the same gather, but the block's rows come from slice (blockIdx.x % 8) of the table only: workgroups are dealt to the 8 XCDs
// round-robin, so each XCD's L2 (4 MB) sees one slice of table/8 bytes -- a 32 MB table becomes eight L2-resident 4 MB slices
template <int DEPTH>
__global__ __launch_bounds__(256) void gather_sliced_kernel(const uint4* __restrict__ table, uint32_t slice_rows, int iters, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) / 64 * 2654435761u + 12345u;
    const uint32_t base = (blockIdx.x & 7) * slice_rows;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it += DEPTH) {
        uint4 v[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int g = 0; g < 8; g++) {
                uint32_t s2 = seed + (uint32_t)(it + d) * 97u + (uint32_t)(4 * g + (lane >> 4)) * 7919u;
                const uint32_t row = base + ((lcg(s2) >> 8) & (slice_rows - 1));
                v[d][g] = table[(size_t)row * 16 + (lane & 15)];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int g = 0; g < 8; g++) { acc.x ^= v[d][g].x; acc.y += v[d][g].y; acc.z ^= v[d][g].z; acc.w += v[d][g].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// the sliced gather again, rows going straight to LDS (global_load_lds_dwordx4: 1 KB = four rows per instruction) as the
// stage-2 kernels do: two 8 KB tiles in flight per wave, vmcnt(8) before a buffer is reused
__global__ __launch_bounds__(256) void gather_sliced_dma_kernel(const uint4* __restrict__ table, uint32_t slice_rows, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) char buf[4][2][8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) / 64 * 2654435761u + 12345u;
    const uint32_t base = (blockIdx.x & 7) * slice_rows;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint32_t s2 = seed + (uint32_t)it * 97u + (uint32_t)(4 * g + (lane >> 4)) * 7919u;
            const uint32_t row = base + ((lcg(s2) >> 8) & (slice_rows - 1));
            __builtin_amdgcn_global_load_lds(table + (size_t)row * 16 + (lane & 15), (__attribute__((address_space(3))) void*)(&buf[wave][it & 1][g * 1024]), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][0][lane] == 0x7b && iters < 0) sink[0] = 1;
}

// every wave walks tiles (start + it) % ntiles with the same `start` per block group -> shared through L2
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ table, uint32_t ntiles, int iters, int shared, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) / 64;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint32_t t = shared ? (wave & 7u) : (wave * 131u);
    for (int it = 0; it < iters; it++) {
        const uint4* p = table + (size_t)((t + (uint32_t)it) % ntiles) * 512;   // 8 KB tile = 512 uint4
        uint4 v[8];
#pragma unroll
        for (int g = 0; g < 8; g++) v[g] = p[g * 64 + lane];
#pragma unroll
        for (int g = 0; g < 8; g++) { acc.x ^= v[g].x; acc.y += v[g].y; acc.z ^= v[g].z; acc.w += v[g].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// half-wave h: row-wise atomic max on 32 consecutive dwords of a random row
template <int SCOPE>
__global__ __launch_bounds__(256) void atomic_kernel(uint32_t* acc, uint32_t nrows_mask, int iters) {
    const int lane = threadIdx.x & 63;
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) / 32 * 2654435761u + 777u;     // per-half-wave stream
    for (int it = 0; it < iters; it++) {
        const uint32_t row = lcg(seed) >> 8 & nrows_mask;
        const uint32_t val = (seed >> 4) ^ lane;
        __hip_atomic_fetch_max(acc + (size_t)row * 32 + (lane & 31), val, __ATOMIC_RELAXED, SCOPE);
    }
}

__global__ __launch_bounds__(1024) void lds_atomic_kernel(int iters, int swizzle, uint32_t* sink) {
    extern __shared__ uint32_t acc[];   // 1024 rows x 32 dwords = 128 KB
    for (int i = threadIdx.x; i < 1024 * 32; i += 1024) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) / 32 * 2654435761u + 99u;
    for (int it = 0; it < iters; it++) {
        const uint32_t row = lcg(seed) >> 8 & 1023u;
        const uint32_t col = swizzle ? ((lane + row) & 31) : (lane & 31);
        atomicMax(&acc[row * 32 + col], (seed >> 4) ^ lane);
    }
    __syncthreads();
    if (threadIdx.x == 0 && acc[5] == 0x12345678u) sink[0] = acc[7];
}


int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, ncu, prop.clockRate / 1000);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    const size_t big = (size_t)256 << 20;
    uint4* table; CK(hipMalloc(&table, big)); CK(hipMemset(table, 1, big));
    auto run = [&](const char* name, double bytes_or_ops, const char* unit, auto launch) {
        launch();  // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 3; r++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-58s %8.3f ms  %10.2f %s\n", name, ms, bytes_or_ops / (ms * 1e-3) / 1e9, unit);
        CK(hipGetLastError());
    };
    char name[128];
    // ---- gather ------------------------------------------------------------------------------------------------------
    for (int wpc : {8, 16}) {
        const int blocks = ncu * wpc / 4, iters = 512;
        for (size_t mb : {1, 2, 4, 8, 32}) {
            const uint32_t nrows = (uint32_t)((mb << 20) / 256);
            const double bytes = (double)blocks * 4 * iters * 8192.0;
            snprintf(name, sizeof name, "gather 256B rows, table %3zu MB, %2d waves/CU, depth 1", mb, wpc);
            run(name, bytes, "GB/s", [&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, nrows - 1, iters, sink); });
            snprintf(name, sizeof name, "gather 256B rows, table %3zu MB, %2d waves/CU, depth 2", mb, wpc);
            run(name, bytes, "GB/s", [&] { hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, nrows - 1, iters, sink); });
        }
    }
    // ---- gather, table cut into one slice per XCD ---------------------------------------------------------------------------
    for (int wpc : {8, 16}) {
        const int blocks = ncu * wpc / 4, iters = 512;
        for (size_t mb : {8, 16, 32, 64}) {
            const uint32_t slice_rows = (uint32_t)((mb << 20) / 256 / 8);
            const double bytes = (double)blocks * 4 * iters * 8192.0;
            snprintf(name, sizeof name, "gather 256B rows, table %3zu MB in 8 XCD slices, %2d waves/CU, depth 2", mb, wpc);
            run(name, bytes, "GB/s", [&] { hipLaunchKernelGGL(gather_sliced_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, slice_rows, iters, sink); });
        }
    }
    for (int wpc : {8, 16}) {
        const int blocks = ncu * wpc / 4, iters = 512;
        for (size_t mb : {8, 32, 64}) {
            const uint32_t slice_rows = (uint32_t)((mb << 20) / 256 / 8);
            const double bytes = (double)blocks * 4 * iters * 8192.0;
            snprintf(name, sizeof name, "gather -> LDS (DMA), table %3zu MB in 8 XCD slices, %2d waves/CU", mb, wpc);
            run(name, bytes, "GB/s", [&] { hipLaunchKernelGGL(gather_sliced_dma_kernel, dim3(blocks), dim3(256), 0, 0, table, slice_rows, iters, sink); });
        }
    }
    // ---- stream ------------------------------------------------------------------------------------------------------
    for (int shared : {1, 0}) {
        const int blocks = ncu * 4, iters = 1024;
        const uint32_t ntiles = (uint32_t)(((size_t)32 << 20) / 8192);
        snprintf(name, sizeof name, "stream 8KB tiles of a 32 MB table, 16 waves/CU, %s", shared ? "same walk (L2-shared)" : "disjoint walks");
        run(name, (double)blocks * 4 * iters * 8192.0, "GB/s", [&] { hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, table, ntiles, iters, shared, sink); });
    }
    // ---- atomics -----------------------------------------------------------------------------------------------------
    for (size_t mb : {2, 16, 128}) {
        const uint32_t nrows = (uint32_t)((mb << 20) / 128);
        const int blocks = ncu * 4, iters = 256;
        const double rows = (double)blocks * 8 * iters;
        snprintf(name, sizeof name, "row atomic umax (128 B), array %3zu MB, workgroup scope", mb);
        run(name, rows, "G rows/s", [&] { hipLaunchKernelGGL(atomic_kernel<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)table, nrows - 1, iters); });
        snprintf(name, sizeof name, "row atomic umax (128 B), array %3zu MB, agent scope", mb);
        run(name, rows, "G rows/s", [&] { hipLaunchKernelGGL(atomic_kernel<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)table, nrows - 1, iters); });
    }
    // ---- LDS ---------------------------------------------------------------------------------------------------------
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_atomic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int sw : {0, 1}) {
        const int iters = 4096;
        snprintf(name, sizeof name, "LDS row atomic max, 128 KB accumulator, 16 waves/CU, %s", sw ? "column rotated by row" : "plain layout");
        run(name, (double)ncu * 32 * iters, "G rows/s", [&] { hipLaunchKernelGGL(lds_atomic_kernel, dim3(ncu), dim3(1024), 128 * 1024, 0, iters, sw, sink); });
    }
    return 0;
}
