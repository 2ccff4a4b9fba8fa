// What does the chip deliver when the 256-byte centroid rows of stage 2 land in REGISTERS in MFMA A-operand order (no LDS
// staging)?  Companion of s2_design_probe.hip (whose register gather reads a whole row with 16 consecutive lanes -- the DMA's
// layout, which an MFMA cannot consume without a transpose).
//   A16 : v_mfma_f32_16x16x32_f16 order.  A tile is 16 rows; load s (s = 0..3) gives lane L the 16-byte piece 4s + L/16 of row
//         L%16, i.e. every instruction touches 16 rows x 64 contiguous bytes.
//   B32 : v_mfma_f32_32x32x16_f16 order.  A tile is 32 rows; load s (s = 0..7) gives lane L piece 2s + L/32 of row L%32: 32 rows
//         x 32 bytes per instruction.
//   R16 : the design probe's layout for reference (piece L%16 of rows 4g + L/16: 4 rows x 256 bytes per instruction).
// Each with the rows only (xor-folded) and, for A16, with the stage's arithmetic on top: 8 MFMAs per 16-row tile against 32
// query columns, the in-lane column maxima and a 128-byte flush every other tile.  The table is 32 MB cut into eight 4 MB
// slices, workgroup L confined to slice L % 8 (the XCD-sliced stage 2's access); rows are drawn per (wave, tile, row).
// Build: hipcc --offload-arch=gfx950 -O3 -o profiles/microbench/s2_reg_probe profiles/microbench/s2_reg_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// MODE 0: A16, 1: B32, 2: R16.  DEPTH tiles of 4 KB (A16) / 8 KB (B32, R16) requested before the first is consumed.
template <int MODE, int DEPTH, int MINW>
__global__ __launch_bounds__(256, MINW) void gather_reg_kernel(const uint4* __restrict__ table, uint32_t slice_rows, int tiles, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) / 64;
    const uint32_t base = (blockIdx.x & 7) * slice_rows;
    constexpr int NL = MODE == 0 ? 4 : 8;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 1
    for (int t = 0; t < tiles; t += DEPTH) {
        uint4 v[DEPTH][NL];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int s = 0; s < NL; s++) {
                uint32_t rowid, piece;
                if (MODE == 0) { rowid = lane & 15; piece = 4 * s + (lane >> 4); }
                else if (MODE == 1) { rowid = lane & 31; piece = 2 * s + (lane >> 5); }
                else { rowid = 4 * s + (lane >> 4); piece = lane & 15; }
                const uint32_t row = base + (mix(wave * 0x9e3779b9u + (uint32_t)(t + d) * 131u + rowid) & (slice_rows - 1));
                v[d][s] = table[(size_t)row * 16 + piece];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int s = 0; s < NL; s++) { acc.x ^= v[d][s].x; acc.y += v[d][s].y; acc.z ^= v[d][s].z; acc.w += v[d][s].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// A16 with the arithmetic: a rotating register pipeline AHEAD tiles deep (tile t + AHEAD is requested when tile t is consumed);
// the row ids come from a code stream in memory (one coalesced 64-byte read per tile, requested 2 AHEAD + 1 tiles early).
template <int AHEAD, int MINW>
__global__ __launch_bounds__(256, MINW) void stage_like_kernel(const uint4* __restrict__ table, const int32_t* __restrict__ codes,
                                                                 const _Float16* __restrict__ q, uint32_t slice_rows, int tiles,
                                                                 float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) / 64;
    const uint32_t base = (blockIdx.x & 7) * slice_rows;
    const int r = lane & 15, kq = lane >> 4;
    h8 bq[2][4];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int s = 0; s < 4; s++) bq[c][s] = *reinterpret_cast<const h8*>(q + ((size_t)(16 * c + r)) * 128 + 32 * s + 8 * kq);
    const int32_t* cs = codes + (size_t)wave * tiles * 16;
    constexpr int CA = 2 * AHEAD + 1;   // the code of tile t + AHEAD is older than the rows of tile t: the wait for those covers it
    int32_t cd[CA];
    uint4 v[AHEAD][4];
#pragma unroll
    for (int t = 0; t < CA; t++) cd[t] = cs[t * 16 + r];
#pragma unroll
    for (int t = 0; t < AHEAD; t++) {
        const uint32_t row = base + ((uint32_t)cd[t] & (slice_rows - 1));
#pragma unroll
        for (int s = 0; s < 4; s++) v[t][s] = table[(size_t)row * 16 + 4 * s + kq];
    }
    float cm0 = -9999.0f, cm1 = -9999.0f;
    float* o = out + (size_t)wave * 64;
    // tiles is a multiple of AHEAD * CA so that every register index below is static
    for (int t0 = 0; t0 < tiles; t0 += AHEAD * CA) {
#pragma unroll
        for (int u = 0; u < AHEAD * CA; u++) {
            const int t = t0 + u;
            f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            h8 av[4];
#pragma unroll
            for (int s = 0; s < 4; s++) av[s] = __builtin_bit_cast(h8, v[u % AHEAD][s]);
            // refill this slot with tile t + AHEAD (its code was requested two tiles before that), and request code t + CA
            {
                int tn = t + AHEAD; tn = tn < tiles ? tn : tiles - 1;
                const uint32_t row = base + ((uint32_t)cd[(u + AHEAD) % CA] & (slice_rows - 1));
#pragma unroll
                for (int s = 0; s < 4; s++) v[u % AHEAD][s] = table[(size_t)row * 16 + 4 * s + kq];
                int tc = t + CA; tc = tc < tiles ? tc : tiles - 1;
                cd[u % CA] = cs[tc * 16 + r];
                (void)tn;
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[s], bq[0][s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[s], bq[1][s], a1, 0, 0, 0);
            }
            cm0 = fmaxf(fmaxf(fmaxf(cm0, a0[0]), fmaxf(a0[1], a0[2])), a0[3]);
            cm1 = fmaxf(fmaxf(fmaxf(cm1, a1[0]), fmaxf(a1[1], a1[2])), a1[3]);
            if (u & 1) {   // a "passage" ends every other tile: combine the four row groups, 128-byte store
                float x0 = fmaxf(cm0, __shfl_xor(cm0, 16, 64)), x1 = fmaxf(cm1, __shfl_xor(cm1, 16, 64));
                x0 = fmaxf(x0, __shfl_xor(x0, 32, 64)); x1 = fmaxf(x1, __shfl_xor(x1, 32, 64));
                if (lane < 16) { o[r] = x0; o[16 + r] = x1; }
                cm0 = -9999.0f; cm1 = -9999.0f;
            }
        }
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    const size_t tb = (size_t)32 << 20;
    uint4* table; CK(hipMalloc(&table, tb)); CK(hipMemset(table, 0x3c, tb));
    const uint32_t slice_rows = (uint32_t)(tb / 256 / 8);
    auto run = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < 3; k++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-78s %8.3f ms  %9.1f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
        CK(hipGetLastError());
    };
    char name[160];
    const int tiles16 = 1260, tiles32 = 630;   // 5.2 MB of rows per wave
#define GR(MODE, DEPTH, MINW, WPC, LABEL) do { \
        const int blocks = ncu * (WPC) / 4; \
        const int tl = (MODE) == 0 ? tiles16 : tiles32; \
        snprintf(name, sizeof name, "%s, %d tiles in flight, %2d waves/CU", LABEL, DEPTH, WPC); \
        run(name, (double)blocks * 4 * tl * ((MODE) == 0 ? 4096.0 : 8192.0), [&] { \
            hipLaunchKernelGGL((gather_reg_kernel<MODE, DEPTH, MINW>), dim3(blocks), dim3(256), 0, 0, table, slice_rows, tl, sink); }); \
    } while (0)
    GR(2, 2, 2, 8, "R16 (4 rows x 256 B per instruction), rows only");
    GR(2, 2, 4, 16, "R16 (4 rows x 256 B per instruction), rows only");
    GR(0, 2, 2, 8, "A16 (16 rows x 64 B per instruction), rows only");
    GR(0, 4, 2, 8, "A16 (16 rows x 64 B per instruction), rows only");
    GR(0, 4, 3, 12, "A16 (16 rows x 64 B per instruction), rows only");
    GR(0, 2, 4, 16, "A16 (16 rows x 64 B per instruction), rows only");
    GR(0, 4, 4, 16, "A16 (16 rows x 64 B per instruction), rows only");
    GR(1, 1, 2, 8, "B32 (32 rows x 32 B per instruction), rows only");
    GR(1, 2, 2, 8, "B32 (32 rows x 32 B per instruction), rows only");
    GR(1, 2, 4, 16, "B32 (32 rows x 32 B per instruction), rows only");

    // ---- with the arithmetic ----
    const int maxwaves = ncu * 16;
    int32_t* codes; CK(hipMalloc(&codes, (size_t)maxwaves * tiles16 * 16 * 4));
    {
        int32_t* h = (int32_t*)malloc((size_t)maxwaves * tiles16 * 16 * 4);
        uint32_t s = 12345u;
        for (size_t k = 0; k < (size_t)maxwaves * tiles16 * 16; k++) { s = s * 1664525u + 1013904223u; h[k] = (int32_t)(s >> 8); }
        CK(hipMemcpy(codes, h, (size_t)maxwaves * tiles16 * 16 * 4, hipMemcpyHostToDevice));
        free(h);
    }
    _Float16* q; CK(hipMalloc(&q, 32 * 128 * 2)); CK(hipMemset(q, 0x3c, 32 * 128 * 2));
    float* out; CK(hipMalloc(&out, (size_t)maxwaves * 64 * 4));
#define SL(AHEAD, MINW, WPC) do { \
        const int blocks = ncu * (WPC) / 4; \
        snprintf(name, sizeof name, "A16 + 8 MFMA 16x16x32 + maxima + flush, %d tiles ahead, %2d waves/CU", AHEAD, WPC); \
        run(name, (double)blocks * 4 * tiles16 * 4096.0, [&] { \
            hipLaunchKernelGGL((stage_like_kernel<AHEAD, MINW>), dim3(blocks), dim3(256), 0, 0, table, codes, q, slice_rows, tiles16, out); }); \
    } while (0)
    SL(2, 2, 8);
    SL(3, 2, 8);
    SL(4, 2, 8);
    SL(3, 3, 12);
    SL(4, 3, 12);
    SL(2, 4, 16);
    SL(3, 4, 16);
    SL(4, 4, 16);
    printf("(stage 2 today: 41 GB of padded rows per 1024 queries in 2.45 ms = 16.7 TB/s through LDS)\n");
    return 0;
}
