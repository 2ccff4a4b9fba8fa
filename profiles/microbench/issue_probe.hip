// How MFMA and other instructions share a SIMD's issue on gfx950: cycles per loop iteration (s_memtime) of hand-written
// instruction sequences, at one and two waves per SIMD.  Behind DESIGN.md's reading of the stage-2 / S3 ablations.
//   mfma16      16 x v_mfma_f32_32x32x16_f16 (two accumulator chains)
//   valuN       N independent v_fma_f32
//   blocked     16 MFMAs, then N VALU
//   interleaved (1 MFMA, N/16 VALU) x 16
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip ; run: ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VALU(r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y))
#define VALU4 VALU(f[0]); VALU(f[1]); VALU(f[2]); VALU(f[3]);
#define VALU8 VALU4 VALU(f[4]); VALU(f[5]); VALU(f[6]); VALU(f[7]);
#define SALU4 asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(sc));

template <int MODE, int STAGGER = 0>
__global__ __launch_bounds__(512) void probe(int iters, long long* out, float* sink) {
    extern __shared__ char smem[];
    hf8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(threadIdx.x * 0.001f + k); b[k] = (_Float16)(k * 0.5f); }
    f32x16 c0, c1, c2;
    for (int k = 0; k < 16; k++) { c0[k] = 0.0f; c1[k] = 0.0f; c2[k] = 0.0f; }
    float f[8];
    for (int k = 0; k < 8; k++) f[k] = (float)k;
    const float x = 1.0001f, y = 0.5f;
    unsigned sc = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (STAGGER) {  // waves in odd slots of their SIMD start half an iteration late: the two waves of a SIMD leave lockstep
        if (blockDim.x == 512 ? (threadIdx.x >> 8) : (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1)) __builtin_amdgcn_s_sleep(STAGGER);
    }
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { for (int g = 0; g < 8; g++) { MFMA(c0); MFMA(c1); } }
        if (MODE == 1) { for (int g = 0; g < 8; g++) { VALU8 } }
        if (MODE == 2) { for (int g = 0; g < 8; g++) { MFMA(c0); MFMA(c1); } for (int g = 0; g < 8; g++) { VALU8 } }
        if (MODE == 3) { for (int g = 0; g < 8; g++) { MFMA(c0); VALU4 MFMA(c1); VALU4 } }
        if (MODE == 4) { for (int g = 0; g < 8; g++) { MFMA(c0); VALU8 MFMA(c1); VALU8 } }
        if (MODE == 5) { for (int g = 0; g < 8; g++) { MFMA(c0); MFMA(c1); } for (int g = 0; g < 16; g++) { VALU8 } }
        if (MODE == 6) { for (int g = 0; g < 16; g++) { VALU8 } }
        if (MODE == 7) { for (int g = 0; g < 8; g++) { MFMA(c0); SALU4 MFMA(c1); SALU4 } }
        // S3's 24 MFMAs per tile: hi chain + lo chain(s)
        if (MODE == 9) { for (int g = 0; g < 8; g++) { MFMA(c0); MFMA(c1); MFMA(c1); } }            // as built: the two lo products back to back on ONE accumulator
        if (MODE == 10) { for (int g = 0; g < 8; g++) { MFMA(c0); MFMA(c1); MFMA(c2); } }          // three accumulators
        if (MODE == 11) { for (int g = 0; g < 8; g++) { MFMA(c1); MFMA(c0); MFMA(c1); } }           // one accumulator for lo, the hi product between its two
        if (MODE == 12) { for (int g = 0; g < 24; g++) { MFMA(c0); } }                                 // one chain
        if (MODE == 8) { for (int g = 0; g < 8; g++) { MFMA(c0); VALU(f[0]); VALU(f[1]); MFMA(c1); VALU(f[2]); VALU(f[3]); } }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = t1 - t0;
    float s = 0; for (int k = 0; k < 16; k++) s += c0[k] + c1[k] + c2[k]; for (int k = 0; k < 8; k++) s += f[k];
    if (s == 12345.678f) sink[0] = s + sc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, iters = 2000;
    long long* out; float* sink; CK(hipMalloc(&out, ncu * 8 * 8)); CK(hipMalloc(&sink, 64));
    const char* names[13] = {"mfma16", "valu64", "blocked: mfma16 then valu64", "interleaved: (mfma, valu4) x16", "interleaved: (mfma, valu8) x16",
                            "blocked: mfma16 then valu128", "valu128", "interleaved: (mfma, salu4) x16", "interleaved: (mfma, valu2) x16",
                            "mfma24: (hi, lo, lo) x8, 2 accumulators", "mfma24: (hi, lo1, lo2) x8, 3 accumulators", "mfma24: (lo, hi, lo) x8, 2 accumulators", "mfma24: one chain"};
    for (int wps = 1; wps <= 2; wps++) {
        const size_t lds = wps == 1 ? 100 * 1024 : 60 * 1024;  // one / two 256-thread blocks per CU
        printf("%d wave(s) per SIMD\n", wps);
        auto run = [&](auto kern, int mode) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(ncu * wps), dim3(256), lds, 0, iters, out, sink);
            CK(hipDeviceSynchronize());
            long long h[8192]; CK(hipMemcpy(h, out, sizeof(long long) * ncu * wps * 4, hipMemcpyDeviceToHost));
            double s = 0; for (int k = 0; k < ncu * wps * 4; k++) s += (double)h[k];
            printf("  %-36s %8.1f cycles per iteration per wave\n", names[mode], s / (ncu * wps * 4) / iters);
        };
        if (wps == 2) {
            auto run512 = [&](auto kern, int mode, const char* tag) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
                hipLaunchKernelGGL(kern, dim3(ncu), dim3(512), 100 * 1024, 0, iters, out, sink);
                CK(hipDeviceSynchronize());
                long long h[8192]; CK(hipMemcpy(h, out, sizeof(long long) * ncu * 8, hipMemcpyDeviceToHost));
                double s = 0; for (int k = 0; k < ncu * 8; k++) s += (double)h[k];
                printf("  %-36s %8.1f cycles per iteration per wave  [512-thread blocks, %s]\n", names[mode], s / (ncu * 8) / iters, tag);
            };
            run512(probe<2, 0>, 2, "waves in step"); run512(probe<2, 6>, 2, "waves 4..7 start 384 cycles late");
            run512(probe<5, 0>, 5, "waves in step"); run512(probe<5, 10>, 5, "waves 4..7 start 640 cycles late");
        }
        run(probe<0>, 0); run(probe<9>, 9); run(probe<10>, 10); run(probe<11>, 11); run(probe<12>, 12); run(probe<1>, 1); run(probe<2>, 2); run(probe<3>, 3); run(probe<8>, 8); run(probe<4>, 4); run(probe<5>, 5); run(probe<6>, 6); run(probe<7>, 7);
    }
    return 0;
}
