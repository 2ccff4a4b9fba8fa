// What the whole chip sustains on v_mfma_f32_32x32x16_f16 (the instruction every split kernel of this library issues), against
// the 2.5 PFLOP/s dense-fp16 figure `bench.py` prices stage 0 / 2 / 3 with: every CU runs W waves per SIMD, each wave C
// independent accumulator chains, nothing but MFMAs in the loop.  Reports TFLOP/s from HIP events and the s_memtime ticks per
// MFMA and SIMD, so that the phase profiles of the kernels (-DX2_PROFILE, -DS0Q_PROFILE: s_memtime ticks) can be read in time.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void peak(int iters, float* sink, long long* ticks) {
    hf8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(threadIdx.x * 0.001f + k); b[k] = (_Float16)(k * 0.5f); }
    f32x16 c[CHAINS];
    for (int j = 0; j < CHAINS; j++) for (int k = 0; k < 16; k++) c[j][k] = 0.0f;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16 / CHAINS; g++)
#pragma unroll
            for (int j = 0; j < CHAINS; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int j = 0; j < CHAINS; j++) s += c[j][0];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int CHAINS>
static void run(int waves_per_simd, int ncu) {
    const int iters = 20000;
    float* sink; long long* ticks;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&ticks, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(ncu), block(64 * 4 * waves_per_simd);
    hipLaunchKernelGGL(peak<CHAINS>, grid, block, 0, 0, 100, sink, ticks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(peak<CHAINS>, grid, block, 0, 0, iters, sink, ticks);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double mfmas_per_simd = (double)iters * 16 * waves_per_simd;
    const double flop = (double)ncu * 4 * mfmas_per_simd * 2.0 * 32 * 32 * 16;
    printf("CUs %3d  waves/SIMD %d  chains/wave %d : %7.1f TFLOP/s   %.1f ns per MFMA and SIMD   %.1f s_memtime ticks per MFMA and SIMD (tick = %.3f ns)\n",
           ncu, waves_per_simd, CHAINS, flop / (ms * 1e-3) / 1e12, ms * 1e6 / mfmas_per_simd, (double)t / mfmas_per_simd, ms * 1e6 / (double)t);
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main() {
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.name, ncu, p.clockRate / 1000);
    for (int ncus : {1, ncu}) {
        run<1>(1, ncus); run<2>(1, ncus); run<4>(1, ncus);
        run<1>(2, ncus); run<2>(2, ncus); run<4>(2, ncus);
        run<2>(3, ncus); run<2>(4, ncus);
    }
    return 0;
}
