// checks flmr_both_halves (v_permlane32_swap) against __shfl_xor(v, 32): prints mismatching lanes
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_common.h"
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_device.h"
#include <cstdio>
thread_local char flmr_err_buf[512] = {0};
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float v = 100.0f + lane;
    float lo, hi; flmr_both_halves(v, lo, hi);
    out[lane] = lo; out[64 + lane] = hi; out[128 + lane] = __shfl_xor(v, 32, 64);
    out[192 + lane] = flmr_xhalf_max(v); out[256 + lane] = flmr_xhalf_sum(v);
}
int main() {
    float* d; hipMalloc(&d, 320 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: lo %.0f hi %.0f shfl %.0f max %.0f sum %.0f\n", l, h[l], h[64 + l], h[128 + l], h[192 + l], h[256 + l]);
    return 0;
}
