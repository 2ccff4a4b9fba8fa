// Checks the VALU forms of the "keep one, send one" exchanges of the transpose-reduce in s0_centroid_scores_qs2's inline idx
// decision against their __shfl_xor (ds_bpermute) forms, lane by lane:
//   xor 16: v_permlane16_swap_b32 (gfx950)   xor 8: DPP row_ror:8   xor 4: DPP row_shl:4 / row_shr:4 by bank   xor 2 / 1: DPP quad_perm
//   hipcc --offload-arch=gfx950 -O3 -o xor_exchange_check xor_exchange_check.hip && ./xor_exchange_check
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_common.h"
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_device.h"
#include <cstdio>
thread_local char flmr_err_buf[512] = {0};
__global__ void k(float* out) {
    const int lane = threadIdx.x, i = lane & 31;
    const float x0 = 1000.0f + lane * 3.0f + ((lane * 7) % 5), x1 = 2000.0f - lane * 2.0f + ((lane * 11) % 7);
    float* o = out + lane;
    {   // xor 16
        const bool up = (i & 16) != 0;
        const float keep = up ? x1 : x0, send = up ? x0 : x1;
        o[0] = fmaxf(keep, __shfl_xor(send, 16, 64));
        o[64] = flmr_x16_max(x0, x1);
    }
    {   // xor 8
        const bool up = (i & 8) != 0;
        const float keep = up ? x1 : x0, send = up ? x0 : x1;
        o[128] = fmaxf(keep, __shfl_xor(send, 8, 64));
        o[192] = flmr_fmax_raw(keep, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128 /* row_ror:8 */, 0xF, 0xF, false)));
    }
    {   // xor 2
        const bool up = (i & 2) != 0;
        const float keep = up ? x1 : x0, send = up ? x0 : x1;
        o[256] = fmaxf(keep, __shfl_xor(send, 2, 64));
        o[320] = flmr_fmax_raw(keep, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false)));
    }
    {   // xor 4: two DPP moves, row_shl:4 into the lanes with bit 2 clear (banks 0 and 2 of a row), row_shr:4 into the others
        const bool up = (i & 4) != 0;
        const float keep = up ? x1 : x0, send = up ? x0 : x1;
        o[512] = fmaxf(keep, __shfl_xor(send, 4, 64));
        o[576] = flmr_fmax_raw(keep, flmr_dpp_xor4(send));
    }
    {   // xor 1
        o[384] = fmaxf(x0, __shfl_xor(x0, 1, 64));
        o[448] = flmr_fmax_raw(x0, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x0), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false)));
    }
}
int main() {
    float* d; hipMalloc(&d, 640 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[640]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 5; p++)
        for (int l = 0; l < 64; l++)
            if (h[128 * p + l] != h[128 * p + 64 + l]) { if (bad < 8) printf("pair %d lane %d: shfl %.0f valu %.0f\n", p, l, h[128 * p + l], h[128 * p + 64 + l]); bad++; }
    printf("mismatches %d\n", bad);
    return bad != 0;
}
