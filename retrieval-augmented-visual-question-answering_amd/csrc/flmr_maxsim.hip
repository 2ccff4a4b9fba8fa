// Stage 3: exact late-interaction score of the finalists, fused end to end.
//
// Reference (CPU path): TPC/search/index_storage.py:160-177 --
//   decompress_residuals_cpp (TPC/search/decompress_residuals.cpp:27-78)  D[t,:] = weight + centroid
//   F.normalize(D, p=2, dim=-1)                                            D[t,:] / max(||D[t,:]||, 1e-12)
//   colbert_score_packed (TPC/modeling/colbert.py:289-311)                 S = D @ Q^T (all nq rows of Q)
//   segmented_maxsim_cpp (TPC/modeling/segmented_maxsim.cpp:22-93)         per doc: max over tokens, INIT 0, sum over nq
//
// MI355X design: one workgroup per (query, finalist document); nothing intermediate touches HBM.
//   * each lane decompresses HALF a token row (64 dims) straight into registers: 8*nbits residual bytes,
//     a 256-byte run of the centroid row, and the fused 256-entry byte->weights table in LDS (4 KB at nbits=2);
//     that register image IS the A operand of v_mfma_f32_32x32x2_f32 (lane = (token, k-half)), so the
//     32 tokens x 32 query-tokens x 128 contraction needs no LDS staging of D at all;
//   * Q rows are the B operand (lane = (query token, k-half)), read from L2 (16 KB per 32 query tokens,
//     shared by all of a query's workgroups, which sit on one XCD because blockIdx.x = query);
//   * L2 normalisation is applied to the 32x32 score tile (one multiply per score by 1/max(||d||,eps))
//     instead of to the 128-wide row; the per-token max starts at 0 exactly as the CPU extension does, so
//     column maxima are non-negative and an integer LDS atomicMax on the float bits combines the 4 waves;
//   * the final sum over query tokens is k-ascending fp32 (the oracle's order; torch's own reduction order
//     is not defined, hence the 1e-4 score tolerance of the parity tests).
#include "flmr_device.h"

template <int NBITS>
__device__ __forceinline__ void decompress_half_row(const uint8_t* __restrict__ res /* 8*NBITS bytes */,
                                                    const float* __restrict__ cen /* 64 floats */,
                                                    const float* wlut /* LDS [256][8/NBITS] */, float* a /* 64 */) {
    constexpr int VPB = 8 / NBITS;
    constexpr int NB = 8 * NBITS;  // bytes covering 64 dims
    const uint2* r2 = reinterpret_cast<const uint2*>(res);
    const float4* c4 = reinterpret_cast<const float4*>(cen);
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const float4 v = c4[t];
        a[4 * t + 0] = v.x; a[4 * t + 1] = v.y; a[4 * t + 2] = v.z; a[4 * t + 3] = v.w;
    }
#pragma unroll
    for (int w = 0; w < NB / 8; w++) {
        const uint2 pk = r2[w];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t word = e < 4 ? pk.x : pk.y;
            const uint32_t byte = (word >> (8 * (e & 3))) & 255u;
            const int kb = w * 8 + e;
#pragma unroll
            for (int l = 0; l < VPB; l++) a[kb * VPB + l] = wlut[byte * VPB + l] + a[kb * VPB + l];
        }
    }
}

template <int NBITS>
__global__ __launch_bounds__(256) void maxsim_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                     const uint8_t* __restrict__ residuals,
                                                     const int64_t* __restrict__ doc_offsets,
                                                     const float* __restrict__ centroids,
                                                     const float* __restrict__ wlut_g, int y_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    constexpr int PACKED = FLMR_DIM / VPB;
    float* wlut = reinterpret_cast<float*>(smem);                       // [256 * VPB]
    int* colmax = reinterpret_cast<int*>(smem + 256 * VPB * sizeof(float));  // [nq]
    const int b = blockIdx.x, d = y_base + blockIdx.y;
    if (d >= m.counts[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    const int pid = m.pids[(size_t)b * m.pid_stride + d];
    const int64_t off = doc_offsets[pid];
    const int len = (int)(doc_offsets[pid + 1] - off);
    for (int t = tid; t < 256 * VPB; t += 256) wlut[t] = wlut_g[t];
    for (int t = tid; t < qlen; t += 256) colmax[t] = 0;  // +0.0f
    __syncthreads();
    const float* Qb = m.Q + (size_t)b * m.nq * FLMR_DIM;

    for (int t0 = wave * 32; t0 < len; t0 += 128) {
        const int tok = t0 + i;
        const bool valid = tok < len;
        float a[64];
        if (valid) {
            const int code = codes[off + tok];
            decompress_half_row<NBITS>(residuals + (size_t)(off + tok) * PACKED + h * (PACKED / 2),
                                       centroids + (size_t)code * FLMR_DIM + 64 * h, wlut, a);
        } else {
#pragma unroll
            for (int t = 0; t < 64; t++) a[t] = 0.0f;
        }
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 64; t++) ss = fmaf(a[t], a[t], ss);
        ss += __shfl_xor(ss, 32, 64);
        float nrm = sqrtf(ss);
        nrm = nrm < 1e-12f ? 1e-12f : nrm;
        const float inv = 1.0f / nrm;

        for (int q0 = 0; q0 < qlen; q0 += 32) {
            const int col = q0 + i;
            float bv[64];
            if (col < qlen) {
                const float4* p = reinterpret_cast<const float4*>(Qb + (size_t)col * FLMR_DIM + 64 * h);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float4 v = p[t];
                    bv[4 * t + 0] = v.x; bv[4 * t + 1] = v.y; bv[4 * t + 2] = v.z; bv[4 * t + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 64; t++) bv[t] = 0.0f;
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 64; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[s], acc, 0, 0, 0);
            float mx = 0.0f;  // segmented_maxsim.cpp:58-59: running max starts at zero
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // token row of this accumulator register
                const float inv_r = __shfl(inv, row, 64);
                mx = fmaxf(mx, acc[r] * inv_r);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (h == 0 && col < qlen) atomicMax(&colmax[col], __float_as_int(mx));
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int k = 0; k < qlen; k++) s += __int_as_float(colmax[k]);
        if (m.keys) m.keys[(size_t)b * m.key_stride + d] = flmr_make_key(s, pid);
        if (m.scores) m.scores[(size_t)b * m.key_stride + d] = s;
    }
}

template <int NBITS>
static int launch_maxsim_t(const flmr_maxsim_args& a, hipStream_t st) {
    const flmr_index* ix = a.ix;
    const size_t lds = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)a.nq * sizeof(int);
    if (lds > 64 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large for the MaxSim kernel's LDS column maxima", a.nq);
    for (int y0 = 0; y0 < a.max_count; y0 += 32768) {
        const int ny = (a.max_count - y0) < 32768 ? (a.max_count - y0) : 32768;
        hipLaunchKernelGGL(maxsim_kernel<NBITS>, dim3(a.nqueries, ny), dim3(256), lds, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids, ix->wlut, y0);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_maxsim(const flmr_maxsim_args& a, hipStream_t st) {
    if (a.max_count <= 0) return FLMR_OK;
    switch (a.ix->nbits) {
        case 1: return launch_maxsim_t<1>(a, st);
        case 2: return launch_maxsim_t<2>(a, st);
        case 4: return launch_maxsim_t<4>(a, st);
        case 8: return launch_maxsim_t<8>(a, st);
    }
    FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d", a.ix->nbits);
}

// ------------------------------------------------------------------------------------------------
// exclusive prefix of doc lengths for a pid list (used by the op-level decompress / lookup entry points)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_lengths_kernel(const int32_t* pids, const int64_t* doclens,
                                                            const int64_t* offsets, int32_t n, int64_t* out) {
    __shared__ int scan_lds[17];
    int64_t base = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int len = 0;
        if (i < n) {
            const int p = pids ? pids[i] : i;
            len = (int)(doclens ? doclens[p] : (offsets[p + 1] - offsets[p]));
        }
        int total;
        const int ex = flmr_block_exclusive_scan(len, scan_lds, &total);
        if (i < n) out[i] = base + ex;
        base += total;
    }
    if (threadIdx.x == 0) out[n] = base;
}

int flmr_launch_exclusive_scan_lengths(const int32_t* pids, const int64_t* doclens, const int64_t* offsets, int32_t n,
                                       int64_t* out_offsets, hipStream_t st) {
    hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, st, pids, doclens, offsets, n, out_offsets);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
