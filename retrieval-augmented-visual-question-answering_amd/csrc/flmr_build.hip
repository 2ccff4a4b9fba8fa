// Index build kernels: what the reference does in ResidualCodec.compress / binarize / packbits
// (colbert/indexing/codecs/residual.py:169-222) right before the search path can run.
//   flmr_nearest_centroids   codes = argmax_c  centroids[c] . emb[t]        (residual.py:206-216, compress_into_codes)
//   flmr_compress_residuals  residual = emb - centroids[code]; bucketize against the cut-offs; emit the nbits of every
//                            bucket index LSB first and pack MSB first (residual.py:186-204: the `>> arange(nbits) & 1`
//                            expansion followed by np.packbits), i.e. each nbits group is bit-reversed inside its byte.
// The argmax runs on the stage-0 fp16-split MFMA kernel (flmr_stage0.hip, ARGMAX instantiation): the index stores its
// centroids as half (residual.py:161) so the split product is within one fp32 rounding of the exact dot product.
#include <hip/hip_runtime.h>

#include "flmr_common.h"
#include "flmr_device.h"

namespace {

__global__ void build_qlens_kernel(int32_t* q_lens, int nqueries, int64_t remaining) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nqueries) return;
    const int64_t left = remaining - (int64_t)b * 32;
    q_lens[b] = left >= 32 ? 32 : (left > 0 ? (int)left : 0);
}

// rows K..Kpad-1 := row 0 (an exact duplicate can never win the first-index argmax)
__global__ void pad_centroids_kernel(const float* src, int64_t K, int64_t Kpad, float* dst) {
    const size_t total = (size_t)Kpad * FLMR_DIM;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / FLMR_DIM;
        dst[e] = row < (size_t)K ? src[e] : src[e % FLMR_DIM];
    }
}

template <int NBITS>
__global__ __launch_bounds__(256) void compress_residuals_kernel(const float* __restrict__ cen, const float* __restrict__ emb,
                                                                 const int32_t* __restrict__ codes, int64_t n,
                                                                 const float* __restrict__ cutoffs,
                                                                 uint8_t* __restrict__ out) {
    constexpr int VPB = 8 / NBITS, NCUT = (1 << NBITS) - 1, BPT = FLMR_DIM / VPB;
    __shared__ float cut[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) cut[t] = t < NCUT ? cutoffs[t] : __builtin_inff();
    __syncthreads();
    const size_t total = (size_t)n * BPT;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t tok = e / BPT;
        const int d0 = (int)(e % BPT) * VPB;
        const float* x = emb + tok * FLMR_DIM + d0;
        const float* c = cen + (size_t)codes[tok] * FLMR_DIM + d0;
        uint32_t byte = 0;
#pragma unroll
        for (int l = 0; l < VPB; l++) {
            const float r = x[l] - c[l];
            // torch.bucketize(r, cutoffs) (right=False) = number of cut-offs strictly below r
            int bucket;
            if (NBITS <= 2) {
                bucket = 0;
#pragma unroll
                for (int j = 0; j < NCUT; j++) bucket += cut[j] < r ? 1 : 0;
            } else {
                int lo = 0, hi = NCUT;  // first index whose cut-off is >= r
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cut[mid] < r) lo = mid + 1; else hi = mid;
                }
                bucket = lo;
            }
            const uint32_t rev = __brev((uint32_t)bucket) >> (32 - NBITS);
            byte |= rev << (8 - NBITS * (l + 1));
        }
        out[e] = (uint8_t)byte;
    }
}

struct dev_buf {
    void* p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        FLMR_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return FLMR_OK;
    }
};

}  // namespace

#define RUN(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

extern "C" int flmr_nearest_centroids(const float* centroids, int64_t K, const float* emb, int64_t n, int32_t* out_codes,
                                      flmr_stream_t stream) {
    if (!centroids || !emb || !out_codes) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (K < 1 || K > (1ll << 30) || n < 0) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes K=%lld n=%lld", (long long)K, (long long)n);
    if (n == 0) return FLMR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FLMR_HIP(hipStreamSynchronize(st));  // the checks below run on the default stream
    int32_t exact = 0;
    RUN(flmr_check_f16_exact(centroids, (size_t)K * FLMR_DIM, &exact));
    if (!exact) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "centroids must be fp16-representable (the index stores them as half, residual.py:161): round them with .half().float() first");
    const int64_t Kpad = flmr_round_up(K, 64);
    dev_buf cpad;
    const float* cen = centroids;
    if (Kpad != K) {
        RUN(cpad.alloc((size_t)Kpad * FLMR_DIM * sizeof(float)));
        hipLaunchKernelGGL(pad_centroids_kernel, dim3(1024), dim3(256), 0, st, centroids, K, Kpad, static_cast<float*>(cpad.p));
        cen = static_cast<const float*>(cpad.p);
    }
    const int64_t nblk = Kpad / 64, idx_words = Kpad / 32;
    const size_t per_query = (size_t)2 * 32 * FLMR_DIM * sizeof(_Float16) + (size_t)nblk * 32 * 8 + (size_t)idx_words * 4 + 4;
    int64_t chunk = (int64_t)((1ull << 30) / per_query);
    chunk = chunk < 8 ? 8 : (chunk > 4096 ? 4096 : chunk);
    const int64_t nqueries = (n + 31) / 32;
    if (chunk > nqueries) chunk = nqueries;
    dev_buf qh, ql, pv, pi, ib, qlen;
    RUN(qh.alloc((size_t)chunk * 32 * FLMR_DIM * sizeof(_Float16)));
    RUN(ql.alloc((size_t)chunk * 32 * FLMR_DIM * sizeof(_Float16)));
    RUN(pv.alloc((size_t)chunk * nblk * 32 * sizeof(float)));
    RUN(pi.alloc((size_t)chunk * nblk * 32 * sizeof(int32_t)));
    RUN(ib.alloc((size_t)chunk * idx_words * sizeof(uint32_t)));
    RUN(qlen.alloc((size_t)chunk * sizeof(int32_t)));
    for (int64_t q0 = 0; q0 < nqueries; q0 += chunk) {
        const int nqb = (int)((nqueries - q0) < chunk ? (nqueries - q0) : chunk);
        hipLaunchKernelGGL(build_qlens_kernel, dim3((nqb + 255) / 256), dim3(256), 0, st, static_cast<int32_t*>(qlen.p), nqb,
                           n - q0 * 32);
        flmr_s0_args a{};
        a.centroids = cen; a.Q = emb + (size_t)q0 * 32 * FLMR_DIM; a.q_lens = static_cast<const int32_t*>(qlen.p);
        a.K = (int32_t)Kpad; a.nqueries = nqb; a.nq = 32; a.nq_cand = 32; a.ncol = 32; a.ncells = 1;
        a.cs = nullptr; a.idx_bits = static_cast<uint32_t*>(ib.p); a.idx_words = (int32_t)idx_words;
        a.part_val = static_cast<float*>(pv.p); a.part_idx = static_cast<int32_t*>(pi.p);
        a.cells = nullptr; a.ncell = nullptr; a.max_cells = 0;
        a.centroids_f16 = nullptr; a.q_hi = static_cast<_Float16*>(qh.p); a.q_lo = static_cast<_Float16*>(ql.p);
        a.centroids_f16_exact = 1;
        RUN(flmr_launch_centroid_argmax(a, out_codes + (size_t)q0 * 32, st));
    }
    FLMR_HIP(hipStreamSynchronize(st));  // the scratch buffers are freed on return
    return FLMR_OK;
}

extern "C" int flmr_compress_residuals(const float* centroids, int64_t K, const float* emb, const int32_t* codes, int64_t n,
                                       const float* bucket_cutoffs, int32_t nbits, uint8_t* out_residuals,
                                       flmr_stream_t stream) {
    if (!centroids || !emb || !codes || !bucket_cutoffs || !out_residuals) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (K < 1 || n < 0) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    if (n == 0) return FLMR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)n * (FLMR_DIM * nbits / 8);
    const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
    switch (nbits) {
        case 1: hipLaunchKernelGGL(compress_residuals_kernel<1>, dim3(grid), dim3(256), 0, st, centroids, emb, codes, n, bucket_cutoffs, out_residuals); break;
        case 2: hipLaunchKernelGGL(compress_residuals_kernel<2>, dim3(grid), dim3(256), 0, st, centroids, emb, codes, n, bucket_cutoffs, out_residuals); break;
        case 4: hipLaunchKernelGGL(compress_residuals_kernel<4>, dim3(grid), dim3(256), 0, st, centroids, emb, codes, n, bucket_cutoffs, out_residuals); break;
        case 8: hipLaunchKernelGGL(compress_residuals_kernel<8>, dim3(grid), dim3(256), 0, st, centroids, emb, codes, n, bucket_cutoffs, out_residuals); break;
        default: FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d (supported: 1, 2, 4, 8)", nbits);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
#undef RUN
