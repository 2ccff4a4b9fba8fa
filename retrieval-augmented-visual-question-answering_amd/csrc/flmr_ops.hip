// Op-level entry points: 1:1 replacements of the reference's four pybind functions + the scoring head.
//   filter_pids_cpp           TPC/search/filter_pids.cpp:126-164
//   decompress_residuals_cpp  TPC/search/decompress_residuals.cpp:80-155 (CUDA twin: indexing/codecs/decompress_residuals.cu)
//   segmented_lookup_cpp      TPC/search/segmented_lookup.cpp:127-144
//   segmented_maxsim_cpp      TPC/modeling/segmented_maxsim.cpp:49-93
//   colbert_score (padded)    TPC/modeling/colbert.py:235-286
// These exist so each stage can be parity-tested in isolation and slotted behind the reference's class
// attributes (IndexScorer.filter_pids, ...).  The batched search path (flmr_search.hip) does not use them.
#include "flmr_device.h"

// RAII-less scratch helper: ops allocate small temporaries and free them after a stream sync.
struct scratch {
    void* p[12];
    int n = 0;
    ~scratch() { for (int i = 0; i < n; i++) (void)hipFree(p[i]); }
    template <typename T>
    int alloc(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) { snprintf(flmr_err_buf, sizeof(flmr_err_buf), "hipMalloc: %s", hipGetErrorString(e)); return FLMR_ERR_HIP; }
        p[n++] = q;
        *out = static_cast<T*>(q);
        return FLMR_OK;
    }
};
#define RUN(x)             \
    do {                   \
        int rc__ = (x);    \
        if (rc__) return rc__; \
    } while (0)

// ---- filter_pids -------------------------------------------------------------------------------
__global__ void pack_idx_bits_kernel(const uint8_t* idx, int K, uint32_t* bits, int words) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint32_t v = 0;
    for (int b = 0; b < 32; b++) {
        const int c = w * 32 + b;
        if (c < K && idx[c]) v |= 1u << b;
    }
    bits[w] = v;
}
__global__ void set_i32_kernel(int32_t* p, int32_t v) { *p = v; }

extern "C" int flmr_filter_pids(const int32_t* pids, int64_t npids, const float* cs, int32_t K, int32_t nq,
                                const int32_t* codes, const int64_t* doclens, const int64_t* offsets, const uint8_t* idx,
                                int32_t ndocs, int32_t* out_pids, int32_t* out_count, flmr_stream_t stream) {
    if (!pids || !cs || !codes || !offsets || !idx || !out_pids || !out_count) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (ndocs < 4 || ndocs > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "ndocs=%d (4..%d)", ndocs, FLMR_MAX_NDOCS);
    if (nq < 1 || nq > FLMR_MAX_NQ_CAND) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d (1..%d)", nq, FLMR_MAX_NQ_CAND);
    if (npids > 0x7fffffffLL) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "npids too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    scratch sc;
    const int words = (int)flmr_ceil_div(K, 32);
    uint32_t* bits; uint64_t *keys1, *keys2; int32_t *cnt, *s1, *n1;
    RUN(sc.alloc(&bits, words));
    RUN(sc.alloc(&keys1, (size_t)(npids > 0 ? npids : 1)));
    RUN(sc.alloc(&keys2, ndocs));
    RUN(sc.alloc(&cnt, 1));
    RUN(sc.alloc(&s1, ndocs));
    RUN(sc.alloc(&n1, 1));
    hipLaunchKernelGGL(pack_idx_bits_kernel, dim3((unsigned)flmr_ceil_div(words, 256)), dim3(256), 0, st, idx, K, bits, words);
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, st, cnt, (int32_t)npids);
    flmr_filter_args f{};
    f.cs = cs; f.cs_query_stride = 0; f.K = K; f.ncol = nq; f.nq_cand = nq; f.nqueries = 1; f.q_lens = nullptr;
    f.codes = codes; f.doclens = doclens; f.offsets = offsets;
    const int64_t stride = npids > 0 ? npids : 1;
    RUN(flmr_launch_filter_stage1(f, bits, words, pids, stride, cnt, keys1, nullptr, 0, nullptr, nullptr, st));
    RUN(flmr_launch_select_topn(keys1, stride, cnt, 1, ndocs, s1, ndocs, n1, st));
    RUN(flmr_launch_filter_stage2(f, s1, ndocs, n1, ndocs, keys2, ndocs, st));
    RUN(flmr_launch_sort_topn(keys2, ndocs, n1, ndocs, 1, ndocs / 4, out_pids, nullptr, ndocs / 4, out_count, 0, 0, st));
    FLMR_HIP(hipStreamSynchronize(st));
    return FLMR_OK;
}

// ---- decompress_residuals ------------------------------------------------------------------------
// one workgroup per document, one half-wave per token row, lane j owns dims 4j..4j+3
__global__ __launch_bounds__(256) void decompress_rows_kernel(const int32_t* pids, const int64_t* doclens,
                                                              const int64_t* offsets, const float* bucket_weights,
                                                              const uint8_t* rev, const uint8_t* combos,
                                                              const uint8_t* residuals, const int32_t* codes,
                                                              const float* centroids, int dim, int nbits,
                                                              const int64_t* out_row_offsets, float* out,
                                                              int64_t out_capacity_rows) {
    const int pid = pids[blockIdx.x];
    const int64_t off = offsets[pid];
    const int len = (int)(doclens ? doclens[pid] : (offsets[pid + 1] - offsets[pid]));
    const int64_t orow = out_row_offsets[blockIdx.x];
    const int vpb = 8 / nbits, packed = dim / vpb;
    const int hw = threadIdx.x >> 5, j = threadIdx.x & 31;
    for (int t = hw; t < len; t += 8) {
        if (orow + t >= out_capacity_rows) return;
        const int code = codes[off + t];
        const uint8_t* rb = residuals + (size_t)(off + t) * packed;
        for (int d0 = 4 * j; d0 < dim; d0 += 128) {
            float4 v;
            float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int d = d0 + e;
                const int x = rev[rb[d / vpb]];
                vp[e] = bucket_weights[combos[x * vpb + (d % vpb)]] + centroids[(size_t)code * dim + d];
            }
            *reinterpret_cast<float4*>(out + (size_t)(orow + t) * dim + d0) = v;
        }
    }
}

extern "C" int flmr_decompress_residuals(const int32_t* pids, int32_t npids, const int64_t* doclens, const int64_t* offsets,
                                         const float* bucket_weights, const uint8_t* rev, const uint8_t* combos,
                                         const uint8_t* residuals, const int32_t* codes, const float* centroids,
                                         int32_t dim, int32_t nbits, float* out, int64_t out_capacity_rows,
                                         int64_t* out_row_offsets, flmr_stream_t stream) {
    if (!pids || !offsets || !bucket_weights || !rev || !combos || !residuals || !codes || !centroids || !out)
        FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (dim % 128 != 0) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dim=%d must be a multiple of 128", dim);
    if (nbits != 1 && nbits != 2 && nbits != 4 && nbits != 8) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d", nbits);
    if (npids <= 0) return FLMR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    scratch sc;
    int64_t* ro = out_row_offsets;
    if (!ro) RUN(sc.alloc(&ro, (size_t)npids + 1));
    RUN(flmr_launch_exclusive_scan_lengths(pids, doclens, offsets, npids, ro, st));
    hipLaunchKernelGGL(decompress_rows_kernel, dim3(npids), dim3(256), 0, st, pids, doclens, offsets, bucket_weights, rev,
                       combos, residuals, codes, centroids, dim, nbits, ro, out, out_capacity_rows);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipStreamSynchronize(st));
    return FLMR_OK;
}

// ---- segmented_lookup ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segmented_copy_kernel(const uint8_t* in, int64_t row_bytes, const int64_t* lengths,
                                                             const int64_t* offsets, const int64_t* out_row_offsets,
                                                             uint8_t* out, int64_t out_capacity_rows) {
    const int i = blockIdx.x;
    const int64_t nbytes = lengths[i] * row_bytes;
    if (out_row_offsets[i] + lengths[i] > out_capacity_rows) return;
    const uint8_t* src = in + offsets[i] * row_bytes;
    uint8_t* dst = out + out_row_offsets[i] * row_bytes;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)nbytes) & 3) == 0) {
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
        uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
        for (int64_t e = threadIdx.x; e < nbytes / 4; e += blockDim.x) d4[e] = s4[e];
    } else {
        for (int64_t e = threadIdx.x; e < nbytes; e += blockDim.x) dst[e] = src[e];
    }
}

extern "C" int flmr_segmented_lookup(const void* input, int64_t row_bytes, const int64_t* lengths, const int64_t* offsets,
                                     int32_t nseg, void* out, int64_t out_capacity_rows, int64_t* out_row_offsets,
                                     flmr_stream_t stream) {
    if (!input || !lengths || !offsets || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (nseg <= 0) return FLMR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    scratch sc;
    int64_t* ro = out_row_offsets;
    if (!ro) RUN(sc.alloc(&ro, (size_t)nseg + 1));
    RUN(flmr_launch_exclusive_scan_lengths(nullptr, lengths, nullptr, nseg, ro, st));
    hipLaunchKernelGGL(segmented_copy_kernel, dim3(nseg), dim3(256), 0, st, static_cast<const uint8_t*>(input), row_bytes,
                       lengths, offsets, ro, static_cast<uint8_t*>(out), out_capacity_rows);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipStreamSynchronize(st));
    return FLMR_OK;
}

// ---- segmented_maxsim ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segmented_maxsim_kernel(const float* scores, const int64_t* row_offsets, int nq,
                                                               float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* colmax = reinterpret_cast<float*>(smem);
    const int i = blockIdx.x;
    const int64_t r0 = row_offsets[i], r1 = row_offsets[i + 1];
    for (int k = threadIdx.x; k < nq; k += blockDim.x) {
        float m = 0.0f;  // zero-initialised running max (segmented_maxsim.cpp:58-59)
        for (int64_t r = r0; r < r1; r++) m = fmaxf(m, scores[(size_t)r * nq + k]);
        colmax[k] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[i] = flmr_seq_sum(colmax, nq);
}

extern "C" int flmr_segmented_maxsim(const float* scores, const int64_t* lengths, int32_t ndocs, int32_t nq, float* out,
                                     flmr_stream_t stream) {
    if (!scores || !lengths || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (ndocs <= 0) return FLMR_OK;
    if ((size_t)nq * 4 > 64 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large", nq);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    scratch sc;
    int64_t* ro;
    RUN(sc.alloc(&ro, (size_t)ndocs + 1));
    RUN(flmr_launch_exclusive_scan_lengths(nullptr, lengths, nullptr, ndocs, ro, st));
    hipLaunchKernelGGL(segmented_maxsim_kernel, dim3(ndocs), dim3(256), (size_t)nq * 4, st, scores, ro, nq, out);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipStreamSynchronize(st));
    return FLMR_OK;
}

// ---- fused decompress + normalise + MaxSim for an explicit pid list ----------------------------------
extern "C" int flmr_score_pids(const flmr_index_t* ix, const float* Q, int32_t nq, const int32_t* pids, int32_t npids,
                               float* out, flmr_stream_t stream) {
    if (!ix || !Q || !pids || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (npids <= 0) return FLMR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    scratch sc;
    int32_t* cnt;
    RUN(sc.alloc(&cnt, 1));
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, st, cnt, npids);
    flmr_maxsim_args m{};
    m.ix = ix; m.Q = Q; m.q_lens = nullptr; m.nqueries = 1; m.nq = nq; m.pids = pids; m.pid_stride = npids;
    m.counts = cnt; m.max_count = npids; m.keys = nullptr; m.key_stride = npids; m.scores = out;
    m.q_hi = nullptr; m.q_lo = nullptr;
    _Float16 *qh = nullptr, *ql = nullptr;
    if (ix->centroids_f16_exact) {
        RUN(sc.alloc(&qh, (size_t)flmr_round_up(nq, 32) * FLMR_DIM));
        RUN(sc.alloc(&ql, (size_t)flmr_round_up(nq, 32) * FLMR_DIM));
        m.q_hi = qh; m.q_lo = ql;
    }
    if (ix->centroids_f16_exact && ix->max_doclen > 0) {   // workspace of the planned-tile kernels (optional)
        const int64_t stride = (int64_t)npids * ((ix->max_doclen + 31) / 32);
        uint2* desc = nullptr;
        int32_t* wbeg = nullptr;
        if (stride * (int64_t)sizeof(uint2) <= ((int64_t)1 << 30) && sc.alloc(&desc, (size_t)stride) == FLMR_OK &&
            sc.alloc(&wbeg, (size_t)npids + 8) == FLMR_OK) {
            m.plan_desc = desc; m.plan_stride = stride; m.plan_wbeg = wbeg; m.plan_wcap = npids + 8;
            const int64_t cmf = (int64_t)npids * flmr_round_up(nq, 32) + 1;
            float* cm = nullptr;
            if (nq > 32 && cmf * (int64_t)sizeof(float) <= ((int64_t)2 << 30) && sc.alloc(&cm, (size_t)cmf) == FLMR_OK) {
                m.colmax_ws = cm; m.colmax_cap = cmf;
            }
        }
    }
    RUN(flmr_launch_maxsim(m, st));
    FLMR_HIP(hipStreamSynchronize(st));
    return FLMR_OK;
}

// ---- colbert_score, padded variant (-9999 padding, no clamp) ------------------------------------------
__global__ __launch_bounds__(256) void colbert_score_padded_kernel(const float* Q, int q_batch, int nq, const float* D,
                                                                   const uint8_t* mask, int Ld, int dim, float* out, float* colmax_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* colmax = reinterpret_cast<unsigned int*>(smem);  // order-preserving uint image of the fp32 max
    const int b = blockIdx.x;
    // gridDim.y > 1: the CROSS form -- every query (blockIdx.y) against every document, out [gridDim.y, B]
    const size_t ob = (size_t)blockIdx.y * gridDim.x + b;
    const float* Qb = Q + (gridDim.y > 1 ? (size_t)blockIdx.y * nq * dim : (q_batch == 1 ? 0 : (size_t)b * nq * dim));
    for (int k = threadIdx.x; k < nq; k += blockDim.x) colmax[k] = 0u;  // below every real value
    __syncthreads();
    for (int e = threadIdx.x; e < Ld * nq; e += blockDim.x) {
        const int t = e / nq, j = e % nq;
        float v = -9999.0f;
        if (mask[(size_t)b * Ld + t]) {
            const float* dr = D + ((size_t)b * Ld + t) * dim;
            const float* qr = Qb + (size_t)j * dim;
            v = 0.0f;
            for (int k = 0; k < dim; k++) v = fmaf(dr[k], qr[k], v);
        }
        atomicMax(&colmax[j], flmr_f2ord(v));
    }
    __syncthreads();
    if (colmax_out)   // the per-column maxima themselves ('flipr' sums the largest of them, colbert.py:246-261)
        for (int k = threadIdx.x; k < nq; k += blockDim.x) colmax_out[ob * nq + k] = (Ld > 0) ? flmr_ord2f(colmax[k]) : FLMR_NEG_INF;
    if (threadIdx.x == 0 && out) {
        float s = 0.0f;
        for (int k = 0; k < nq; k++) s += (Ld > 0) ? flmr_ord2f(colmax[k]) : FLMR_NEG_INF;
        out[ob] = s;
    }
}

// MFMA variant for dim == 128 (the FLMR / ColBERT head): one workgroup per document, the 4 waves walk the document's
// 32-token tiles; both operands are split into fp16 hi/lo on the fly (D rows per tile, Q once per call by
// score_split_q) and contracted with three v_mfma_f32_32x32x16_f16 per 16 dims (hi.hi + 2^-11 (hi.lo + lo.hi), fp32
// accumulation, ~2^-21 relative error per term).  Query chunks of 128 rows are staged in LDS and shared by the waves.
typedef _Float16 ph8 __attribute__((ext_vector_type(8)));
#define PS_QC 4
#define PS_BROW 136

__global__ __launch_bounds__(256) void score_split_q(const float* Q, int rows, int nq, int nqp, _Float16* q_hi, _Float16* q_lo) {
    // Q [q_batch, nq, 128] -> hi/lo [q_batch, nqp, 128], rows >= nq zero
    const size_t total = (size_t)rows * nqp * FLMR_DIM;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t qb = e / ((size_t)nqp * FLMR_DIM);
        const int r = (int)((e / FLMR_DIM) % nqp), d = (int)(e % FLMR_DIM);
        const float v = r < nq ? Q[(qb * nq + r) * FLMR_DIM + d] : 0.0f;
        const _Float16 hi = (_Float16)v;
        q_hi[e] = hi;
        q_lo[e] = (_Float16)((v - (float)hi) * 2048.0f);
    }
}

__global__ __launch_bounds__(256, 2) void colbert_score_padded_mfma_kernel(const _Float16* __restrict__ q_hi,
                                                                           const _Float16* __restrict__ q_lo, int q_batch,
                                                                           int nq, int nqp, const float* __restrict__ D,
                                                                           const uint8_t* __restrict__ mask, int Ld,
                                                                           float* out, float* colmax_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* bq = reinterpret_cast<_Float16*>(smem);                                          // [PS_QC][hi|lo][32][PS_BROW]
    unsigned int* colmax = reinterpret_cast<unsigned int*>(bq + PS_QC * 2 * 32 * PS_BROW);     // [nqp] order-preserving fp32 image
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // gridDim.y > 1: the CROSS form -- every query (blockIdx.y) against every document, out [gridDim.y, B]
    const size_t ob = (size_t)blockIdx.y * gridDim.x + b;
    const size_t qoff = (gridDim.y > 1 ? (size_t)blockIdx.y : (q_batch == 1 ? 0 : (size_t)b)) * nqp * FLMR_DIM;
    for (int k = tid; k < nqp; k += 256) colmax[k] = 0u;  // below every real value
    const int ntiles = (Ld + 31) >> 5;
    for (int qc0 = 0; qc0 < nq; qc0 += 32 * PS_QC) {
        const int ntq = ((nq - qc0 < 32 * PS_QC ? nq - qc0 : 32 * PS_QC) + 31) >> 5;
        __syncthreads();
        for (int e = tid; e < ntq * 1024; e += 256) {
            const int piece = e & 15, row = (e >> 4) & 31, hl = (e >> 9) & 1, qt = e >> 10;
            const _Float16* src = (hl ? q_lo : q_hi) + qoff + (size_t)(qc0 + qt * 32 + row) * FLMR_DIM + piece * 8;
            *reinterpret_cast<ph8*>(bq + ((qt * 2 + hl) * 32 + row) * PS_BROW + piece * 8) = *reinterpret_cast<const ph8*>(src);
        }
        __syncthreads();
        for (int t = wave; t < ntiles; t += 4) {
            const int tok = t * 32 + i;
            const bool inb = tok < Ld;
            ph8 ah[8], al[8];
            {
                const float4* p = reinterpret_cast<const float4*>(D + ((size_t)b * Ld + (inb ? tok : 0)) * FLMR_DIM + 64 * h);
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const float4 x = p[2 * s], y = p[2 * s + 1];
                    const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const _Float16 hi = (_Float16)v[e];
                        ah[s][e] = hi;
                        al[s][e] = (_Float16)((v[e] - (float)hi) * 2048.0f);
                    }
                }
            }
            // which of this lane's 16 accumulator rows are real, unmasked tokens (bit r)
            uint32_t okbits = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < Ld && mask[(size_t)b * Ld + row]) okbits |= 1u << r;
            }
            for (int qt = 0; qt < ntq; qt++) {
                f32x16 acch, accl, accm;
#pragma unroll
                for (int r = 0; r < 16; r++) { acch[r] = 0.0f; accl[r] = 0.0f; accm[r] = 0.0f; }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const ph8 bh = *reinterpret_cast<const ph8*>(bq + ((qt * 2 + 0) * 32 + i) * PS_BROW + 64 * h + 8 * s);
                    const ph8 bl = *reinterpret_cast<const ph8*>(bq + ((qt * 2 + 1) * 32 + i) * PS_BROW + 64 * h + 8 * s);
                    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acch, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, accl, 0, 0, 0);
                    accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, accm, 0, 0, 0);
                }
                float mx = FLMR_NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = ((okbits >> r) & 1u) ? fmaf(accl[r] + accm[r], 1.0f / 2048.0f, acch[r]) : -9999.0f;  // colbert.py:240
                    if (row < Ld) mx = fmaxf(mx, v);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const int col = qc0 + qt * 32 + i;
                if (h == 0 && col < nq && mx > FLMR_NEG_INF) atomicMax(&colmax[col], flmr_f2ord(mx));
            }
        }
    }
    __syncthreads();
    if (colmax_out)
        for (int k = tid; k < nq; k += 256) colmax_out[ob * nq + k] = (Ld > 0) ? flmr_ord2f(colmax[k]) : FLMR_NEG_INF;
    if (tid == 0 && out) {
        float sc = 0.0f;
        for (int k = 0; k < nq; k++) sc += (Ld > 0) ? flmr_ord2f(colmax[k]) : FLMR_NEG_INF;
        out[ob] = sc;
    }
}

// cross: Q holds q_batch independent queries and EVERY one of them is scored against every document (out [q_batch, B])
static int score_padded(const float* Q, int32_t q_batch, int32_t nq, const float* D, const uint8_t* mask,
                        int32_t B, int32_t Ld, int32_t dim, float* out, float* colmax_out, flmr_stream_t stream, bool cross = false) {
    if (!Q || !D || !mask || (!out && !colmax_out)) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (!cross && q_batch != 1 && q_batch != B) FLMR_FAIL(FLMR_ERR_INVALID, "q_batch must be 1 or B");
    if (cross && (q_batch < 1 || q_batch > 65535)) FLMR_FAIL(FLMR_ERR_INVALID, "cross scoring takes 1..65535 queries per call");
    if (B <= 0) return FLMR_OK;
    const dim3 grid((unsigned)B, cross ? (unsigned)q_batch : 1u);
    if (cross && q_batch == 1) cross = false;   // (the broadcast form)
    if ((size_t)nq * 4 > 48 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large", nq);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dim == FLMR_DIM && !flmr_opts().is(FLMR_OPT_SCORE_IMPL, "valu")) {
        scratch sc;
        const int nqp = (int)flmr_round_up(nq, 32);
        _Float16 *qh, *ql;
        RUN(sc.alloc(&qh, (size_t)q_batch * nqp * FLMR_DIM));
        RUN(sc.alloc(&ql, (size_t)q_batch * nqp * FLMR_DIM));
        hipLaunchKernelGGL(score_split_q, dim3(1024), dim3(256), 0, st, Q, q_batch, nq, nqp, qh, ql);
        const size_t lds = (size_t)PS_QC * 2 * 32 * PS_BROW * sizeof(_Float16) + (size_t)nqp * sizeof(unsigned int);
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(colbert_score_padded_mfma_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(colbert_score_padded_mfma_kernel, grid, dim3(256), lds, st, qh, ql, q_batch, nq, nqp, D, mask, Ld, out, colmax_out);
        FLMR_LAUNCH_CHECK();
        FLMR_HIP(hipStreamSynchronize(st));  // the split buffers are released on return
        return FLMR_OK;
    }
    hipLaunchKernelGGL(colbert_score_padded_kernel, grid, dim3(256), (size_t)nq * 4, st, Q, q_batch, nq, D, mask, Ld, dim,
                       out, colmax_out);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

extern "C" int flmr_colbert_score_padded(const float* Q, int32_t q_batch, int32_t nq, const float* D, const uint8_t* mask,
                                         int32_t B, int32_t Ld, int32_t dim, float* out, flmr_stream_t stream) {
    if (!out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    return score_padded(Q, q_batch, nq, D, mask, B, Ld, dim, out, nullptr, stream);
}

// every one of `nqueries` queries against every document: out f32 [nqueries, B] -- the rate matrix of the executor's exhaustive
// search (src/executors/FLMR_executor.py:799-847 fills it four documents at a time through model.score)
extern "C" int flmr_colbert_score_cross(const float* Q, int32_t nqueries, int32_t nq, const float* D, const uint8_t* mask,
                                        int32_t B, int32_t Ld, int32_t dim, float* out, flmr_stream_t stream) {
    if (!out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    return score_padded(Q, nqueries, nq, D, mask, B, Ld, dim, out, nullptr, stream, true);
}

// the per-column maxima [B, nq] before their sum: what colbert_score_reduce's 'flipr' interaction reduces (colbert.py:246-261)
extern "C" int flmr_colbert_colmax_padded(const float* Q, int32_t q_batch, int32_t nq, const float* D, const uint8_t* mask,
                                          int32_t B, int32_t Ld, int32_t dim, float* out_colmax, flmr_stream_t stream) {
    if (!out_colmax) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    return score_padded(Q, q_batch, nq, D, mask, B, Ld, dim, nullptr, out_colmax, stream);
}

// ---- merge of per-shard top-k lists (after the RCCL all-gather) -----------------------------------------
__global__ __launch_bounds__(1024) void merge_topk_kernel(const float* scores, const int32_t* pids, int nshards,
                                                          int nqueries, int k, int npow2, float* out_scores,
                                                          int32_t* out_pids, int32_t* out_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    const int q = blockIdx.x;
    for (int e = threadIdx.x; e < npow2; e += blockDim.x) {
        unsigned long long key = 0ull;
        if (e < nshards * k) {
            const int sh = e / k, i = e % k;
            const size_t src = ((size_t)sh * nqueries + q) * k + i;
            if (pids[src] >= 0) key = flmr_make_key(scores[src], pids[src]);
        }
        s[e] = key;
    }
    __syncthreads();
    flmr_bitonic_sort_desc<unsigned long long>(s, npow2);
    int cnt = 0;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const unsigned long long key = s[i];
        const bool ok = key != 0ull;
        out_pids[(size_t)q * k + i] = ok ? flmr_key_pid(key) : -1;
        out_scores[(size_t)q * k + i] = ok ? flmr_key_score(key) : 0.0f;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < k; i++) cnt += (s[i] != 0ull);
        out_counts[q] = cnt;
    }
}

extern "C" int flmr_merge_topk(const float* scores, const int32_t* pids, int32_t nshards, int32_t nqueries, int32_t k,
                               float* out_scores, int32_t* out_pids, int32_t* out_counts, flmr_stream_t stream) {
    if (!scores || !pids || !out_scores || !out_pids || !out_counts) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (nshards < 1 || nqueries < 1 || k < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    if ((int64_t)nshards * k > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nshards*k=%d > %d", nshards * k, FLMR_MAX_NDOCS);
    int npow2 = 2;
    while (npow2 < nshards * k || npow2 < k) npow2 <<= 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nqueries), dim3(1024), (size_t)npow2 * 8, st, scores, pids, nshards, nqueries,
                       k, npow2, out_scores, out_pids, out_counts);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
