// Stages 1 and 2: centroid-only MaxSim pruning + (score,pid) top-n selection.
//
// Reference: TPC/search/filter_pids.cpp -- maxsim() :27-69, filter_pids_helper() :71-124, filter_pids() :126-164.
//   per doc:   per_tok[k] = max over tokens whose code has idx[code] of centroid_scores[code,k] (init -9999)
//              score = sum_k per_tok[k], accumulated k-ascending in fp32 (:59-63)
//   selection: std::priority_queue<std::pair<float,int>> = descending lexicographic (score, pid)
//
// MI355X design (HBM-bound integer/gather work; no GEMM reshaping):
//   * one wave per candidate document: the doc's code run (doclen x 4 B, contiguous) is read with 64-lane
//     coalesced loads; the K-bit `idx` mask of the query lives in LDS (16 KB at K=131072), so the common case
//     "no token of this doc hits a surviving centroid" costs one LDS probe per token and no further HBM traffic;
//   * hits are broadcast with ballot/readlane and the 128-byte score row of the hit centroid is read by a
//     half-wave (lane k <-> query token k), two hits per iteration;
//   * the per-doc k-ascending fp32 sum is done by transposing 32 docs x 32 columns through LDS so that 32
//     lanes each run one doc's sequential sum: bit-identical to the CPU order at ~1 add per doc per lane;
//   * selection = 64-bit keys (order-preserving score bits << 32 | pid) + LDS-histogram radix select
//     (stage 1, unordered survivors) or an in-LDS bitonic sort (stage 2 / final, ordered output);
//   * blockIdx.x = query: the dispatcher places block b on XCD b % 8, so all blocks of one query share one
//     L2 for its score-table rows and idx mask.
#include "flmr_device.h"

// column-tile loop with a compile-time trip count so the per-tile accumulators stay in registers
#define FLMR_FOR_CT(ct, T) _Pragma("unroll") for (int ct = 0; ct < 4; ct++) if (ct < (T))

#define S1_WAVES 4
#define S1_GROUP 32  // docs transposed per wave before the sequential sums

__device__ __forceinline__ int64_t doc_len_of(const int64_t* doclens, const int64_t* offsets, int pid) {
    return doclens ? doclens[pid] : (offsets[pid + 1] - offsets[pid]);
}

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
    const int lo = __shfl((int)(uint32_t)v, src, 64);
    const int hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src, 64);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// lanes < nslots sum their doc's `nqc` column maxima k-ascending and emit the (score,pid) key
__device__ __forceinline__ void emit_group(const float* tr /* [S1_GROUP][ncolp] */, int ncolp, int nqc, int nslots,
                                           int lane, int my_pid, uint64_t* keys_out /* &keys[group base] */) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < nslots) {
        const float s = flmr_seq_sum(tr + lane * ncolp, nqc);
        keys_out[lane] = flmr_make_key(s, my_pid);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// fold the hits of one 64-token chunk into the running column maxima (lane k <-> column k, two hits / step)
__device__ __forceinline__ void s1_fold_hits(int code, const uint32_t* idxp, const float* cs, int ncol, int T, int nqc,
                                             int k, int h, float* per) {
    const bool hit = (code >= 0) && ((idxp[code >> 5] >> (code & 31)) & 1u);
    unsigned long long m = __ballot(hit);
    while (m) {  // wave-uniform
        const int sa = __builtin_ctzll(m);
        m &= m - 1;
        int sb = sa;
        if (m) { sb = __builtin_ctzll(m); m &= m - 1; }
        const int c = __shfl(code, h ? sb : sa, 64);
        const float* row = cs + (size_t)c * ncol;
        FLMR_FOR_CT(ct, T) if (ct * 32 + k < nqc) per[ct] = fmaxf(per[ct], row[ct * 32 + k]);
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 1.  grid = (nqueries, G), block = 256.  Dynamic LDS: transposes + idx words (if USE_LDS_IDX).
// A wave takes 32 consecutive candidates at a time: lanes fetch the 32 (pid, offset, length) triples in
// parallel, then the code runs of 4 docs are in flight together (2 x 64 tokens each) to cover HBM latency.
// ------------------------------------------------------------------------------------------------
template <bool USE_LDS_IDX>
__global__ __launch_bounds__(256) void filter_stage1_kernel(flmr_filter_args f, const uint32_t* idx_bits,
                                                            int32_t idx_words, const int32_t* cand,
                                                            int64_t cand_stride, const int32_t* cand_count,
                                                            uint64_t* keys) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = cand_count[b];
    const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
    const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;
    const int T = (f.nq_cand + 31) >> 5;  // column tiles of 32; f.ncol is the row stride of the score table
    const int ncolp = T * 32 + 1;
    float* tr = reinterpret_cast<float*>(smem) + (size_t)wave * S1_GROUP * ncolp;
    uint32_t* lidx = reinterpret_cast<uint32_t*>(smem + (size_t)S1_WAVES * S1_GROUP * ncolp * sizeof(float));
    const uint32_t* gidx = idx_bits + (size_t)b * idx_words;
    if (USE_LDS_IDX) {
        for (int w = threadIdx.x; w < idx_words; w += blockDim.x) lidx[w] = gidx[w];
        __syncthreads();
    }
    const uint32_t* idxp = USE_LDS_IDX ? lidx : gidx;
    const float* cs = f.cs + (size_t)b * f.cs_query_stride;
    const int32_t* cand_b = cand + (size_t)b * cand_stride;
    uint64_t* keys_b = keys + (size_t)b * cand_stride;

    const int waves_total = gridDim.y * S1_WAVES;
    const int wid = blockIdx.y * S1_WAVES + wave;
    const int k = lane & 31, h = lane >> 5;

    for (int g0 = wid * S1_GROUP; g0 < P; g0 += waves_total * S1_GROUP) {
        const int ndoc = (P - g0) < S1_GROUP ? (P - g0) : S1_GROUP;
        int my_pid = 0, my_len = 0;
        int64_t my_off = 0;
        if (lane < ndoc) {
            my_pid = cand_b[g0 + lane];
            my_off = f.offsets[my_pid];
            my_len = (int)doc_len_of(f.doclens, f.offsets, my_pid);
        }
        for (int j0 = 0; j0 < ndoc; j0 += 4) {
            int c0[4], c1[4], len[4];
            int64_t off[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = (j0 + u < ndoc) ? (j0 + u) : j0;
                off[u] = shfl_i64(my_off, j);
                len[u] = (j0 + u < ndoc) ? __shfl(my_len, j, 64) : 0;
                c0[u] = (lane < len[u]) ? f.codes[off[u] + lane] : -1;
                c1[u] = (lane + 64 < len[u]) ? f.codes[off[u] + 64 + lane] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j0 + u >= ndoc) break;  // wave-uniform
                float per[4] = {-9999.0f, -9999.0f, -9999.0f, -9999.0f};
                s1_fold_hits(c0[u], idxp, cs, f.ncol, T, nqc, k, h, per);
                if (len[u] > 64) s1_fold_hits(c1[u], idxp, cs, f.ncol, T, nqc, k, h, per);
                for (int t0 = 128; t0 < len[u]; t0 += 64) {  // long documents
                    const int t = t0 + lane;
                    const int code = (t < len[u]) ? f.codes[off[u] + t] : -1;
                    s1_fold_hits(code, idxp, cs, f.ncol, T, nqc, k, h, per);
                }
                FLMR_FOR_CT(ct, T) {
                    const float v = fmaxf(per[ct], __shfl_xor(per[ct], 32, 64));
                    if (h == 0) tr[(j0 + u) * ncolp + ct * 32 + k] = v;
                }
            }
        }
        emit_group(tr, ncolp, nqc, ndoc, lane, my_pid, keys_b + g0);
    }
}

int flmr_launch_filter_stage1(const flmr_filter_args& f, const uint32_t* idx_bits, int32_t idx_words,
                              const int32_t* cand, int64_t cand_stride, const int32_t* cand_count, uint64_t* keys,
                              hipStream_t st) {
    const int T = (f.nq_cand + 31) >> 5;
    const size_t tr_bytes = (size_t)S1_WAVES * S1_GROUP * (T * 32 + 1) * sizeof(float);
    const size_t idx_bytes = (size_t)idx_words * 4;
    const bool lds_idx = idx_bytes <= 40 * 1024;  // K <= 327680; larger K probes the mask through L1/L2
    // enough blocks per query to fill the chip even for one query; waves stride over the candidates
    int G = (int)flmr_ceil_div(256 * 8, f.nqueries);
    if (G < 1) G = 1;
    if (G > 256) G = 256;
    dim3 grid(f.nqueries, G), block(256);
    if (lds_idx)
        hipLaunchKernelGGL(filter_stage1_kernel<true>, grid, block, tr_bytes + idx_bytes, st, f, idx_bits, idx_words,
                           cand, cand_stride, cand_count, keys);
    else
        hipLaunchKernelGGL(filter_stage1_kernel<false>, grid, block, tr_bytes, st, f, idx_bits, idx_words, cand,
                           cand_stride, cand_count, keys);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Stage 2: every centroid counts (idx == all ones).  One wave per surviving doc; each half-wave gathers
// one 128-byte score row per step, 4 steps in flight.  grid = (nqueries, ceil(max_count / 4)), block 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_stage2_kernel(flmr_filter_args f, const int32_t* pids, int64_t pid_stride,
                                                            const int32_t* counts, uint64_t* keys,
                                                            int64_t key_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = counts[b];
    const int d = blockIdx.y * S1_WAVES + wave;
    const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
    const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;
    const int T = (f.nq_cand + 31) >> 5;
    const int ncolp = T * 32 + 1;
    float* tr = reinterpret_cast<float*>(smem) + (size_t)wave * ncolp;
    if (d >= n) return;  // no block-level barrier below
    const float* cs = f.cs + (size_t)b * f.cs_query_stride;
    const int pid = pids[(size_t)b * pid_stride + d];
    const int64_t off = f.offsets[pid];
    const int len = (int)doc_len_of(f.doclens, f.offsets, pid);
    const int k = lane & 31, h = lane >> 5;
    float per[4] = {-9999.0f, -9999.0f, -9999.0f, -9999.0f};
    for (int t0 = 0; t0 < len; t0 += 64) {
        const int t = t0 + lane;
        const int code = (t < len) ? f.codes[off + t] : -1;
        const int nt = (len - t0) < 64 ? (len - t0) : 64;
        for (int j = 0; j < nt; j += 8) {  // 4 rows per half-wave in flight
            int c[4];
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int src = j + 2 * u + h;
                c[u] = __shfl(code, src < nt ? src : j, 64);  // clamp: re-reads a valid row (max is idempotent)
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                FLMR_FOR_CT(ct, T) v[u][ct] = (ct * 32 + k < nqc) ? cs[(size_t)c[u] * f.ncol + ct * 32 + k] : -9999.0f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                FLMR_FOR_CT(ct, T) per[ct] = fmaxf(per[ct], v[u][ct]);
        }
    }
    FLMR_FOR_CT(ct, T) {
        const float v = fmaxf(per[ct], __shfl_xor(per[ct], 32, 64));
        if (h == 0) tr[ct * 32 + k] = v;
    }
    // same-wave LDS write -> read: wave-synchronous, but keep the compiler honest
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const float s = flmr_seq_sum(tr, nqc);
        keys[(size_t)b * key_stride + d] = flmr_make_key(s, pid);
    }
}

int flmr_launch_filter_stage2(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride,
                              const int32_t* counts, int32_t max_count, uint64_t* keys, int64_t key_stride,
                              hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    const size_t lds = (size_t)S1_WAVES * (((f.nq_cand + 31) >> 5) * 32 + 1) * sizeof(float);
    dim3 grid(f.nqueries, (unsigned)flmr_ceil_div(max_count, S1_WAVES)), block(256);
    hipLaunchKernelGGL(filter_stage2_kernel, grid, block, lds, st, f, pids, pid_stride, counts, keys, key_stride);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Radix select: the n largest of count[q] 64-bit keys, unordered.  grid = nqueries, block = 1024.
// 8 passes of 8 bits from the top; the histogram lives in LDS; the leader digit of each wave is
// aggregated with a ballot so the heavily repeated high bytes do not serialise on one LDS address.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_topn_kernel(const uint64_t* keys, int64_t key_stride,
                                                           const int32_t* counts, int32_t n, int32_t* out_pids,
                                                           int64_t out_stride, int32_t* n_out) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_remaining;
    __shared__ int s_out;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int P = counts[b];
    const uint64_t* kb = keys + (size_t)b * key_stride;
    int32_t* ob = out_pids + (size_t)b * out_stride;
    if (P <= n) {
        for (int i = tid; i < P; i += blockDim.x) ob[i] = flmr_key_pid(kb[i]);
        if (tid == 0) n_out[b] = P;
        return;
    }
    if (tid == 0) { s_prefix = 0ull; s_remaining = n; s_out = 0; }
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i0 = 0; i0 < P; i0 += blockDim.x) {
            const int i = i0 + tid;
            bool act = false;
            unsigned int dg = 0;
            if (i < P) {
                const uint64_t key = kb[i];
                act = (pass == 0) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
                dg = (unsigned int)(key >> shift) & 255u;
            }
            unsigned long long am = __ballot(act);
            if (am) {
                const int leader = __builtin_ctzll(am);
                const unsigned int ld = (unsigned int)__shfl((int)dg, leader, 64);
                const unsigned long long same = __ballot(act && dg == ld);
                if (lane == leader) atomicAdd(&hist[ld], (unsigned int)__popcll(same));
                if (act && dg != ld) atomicAdd(&hist[dg], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int rem = s_remaining;
            int dsel = 0;
            for (int dgt = 255; dgt >= 0; --dgt) {
                const int c = (int)hist[dgt];
                if (c >= rem) { dsel = dgt; break; }
                rem -= c;
            }
            s_remaining = rem;  // how many keys with the selected digit (and prefix) are still needed
            s_prefix = prefix | ((unsigned long long)dsel << shift);
        }
        __syncthreads();
    }
    // s_prefix is now the n-th largest key: keep everything >= it (keys are unique per pid)
    const unsigned long long thr = s_prefix;
    for (int i0 = 0; i0 < P; i0 += blockDim.x) {
        const int i = i0 + tid;
        if (i < P) {
            const uint64_t key = kb[i];
            if (key >= thr) {
                const int pos = atomicAdd(&s_out, 1);
                if (pos < n) ob[pos] = flmr_key_pid(key);
            }
        }
    }
    __syncthreads();
    if (tid == 0) n_out[b] = s_out < n ? s_out : n;
}

int flmr_launch_select_topn(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t nqueries,
                            int32_t n, int32_t* out_pids, int64_t out_stride, int32_t* n_out, hipStream_t st) {
    hipLaunchKernelGGL(select_topn_kernel, dim3(nqueries), dim3(1024), 0, st, keys, key_stride, counts, n, out_pids,
                       out_stride, n_out);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Sorted top-n of <= FLMR_MAX_NDOCS keys: in-LDS bitonic sort, descending.  grid = nqueries, block = 1024.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sort_topn_kernel(const uint64_t* keys, int64_t key_stride,
                                                         const int32_t* counts, int32_t npow2, int32_t n,
                                                         int32_t* out_pids, float* out_scores, int64_t out_stride,
                                                         int32_t* n_out, int64_t pid_base, int fill) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cnt = counts[b];
    for (int i = tid; i < npow2; i += blockDim.x) s[i] = (i < cnt) ? keys[(size_t)b * key_stride + i] : 0ull;
    __syncthreads();
    flmr_bitonic_sort_desc<unsigned long long>(s, npow2);
    const int m = cnt < n ? cnt : n;
    for (int i = tid; i < n; i += blockDim.x) {
        if (i < m) {
            const uint64_t key = s[i];
            out_pids[(size_t)b * out_stride + i] = (int32_t)(flmr_key_pid(key) + pid_base);
            if (out_scores) out_scores[(size_t)b * out_stride + i] = flmr_key_score(key);
        } else if (fill) {
            out_pids[(size_t)b * out_stride + i] = -1;
            if (out_scores) out_scores[(size_t)b * out_stride + i] = 0.0f;
        }
    }
    if (tid == 0 && n_out) n_out[b] = m;
}

int flmr_launch_sort_topn(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t max_count,
                          int32_t nqueries, int32_t n, int32_t* out_pids, float* out_scores, int64_t out_stride,
                          int32_t* n_out, int64_t pid_base, int fill, hipStream_t st) {
    if (max_count > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "sort_topn: %d keys > %d", max_count, FLMR_MAX_NDOCS);
    int npow2 = 2;
    while (npow2 < max_count) npow2 <<= 1;
    hipLaunchKernelGGL(sort_topn_kernel, dim3(nqueries), dim3(1024), (size_t)npow2 * 8, st, keys, key_stride, counts,
                       npow2, n, out_pids, out_scores, out_stride, n_out, pid_base, fill);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
