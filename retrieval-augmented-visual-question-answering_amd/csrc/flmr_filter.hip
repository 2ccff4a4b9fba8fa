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

#define S1_WAVES 8  // stage-1 block = 512 threads
#define S2_WAVES 4  // stage-2 block = 256 threads
#define S1_GROUP 32  // docs transposed per wave before the sequential sums

__device__ __forceinline__ int64_t doc_len_of(const int64_t* doclens, const int64_t* offsets, int pid) {
    return doclens ? doclens[pid] : (offsets[pid + 1] - offsets[pid]);
}

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
    const int lo = __shfl((int)(uint32_t)v, src, 64);
    const int hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src, 64);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// lanes < nslots emit the (score,pid) key of their doc: scanned docs sum their `nqc` column maxima k-ascending,
// docs that provably contain no surviving centroid take the precomputed all-miss score (-9999 summed nqc times)
__device__ __forceinline__ void emit_group(const float* tr /* [S1_GROUP][ncolp] */, int ncolp, int nqc, int nslots,
                                           int lane, int my_pid, bool scanned, float miss_score,
                                           uint64_t* keys_out /* &keys[group base] */, int f16 = 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < nslots) {
        const float s = scanned ? flmr_seq_sum(tr + lane * ncolp, nqc, f16) : miss_score;
        keys_out[lane] = flmr_make_key(s, my_pid);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Per-block view of one query's surviving-centroid set: the K-bit mask and its per-word exclusive popcount (the rank of a
// surviving centroid = its row in the compact score table), in LDS when they fit, else read through L1 / L2.
struct s1_idx_view {
    const uint32_t* bits;      // LDS (or global when the mask does not fit)
    const uint16_t* prefix16;  // LDS copy of the prefix (set when the mask is LDS-resident)
    const uint32_t* prefix32;  // global prefix (qualifying_kernel), used when the mask is not LDS-resident
    int compact;               // 1: score rows are addressed by rank, 0: by centroid id (full table)
    int row_cap;               // rows of the compact table (ranks are clamped to it: an over-capacity query is flagged, never read out of bounds)
};

#define S1_HITS_MAX 128   // tokens per document chunk = the most hits a half-wave can list
#define S1_PAIRS_IN_FLIGHT 4  // 8 documents = 16 code loads per wave in flight

// The surviving-centroid hits of one 128-token chunk of TWO documents (one per half-wave) are LISTED: lane i of a half holds
// tokens 4i .. 4i+3 of its document's chunk (cd[0..3], -1 = no token); every hit appends the row id of its centroid -- the
// centroid id (full table) or its rank among the query's surviving centroids (compact rows) -- to the half's list in LDS, at a
// position from ONE exclusive scan of the lanes' hit counts over the half (the order of a list does not matter: a maximum).
// Returns this half's number of hits (uniform over the half).  ~35 instructions per lane and chunk: the scan of a candidate's
// codes is instruction-bound long before it is bandwidth-bound.
__device__ __forceinline__ int s1_list_hits(const int* cd, const s1_idx_view& iv, int lane, uint16_t* list16, int* list32) {
    const int k = lane & 31, h = lane >> 5;
    uint32_t wd[4];
    uint32_t hm = 0u;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        wd[e] = cd[e] >= 0 ? iv.bits[cd[e] >> 5] : 0u;
        hm |= ((cd[e] >= 0) ? ((wd[e] >> (cd[e] & 31)) & 1u) : 0u) << e;
    }
    const int cnt = __popc(hm);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const int y = __shfl_up(incl, d, 32);
        if (k >= d) incl += y;
    }
    const int nh = __shfl(incl, 31, 32);
    if (hm) {
        int at = h * S1_HITS_MAX + incl - cnt;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (!((hm >> e) & 1u)) continue;
            const int c = cd[e];
            if (iv.compact) {
                const int rid = (int)(iv.prefix16 ? (uint32_t)iv.prefix16[c >> 5] : iv.prefix32[c >> 5]) + __popc(wd[e] & ((1u << (c & 31)) - 1u));
                list16[at] = (uint16_t)(rid < iv.row_cap ? rid : iv.row_cap - 1);
            } else {
                list32[at] = c;
            }
            at++;
        }
    }
    return nh;
}

// ------------------------------------------------------------------------------------------------
// Stage 1 by code scan: what runs for a query with more surviving centroids than the scatter form takes (> 1024, or lists
// longer than 8 x the probed cells') -- the usual case on a real corpus, the exception on the synthetic one.
// Persistent workgroups (8 waves), XCD-aware: the work items are (query, part g of G) for the queries that need the scan
// (every query when `skip` is NULL); item t belongs to XCD t % 8 and, inside it, to query (t / 8) / G -- workgroup L only takes
// items congruent to L modulo the grid, and workgroup L runs on XCD L % 8 (probed at index open), so the workgroups resident on
// one XCD work on a handful of queries at a time and their score rows (128 bytes per surviving centroid: 1 MB per query at
// 9 k survivors) stay in that XCD's L2.
// A wave takes 32 consecutive candidates at a time: lanes fetch the 32 (pid, offset, length) triples in parallel (the NEXT
// group's pids are prefetched while the current group is processed); documents are processed two at a time, one per half-wave,
// FOUR pairs per round: 16 code loads in flight, then the hits of all four pairs are listed, then their score rows are
// gathered -- BATCH rows per pair and half in flight, one wait per batch (the first form retired one hit per iteration with a
// DEPENDENT row load each: one L2 round trip per hit, 70 ms per 1024 queries at centroid_score_threshold = 0.25).
// TM = column tiles the kernel is compiled for: 1 (nq_cand <= 32, eight rows per pair in flight) or 4 (two).
// ------------------------------------------------------------------------------------------------
#define S1_ITEMS_G 32   // parts per query

template <bool USE_LDS_IDX, int TM>
__global__ __launch_bounds__(512, 4) void filter_stage1_kernel(flmr_filter_args f, const uint32_t* idx_bits,
                                                            int32_t idx_words, const int32_t* cand,
                                                            int64_t cand_stride, const int32_t* cand_count,
                                                            uint64_t* keys, const uint32_t* hit_bits,
                                                            int64_t hit_words, const int32_t* hit_valid,
                                                            const uint8_t* hit_flags, const int32_t* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int scan_lds[17];
    __shared__ int s_nscan;
    constexpr int BATCH = TM == 1 ? 8 : 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = (f.nq_cand + 31) >> 5;  // column tiles of 32 (<= TM); f.ncol is the row stride of the score table
    const int ncolp = T * 32 + 1;
    // LDS carve: [tr: S1_WAVES*S1_GROUP*ncolp f32][hit lists: S1_WAVES * PAIRS * 2 * S1_HITS_MAX u16 (compact) or i32][scan list:
    // nqueries i32][bits][prefix u16]
    float* tr = reinterpret_cast<float*>(smem) + (size_t)wave * S1_GROUP * ncolp;
    char* after_tr = reinterpret_cast<char*>(reinterpret_cast<float*>(smem) + (size_t)S1_WAVES * S1_GROUP * ncolp);
    const size_t list_elem = f.cs_compact ? sizeof(uint16_t) : sizeof(int);
    char* my_lists = after_tr + (size_t)wave * S1_PAIRS_IN_FLIGHT * 2 * S1_HITS_MAX * list_elem;
    int* scan_list = reinterpret_cast<int*>(after_tr + (size_t)S1_WAVES * S1_PAIRS_IN_FLIGHT * 2 * S1_HITS_MAX * list_elem);
    uint32_t* lidx = reinterpret_cast<uint32_t*>(scan_list + f.nqueries);
    uint16_t* lpre = reinterpret_cast<uint16_t*>(lidx + idx_words);

    // ---- which queries are scanned at all (usually none: every workgroup leaves here) ----
    if (threadIdx.x == 0) s_nscan = 0;
    __syncthreads();
    {
        int base = 0;
        for (int q0 = 0; q0 < f.nqueries; q0 += blockDim.x) {
            const int q = q0 + threadIdx.x;
            const int need = (q < f.nqueries && !(skip && skip[q])) ? 1 : 0;
            int total;
            const int pos = base + flmr_block_exclusive_scan(need, scan_lds, &total);
            if (need) scan_list[pos] = q;
            base += total;
            __syncthreads();
        }
        if (threadIdx.x == 0) s_nscan = base;
    }
    __syncthreads();
    const int nscan = s_nscan;
    if (nscan == 0) return;
    const int nscan8 = (nscan + 7) & ~7;
    const int k = lane & 31, h = lane >> 5;
    const int nitems = nscan8 * S1_ITEMS_G;

    for (int t = blockIdx.x; t < nitems; t += gridDim.x) {
        const int sidx = ((t >> 3) / S1_ITEMS_G) * 8 + (t & 7), part = (t >> 3) % S1_ITEMS_G;
        if (sidx >= nscan) continue;   // (block-uniform)
        const int b = scan_list[sidx];
        const int P = cand_count[b];
        const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
        const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;
        const uint32_t* gidx = idx_bits + (size_t)b * idx_words;
        const float* cs = f.cs + (size_t)b * f.cs_query_stride;
        s1_idx_view iv;
        iv.bits = gidx; iv.prefix16 = nullptr; iv.compact = f.cs_compact; iv.row_cap = f.row_cap;
        iv.prefix32 = f.idx_prefix ? f.idx_prefix + (size_t)b * idx_words : nullptr;
        __syncthreads();   // (the previous item's readers of the mask are done)
        if (USE_LDS_IDX) {
            // mask + exclusive popcount prefix per word (saturating: the compact table holds fewer than 65535 rows)
            int base = 0;
            for (int w0 = 0; w0 < idx_words; w0 += blockDim.x) {
                const int w = w0 + threadIdx.x;
                const uint32_t bits = (w < idx_words) ? gidx[w] : 0u;
                int total;
                const int pos = base + flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
                if (w < idx_words) {
                    lidx[w] = bits;
                    lpre[w] = (uint16_t)(pos < 65535 ? pos : 65535);
                }
                base += total;
                __syncthreads();
            }
            iv.bits = lidx; iv.prefix16 = lpre;
        }
        const int32_t* cand_b = cand + (size_t)b * cand_stride;
        uint64_t* keys_b = keys + (size_t)b * cand_stride;
        const int wid = part * S1_WAVES + wave;
        const int gstride = S1_ITEMS_G * S1_WAVES * S1_GROUP;

        // hit_bits (optional): passage bitmap = union of the IVF lists of the surviving centroids.  A candidate outside
        // it contains no surviving centroid, so its stage-1 score is the all-miss value and its codes are never read.
        const bool hits_on = hit_valid && hit_valid[b];
        const uint32_t* hb = (hit_bits && hits_on && !hit_flags) ? hit_bits + (size_t)b * hit_words : nullptr;
        const uint8_t* hf = (hit_flags && hits_on) ? hit_flags + (size_t)b * cand_stride : nullptr;  // aligned with cand
        const float miss_score = flmr_miss_score(nqc, f.f16_round);

        int my_pid = 0;
        bool my_scan = false;
        int g0 = wid * S1_GROUP;
        if (g0 + lane < P && lane < S1_GROUP) {
            my_pid = cand_b[g0 + lane];
            my_scan = hf ? (hf[g0 + lane] != 0) : hb ? ((hb[my_pid >> 5] >> (my_pid & 31)) & 1u) : true;
        }
        for (; g0 < P; g0 += gstride) {
            const int ndoc = (P - g0) < S1_GROUP ? (P - g0) : S1_GROUP;
            // prefetch the next group's pids + hit bits
            int nx_pid = 0;
            bool nx_scan = false;
            if (g0 + gstride + lane < P && lane < S1_GROUP) {
                nx_pid = cand_b[g0 + gstride + lane];
                nx_scan = hf ? (hf[g0 + gstride + lane] != 0) : hb ? ((hb[nx_pid >> 5] >> (nx_pid & 31)) & 1u) : true;
            }
            int my_len = 0;
            int64_t my_off = 0;
            if (my_scan) {
                my_off = f.offsets[my_pid];
                my_len = (int)doc_len_of(f.doclens, f.offsets, my_pid);
            }
            unsigned long long todo = __ballot(my_scan);
            // Rounds of S1_PAIRS_IN_FLIGHT pairs of documents, one document per half-wave, software-pipelined: the codes of round
            // r + 1 are requested as soon as round r's hits are listed (its code registers are dead then), so they travel while
            // round r's score rows are gathered.  Codes are read once: non-temporal, they must not evict the score rows from L2.
            int jh[S1_PAIRS_IN_FLIGHT], len[S1_PAIRS_IN_FLIGHT];
            int64_t off[S1_PAIRS_IN_FLIGHT];
            int cd[S1_PAIRS_IN_FLIGHT][4];
            // lane k of a half takes tokens t0 + 4k .. + 3 of its document: one 16-byte load (the codes of a passage are
            // contiguous, 4-byte aligned); the lane that straddles the passage's end reads its tokens one by one (nothing is
            // read past the end of the code array)
            auto load_codes = [&](int (&c_)[4], int64_t o, int n, int t0) {
                const int first = t0 + 4 * k;
                c_[0] = c_[1] = c_[2] = c_[3] = -1;
                if (first + 4 <= n) {
                    typedef int s1_int4u __attribute__((ext_vector_type(4), aligned(4)));
                    const s1_int4u v = __builtin_nontemporal_load(reinterpret_cast<const s1_int4u*>(f.codes + o + first));
                    c_[0] = v.x; c_[1] = v.y; c_[2] = v.z; c_[3] = v.w;
                } else if (first < n) {
#pragma unroll
                    for (int e = 0; e < 3; e++)
                        if (first + e < n) c_[e] = __builtin_nontemporal_load(f.codes + o + first + e);
                }
            };
            auto prep = [&](int (&jh_)[S1_PAIRS_IN_FLIGHT], int (&len_)[S1_PAIRS_IN_FLIGHT], int64_t (&off_)[S1_PAIRS_IN_FLIGHT]) {
#pragma unroll
                for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++) {
                    const int ja = todo ? __builtin_ctzll(todo) : -1;
                    todo &= todo - 1;
                    const int jb = todo ? __builtin_ctzll(todo) : -1;
                    todo &= todo - 1;
                    jh_[u] = h ? jb : ja;  // this half's document slot (-1: none)
                    const int j = jh_[u] < 0 ? 0 : jh_[u];
                    off_[u] = shfl_i64(my_off, j);
                    len_[u] = jh_[u] < 0 ? 0 : __shfl(my_len, j, 64);
                    load_codes(cd[u], off_[u], len_[u], 0);
                }
            };
            bool more = todo != 0ull;
            if (more) prep(jh, len, off);
            while (more) {  // wave-uniform
                float per[S1_PAIRS_IN_FLIGHT][TM];
#pragma unroll
                for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++)
#pragma unroll
                    for (int ct = 0; ct < TM; ct++) per[u][ct] = -9999.0f;
                int nh[S1_PAIRS_IN_FLIGHT];
                auto list_round = [&](const int (&c_)[S1_PAIRS_IN_FLIGHT][4]) {
                    int nmax = 0;
#pragma unroll
                    for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++) {
                        nh[u] = s1_list_hits(c_[u], iv, lane, reinterpret_cast<uint16_t*>(my_lists) + u * 2 * S1_HITS_MAX,
                                             reinterpret_cast<int*>(my_lists) + u * 2 * S1_HITS_MAX);
                        const int n0 = __builtin_amdgcn_readlane(nh[u], 0), n1 = __builtin_amdgcn_readlane(nh[u], 32);
                        nmax = nmax > n0 ? nmax : n0;
                        nmax = nmax > n1 ? nmax : n1;
                    }
                    return nmax;
                };
                auto gather_round = [&](int nmax) {
                    if (nmax == 0) return;   // wave-uniform
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int j0 = 0; j0 < nmax; j0 += BATCH) {   // wave-uniform trip count
                        float v[S1_PAIRS_IN_FLIGHT][BATCH][TM];
#pragma unroll
                        for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++)
#pragma unroll
                            for (int w = 0; w < BATCH; w++) {
                                const bool ok = j0 + w < nh[u];
                                const int at = (u * 2 + h) * S1_HITS_MAX + (ok ? j0 + w : 0);
                                const int rid = f.cs_compact ? (int)reinterpret_cast<const uint16_t*>(my_lists)[at]
                                                             : reinterpret_cast<const int*>(my_lists)[at];
#pragma unroll
                                for (int ct = 0; ct < TM; ct++)
                                    v[u][w][ct] = (ok && ct < T && ct * 32 + k < nqc) ? cs[(size_t)rid * f.ncol + ct * 32 + k] : -9999.0f;
                            }
#pragma unroll
                        for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++)
#pragma unroll
                            for (int w = 0; w < BATCH; w++)
#pragma unroll
                                for (int ct = 0; ct < TM; ct++) per[u][ct] = fmaxf(per[u][ct], v[u][w][ct]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // (the lists are rewritten by the next chunk / round)
                };
                const int nmax0 = list_round(cd);
                // this round's bookkeeping, then the next round's codes on their way
                int cjh[S1_PAIRS_IN_FLIGHT], clen[S1_PAIRS_IN_FLIGHT];
                int64_t coff[S1_PAIRS_IN_FLIGHT];
#pragma unroll
                for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++) { cjh[u] = jh[u]; clen[u] = len[u]; coff[u] = off[u]; }
                more = todo != 0ull;
                if (more) prep(jh, len, off);
                gather_round(nmax0);
                // documents longer than 128 tokens: further 128-token chunks of this round (rare)
                for (int t0 = 128; __ballot(t0 < clen[0] || t0 < clen[1] || t0 < clen[2] || t0 < clen[3]) != 0ull; t0 += 128) {
                    int cx[S1_PAIRS_IN_FLIGHT][4];
#pragma unroll
                    for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++) load_codes(cx[u], coff[u], clen[u], t0);
                    gather_round(list_round(cx));
                }
#pragma unroll
                for (int u = 0; u < S1_PAIRS_IN_FLIGHT; u++)
                    if (cjh[u] >= 0) {
#pragma unroll
                        for (int ct = 0; ct < TM; ct++)
                            if (ct < T) tr[cjh[u] * ncolp + ct * 32 + k] = per[u][ct];
                    }
            }
            emit_group(tr, ncolp, nqc, ndoc, lane, my_pid, my_scan, miss_score, keys_b + g0, f.f16_round);
            my_pid = nx_pid; my_scan = nx_scan;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Hit bitmap: union of the IVF lists of the centroids that survive the threshold.  Built only when that union is
// cheaper than scanning (sum of list lengths <= 2 x #candidates); otherwise hit_valid[q] = 0 and stage 1 scans all.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_hit_budget_kernel(const uint32_t* idx_bits, int32_t idx_words,
                                                            const int64_t* ivf_offsets, const int32_t* cand_count,
                                                            int32_t* hit_valid) {
    __shared__ unsigned long long total;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) total = 0ull;
    __syncthreads();
    unsigned long long mine = 0;
    for (int w = threadIdx.x; w < idx_words; w += blockDim.x) {
        uint32_t bits = idx_bits[(size_t)b * idx_words + w];
        while (bits) {
            const int c = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1;
            mine += (unsigned long long)(ivf_offsets[c + 1] - ivf_offsets[c]);
        }
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) hit_valid[b] = total <= 2ull * (unsigned long long)cand_count[b];
}

__global__ __launch_bounds__(256) void s1_mark_hits_kernel(const uint32_t* idx_bits, int32_t idx_words,
                                                           const int32_t* ivf_pids, const int64_t* ivf_offsets,
                                                           const int32_t* hit_valid, uint32_t* hit_bits,
                                                           int64_t hit_words) {
    const int b = blockIdx.x;
    if (!hit_valid[b]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* hb = hit_bits + (size_t)b * hit_words;
    const int w0 = (blockIdx.y * 4 + wave) * 16;
    for (int w = w0; w < w0 + 16 && w < idx_words; w++) {
        uint32_t bits = idx_bits[(size_t)b * idx_words + w];  // wave-uniform
        while (bits) {
            const int c = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1;
            const int64_t beg = ivf_offsets[c], end = ivf_offsets[c + 1];
            for (int64_t e = beg + lane; e < end; e += 64) {
                const int pid = ivf_pids[e];
                atomicOr(&hb[pid >> 5], 1u << (pid & 31));
            }
        }
    }
}

int flmr_launch_hit_bitmap(const uint32_t* idx_bits, int32_t idx_words, int32_t nqueries, const int32_t* ivf_pids,
                           const int64_t* ivf_offsets, const int32_t* cand_count, uint32_t* hit_bits, int64_t hit_words,
                           int32_t* hit_valid, hipStream_t st) {
    FLMR_HIP(hipMemsetAsync(hit_bits, 0, (size_t)nqueries * hit_words * sizeof(uint32_t), st));
    hipLaunchKernelGGL(s1_hit_budget_kernel, dim3(nqueries), dim3(256), 0, st, idx_bits, idx_words, ivf_offsets, cand_count,
                       hit_valid);
    hipLaunchKernelGGL(s1_mark_hits_kernel, dim3(nqueries, (unsigned)flmr_ceil_div(idx_words, 64)), dim3(256), 0, st,
                       idx_bits, idx_words, ivf_pids, ivf_offsets, hit_valid, hit_bits, hit_words);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_filter_stage1(const flmr_filter_args& f, const uint32_t* idx_bits, int32_t idx_words,
                              const int32_t* cand, int64_t cand_stride, const int32_t* cand_count, uint64_t* keys,
                              const uint32_t* hit_bits, int64_t hit_words, const int32_t* hit_valid,
                              const uint8_t* hit_flags, hipStream_t st, const int32_t* skip) {
    const int T = (f.nq_cand + 31) >> 5;
    if (T > 4) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "stage 1: nq_cand=%d > 128", f.nq_cand);
    const size_t tr_bytes = (size_t)S1_WAVES * S1_GROUP * (T * 32 + 1) * sizeof(float);
    const size_t list_bytes = (size_t)S1_WAVES * S1_PAIRS_IN_FLIGHT * 2 * S1_HITS_MAX * (f.cs_compact ? sizeof(uint16_t) : sizeof(int)) +
                              (size_t)f.nqueries * sizeof(int);   // the waves' hit lists + the list of scanned queries
    const size_t idx_bytes = (size_t)idx_words * 4 + (size_t)idx_words * 2 + 16;
    const bool lds_idx = tr_bytes + list_bytes + idx_bytes <= 100 * 1024;  // else probe the mask (and its prefix) through L1/L2
    if (f.cs_compact && !lds_idx && !f.idx_prefix) FLMR_FAIL(FLMR_ERR_INVALID, "compact score rows need the idx prefix");
    const size_t lds = tr_bytes + list_bytes + (lds_idx ? idx_bytes : 0);
    const void* fn = lds_idx ? (T == 1 ? reinterpret_cast<const void*>(filter_stage1_kernel<true, 1>) : reinterpret_cast<const void*>(filter_stage1_kernel<true, 4>))
                             : (T == 1 ? reinterpret_cast<const void*>(filter_stage1_kernel<false, 1>) : reinterpret_cast<const void*>(filter_stage1_kernel<false, 4>));
    if (lds > 48 * 1024) FLMR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: two rounds of the chip's resident workgroups (two per CU at this LDS size), a multiple of 8 so that
    // workgroup L and its items share L % 8; never more than there are items
    int64_t grid = 1024;
    const int64_t max_items = (int64_t)((f.nqueries + 7) & ~7) * S1_ITEMS_G;
    if (grid > max_items) grid = max_items;
    dim3 g((unsigned)grid), block(S1_WAVES * 64);
#define S1_LAUNCH(L, TMV) hipLaunchKernelGGL((filter_stage1_kernel<L, TMV>), g, block, lds, st, f, idx_bits, idx_words, cand, cand_stride, \
                                             cand_count, keys, hit_bits, hit_words, hit_valid, hit_flags, skip)
    if (lds_idx) { if (T == 1) S1_LAUNCH(true, 1); else S1_LAUNCH(true, 4); }
    else { if (T == 1) S1_LAUNCH(false, 1); else S1_LAUNCH(false, 4); }
#undef S1_LAUNCH
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
    const int d = blockIdx.y * S2_WAVES + wave;
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
        const float s = flmr_seq_sum(tr, nqc, f.f16_round);
        keys[(size_t)b * key_stride + d] = flmr_make_key(s, pid);
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 2, recompute variant (default on the fp16-centroid path): instead of gathering 128-byte rows of the per-query
// score table (which then has to be written in full by S0: 4*K*32 bytes per query, the largest HBM stream of the path),
// the centroid scores of a survivor's tokens are RECOMPUTED from the fp16 centroid rows with exactly the MFMA sequence
// S0 uses (2 x v_mfma_f32_32x32x16_f16 per 16 dims against q_hi / q_lo, then fma(lo, 2^-11, hi)): an output element
// depends only on its A row and B column, so the values are bitwise those S0 would have stored.  The 33 MB fp16
// centroid matrix is shared by every query and lives in the Infinity Cache; the per-query tables never exist.
// One wave per survivor; lanes fetch their documents' (pid, offset, length) up front, the next document's codes and the
// next token tile's centroid half-rows are prefetched (same pipeline as the MaxSim kernel).
// grid = (nqueries, G), block = 256, dynamic LDS = 4 * (ncol+1) floats.
// ------------------------------------------------------------------------------------------------
typedef _Float16 s2h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void filter_stage2_mfma_kernel(flmr_filter_args f, const int32_t* pids, int64_t pid_stride,
                                                                    const int32_t* counts, uint64_t* keys, int64_t key_stride,
                                                                    const _Float16* __restrict__ cen16,
                                                                    const _Float16* __restrict__ q_hi,
                                                                    const _Float16* __restrict__ q_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int cnt = counts[b];
    const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
    const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;  // <= 32 on this path
    float* tr = reinterpret_cast<float*>(smem) + (size_t)wave * 33;
    const int W = gridDim.y * 4, w = blockIdx.y * 4 + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // <= 64 documents per wave (launcher)
    if (ndw == 0) return;
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = pids[(size_t)b * pid_stride + w + lane * W];
        my_off = f.offsets[my_pid];
        my_len = (int)doc_len_of(f.doclens, f.offsets, my_pid);
    }
    s2h8 bh[8], bl[8];
    {
        const s2h8* ph = reinterpret_cast<const s2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
        const s2h8* pl = reinterpret_cast<const s2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    auto load_codes = [&](int64_t off, int len, int* cd) {
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = (lane + 64 * r < len) ? f.codes[off + lane + 64 * r] : 0;
    };
    auto issue_rows = [&](s2h8* a, const int* cd, int t, int64_t off, int len) {
        const int tok = t * 32 + i;
        int code = 0;
        if (t < 8) {
            const int sel = t >> 1;
            const int reg = sel == 0 ? cd[0] : sel == 1 ? cd[1] : sel == 2 ? cd[2] : cd[3];
            code = __shfl(reg, (t & 1) * 32 + i, 64);
        } else if (tok < len) {
            code = f.codes[off + tok];
        }
        const s2h8* pc = reinterpret_cast<const s2h8*>(cen16 + (size_t)code * FLMR_DIM + 64 * h);  // row 0 for padding tokens
#pragma unroll
        for (int s = 0; s < 8; s++) a[s] = pc[s];
    };
    int cd[4], ncd[4];
    load_codes(shfl_i64(my_off, 0), __shfl(my_len, 0, 64), cd);
    s2h8 araw[8];
    bool have_raw = false;
    for (int j = 0; j < ndw; j++) {
        const int pid = __shfl(my_pid, j, 64);
        const int len = __shfl(my_len, j, 64);
        const int64_t off = shfl_i64(my_off, j);
        int nlen = 0;
        int64_t noff = 0;
        if (j + 1 < ndw) {
            nlen = __shfl(my_len, j + 1, 64);
            noff = shfl_i64(my_off, j + 1);
            load_codes(noff, nlen, ncd);
        }
        const int ntiles = (len + 31) >> 5;
        if (ntiles > 0 && !have_raw) issue_rows(araw, cd, 0, off, len);
        have_raw = false;
        float cmax = -9999.0f;  // filter_pids.cpp:30-33: per-token maxima start at -9999
        for (int t = 0; t < ntiles; t++) {
            s2h8 av[8];
#pragma unroll
            for (int s = 0; s < 8; s++) av[s] = araw[s];
            if (t + 1 < ntiles) {
                issue_rows(araw, cd, t + 1, off, len);
            } else if (j + 1 < ndw && nlen > 0) {
                issue_rows(araw, ncd, 0, noff, nlen);
                have_raw = true;
            }
            f32x16 ah, al;
#pragma unroll
            for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[s], ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[s], al, 0, 0, 0);
            }
            const int nrow = len - t * 32;  // valid token rows in this tile
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                cmax = fmaxf(cmax, row < nrow ? v : -9999.0f);
            }
        }
        cmax = flmr_xhalf_max(cmax);
        if (h == 0) tr[i] = cmax;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const float sc = flmr_seq_sum(tr, nqc, f.f16_round);
            keys[(size_t)b * key_stride + w + j * W] = flmr_make_key(sc, pid);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = ncd[r];
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 1 WITHOUT score rows: the recovery form for a query with more surviving centroids than the searcher keeps compact rows
// for (`row_ovf[q]`, raised by qualifying_kernel: more than FLMR_ROW_CAP / 65535 centroids above the threshold -- thresholds near
// zero).  The reference has no such limit (index_storage.py:116), so neither may the library: instead of reporting the query, its
// stage 1 is RECOMPUTED here exactly as stage 2 recomputes its scores -- the candidates' token rows of the fp16 centroid table
// through the stage-0 MFMA sequence (bitwise the values stage 0 computes) -- with the idx mask applied per token
// (filter_pids.cpp:36-47: a code outside idx contributes nothing).  Slow (every token of every candidate costs an MFMA row) and
// rare; launched with every batch of the sparse path, its persistent workgroups leave at once when no query is flagged.
// item = (flagged query, block of 256 candidates: 64 per wave).  grid <= 1024, block = 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void filter_stage1_recompute_kernel(flmr_filter_args f, const uint32_t* __restrict__ idx_bits, int idx_words,
                                                                         const int32_t* __restrict__ cand, int64_t cand_stride,
                                                                         const int32_t* __restrict__ cand_count, const int32_t* __restrict__ row_ovf,
                                                                         uint64_t* keys, const _Float16* __restrict__ cen16,
                                                                         const _Float16* __restrict__ q_hi, const _Float16* __restrict__ q_lo) {
    __shared__ float tr_all[4 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    {
        int anyf = 0;
        for (int q = threadIdx.x; q < f.nqueries; q += 256) anyf |= row_ovf[q];
        if (!__syncthreads_or(anyf)) return;   // the usual case: no query of the batch is over the row capacity
    }
    float* tr = tr_all + wave * 33;
    for (int b = 0; b < f.nqueries; b++) {
        if (!row_ovf[b]) continue;   // (block-uniform)
        const int cnt = cand_count[b];
        const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
        const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;
        const uint32_t* gidx = idx_bits + (size_t)b * idx_words;
        s2h8 bh[8], bl[8];
        {
            const s2h8* ph = reinterpret_cast<const s2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
            const s2h8* pl = reinterpret_cast<const s2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
            for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
        }
        for (int blk = blockIdx.x; blk * 256 < cnt; blk += gridDim.x) {
            const int d0 = blk * 256 + wave * 64;
            const int ndw = cnt - d0 < 64 ? cnt - d0 : 64;
            if (ndw <= 0) continue;   // (wave-uniform; no block-wide barrier below)
            int my_pid = 0, my_len = 0;
            int64_t my_off = 0;
            if (lane < ndw) {
                my_pid = cand[(size_t)b * cand_stride + d0 + lane];
                my_off = f.offsets[my_pid];
                my_len = (int)doc_len_of(f.doclens, f.offsets, my_pid);
            }
            for (int j = 0; j < ndw; j++) {
                const int pid = __shfl(my_pid, j, 64);
                const int len = __shfl(my_len, j, 64);
                const int64_t off = shfl_i64(my_off, j);
                const int ntiles = (len + 31) >> 5;
                float cmax = -9999.0f;  // filter_pids.cpp:30-33
                bool any = false;
                for (int t = 0; t < ntiles; t++) {
                    const int tok = t * 32 + i;
                    const int code = tok < len ? f.codes[off + tok] : 0;
                    const bool ok = tok < len && ((gidx[code >> 5] >> (code & 31)) & 1u);
                    const uint32_t okmask = (uint32_t)__ballot(ok);   // bit r: token row r of the tile counts (lanes 0..31 = rows)
                    if (okmask == 0u) continue;   // (wave-uniform)
                    any = true;
                    const s2h8* pc = reinterpret_cast<const s2h8*>(cen16 + (size_t)code * FLMR_DIM + 64 * h);
                    s2h8 av[8];
#pragma unroll
                    for (int s = 0; s < 8; s++) av[s] = pc[s];
                    f32x16 ah, al;
#pragma unroll
                    for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[s], ah, 0, 0, 0);
                        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[s], al, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                        cmax = fmaxf(cmax, ((okmask >> row) & 1u) ? v : -9999.0f);
                    }
                }
                cmax = flmr_xhalf_max(cmax);
                if (h == 0) tr[i] = cmax;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    const float sc = any ? flmr_seq_sum(tr, nqc, f.f16_round) : flmr_miss_score(nqc, f.f16_round);
                    keys[(size_t)b * cand_stride + d0 + j] = flmr_make_key(sc, pid);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

int flmr_launch_filter_stage1_recompute(const flmr_filter_args& f, const uint32_t* idx_bits, int32_t idx_words, const int32_t* cand,
                                        int64_t cand_stride, const int32_t* cand_count, const int32_t* row_ovf, uint64_t* keys,
                                        const _Float16* cen16, const _Float16* q_hi, const _Float16* q_lo, hipStream_t st) {
    if (f.ncol != 32 || !cen16) FLMR_FAIL(FLMR_ERR_INVALID, "stage-1 recompute needs the sparse path (one column tile, fp16 centroids)");
    hipLaunchKernelGGL(filter_stage1_recompute_kernel, dim3(1024), dim3(256), 0, st, f, idx_bits, idx_words, cand, cand_stride, cand_count,
                       row_ovf, keys, cen16, q_hi, q_lo);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// WAVES = 4, B_LDS = false: the query's fp16 hi/lo operand (64 VGPRs) stays in registers -> 217 VGPRs, 2 waves per SIMD.
// WAVES = 16, B_LDS = true: one 1024-thread block per CU shares the operand through LDS (16 KB, stored in the lane order of the
// MFMA B operand: conflict-free ds_read_b128) -> <= 128 VGPRs, 4 waves per SIMD to hide the row-gather latency.
template <int WAVES, bool B_LDS>
__global__ __launch_bounds__(64 * WAVES, B_LDS ? 1 : 2) void filter_stage2_lds_kernel(flmr_filter_args f, const int32_t* pids, int64_t pid_stride,
                                                                    const int32_t* counts, uint64_t* keys, int64_t key_stride,
                                                                    const _Float16* __restrict__ cen16,
                                                                    const _Float16* __restrict__ q_hi,
                                                                    const _Float16* __restrict__ q_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int cnt = counts[b];
    const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
    const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;  // <= 32 on this path
    float* tr = reinterpret_cast<float*>(smem) + (size_t)wave * 33;
    // this wave's row buffer: 32 centroid rows x 256 B, filled by direct global->LDS loads (16-byte pieces XOR-swizzled by row)
    constexpr size_t TR_BYTES = (WAVES * 33 * sizeof(float) + 15) / 16 * 16;
    char* rowbuf = smem + TR_BYTES + (size_t)wave * (32 * 256);
    const s2h8* const Bh = reinterpret_cast<const s2h8*>(smem + TR_BYTES + (size_t)WAVES * (32 * 256));  // [8][64], then Bl [8][64]
    if constexpr (B_LDS) {
        s2h8* Bw = reinterpret_cast<s2h8*>(smem + TR_BYTES + (size_t)WAVES * (32 * 256));
        for (int t = threadIdx.x; t < 1024; t += 64 * WAVES) {
            const int lo = t >> 9, sidx = (t >> 6) & 7, ln = t & 63;
            const _Float16* src = (lo ? q_lo : q_hi) + ((size_t)b * f.ncol + (ln & 31)) * FLMR_DIM + 64 * (ln >> 5) + 8 * sidx;
            Bw[t] = *reinterpret_cast<const s2h8*>(src);
        }
        __syncthreads();
    }
    const int W = gridDim.y * WAVES, w = blockIdx.y * WAVES + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // <= 64 documents per wave (launcher)
    if (ndw == 0) return;
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = pids[(size_t)b * pid_stride + w + lane * W];
        my_off = f.offsets[my_pid];
        my_len = (int)doc_len_of(f.doclens, f.offsets, my_pid);
    }
    s2h8 bh[B_LDS ? 1 : 8], bl[B_LDS ? 1 : 8];
    if constexpr (!B_LDS) {
        const s2h8* ph = reinterpret_cast<const s2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
        const s2h8* pl = reinterpret_cast<const s2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    auto load_codes = [&](int64_t off, int len, int* cd) {
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = (lane + 64 * r < len) ? f.codes[off + lane + 64 * r] : 0;
    };
    // One gather instruction fetches FOUR WHOLE ROWS straight into LDS (global_load_lds_dwordx4; lane L: row 4g + L/16,
    // 16-byte piece L%16): 8 cache lines per instruction, each fully used, and no VGPRs held while the loads are in flight.
    // The register form (lane (i,h) reads its own half row 16 bytes at a time) touches 64 lines per instruction for the same
    // 1 KB.  The MFMA layout is read back from LDS; piece p of row r is kept at position p ^ (r & 15) so that both the
    // contiguous DMA writes and the ds_read_b128 are conflict-free.  Measured: 1.04 vs 1.08 ms -- the kernel is bound by the
    // number of line fills a CU keeps in flight (x latency), see DESIGN.md, so the form of the gather matters little.
    auto issue_rows = [&](const int* cd, int t, int64_t off, int len) {
        const int tok = t * 32 + i;
        int code = 0;
        if (t < 8) {
            const int sel = t >> 1;
            const int reg = sel == 0 ? cd[0] : sel == 1 ? cd[1] : sel == 2 ? cd[2] : cd[3];
            code = __shfl(reg, (t & 1) * 32 + i, 64);
        } else if (tok < len) {
            code = f.codes[off + tok];
        }
        const int rl = lane >> 4, pp = lane & 15;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int row = 4 * g + rl;
            const int c = __shfl(code, row, 64);  // lane `row` (< 32) holds token `row`'s code (row 0 for padding tokens)
            const _Float16* src = cen16 + (size_t)c * FLMR_DIM + ((pp ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(rowbuf + g * 1024), 16, 0, 0);
        }
    };
    auto take_rows = [&](s2h8* a) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tile's DMA loads have landed
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 8; s++)
            a[s] = *reinterpret_cast<const s2h8*>(rowbuf + i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the buffer may be refilled
        asm volatile("" ::: "memory");
    };
    int cd[4], ncd[4];
    load_codes(shfl_i64(my_off, 0), __shfl(my_len, 0, 64), cd);
    bool have_raw = false;
    for (int j = 0; j < ndw; j++) {
        const int pid = __shfl(my_pid, j, 64);
        const int len = __shfl(my_len, j, 64);
        const int64_t off = shfl_i64(my_off, j);
        int nlen = 0;
        int64_t noff = 0;
        if (j + 1 < ndw) {
            nlen = __shfl(my_len, j + 1, 64);
            noff = shfl_i64(my_off, j + 1);
            load_codes(noff, nlen, ncd);
        }
        const int ntiles = (len + 31) >> 5;
        if (ntiles > 0 && !have_raw) issue_rows(cd, 0, off, len);
        have_raw = false;
        float cmax = -9999.0f;  // filter_pids.cpp:30-33: per-token maxima start at -9999
        for (int t = 0; t < ntiles; t++) {
            s2h8 av[8];
            take_rows(av);
            if (t + 1 < ntiles) {
                issue_rows(cd, t + 1, off, len);
            } else if (j + 1 < ndw && nlen > 0) {
                issue_rows(ncd, 0, noff, nlen);
                have_raw = true;
            }
            f32x16 ah, al;
#pragma unroll
            for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                if constexpr (B_LDS) {
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], Bh[s * 64 + lane], ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], Bh[512 + s * 64 + lane], al, 0, 0, 0);
                } else {
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[s], ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[s], al, 0, 0, 0);
                }
            }
            const int nrow = len - t * 32;  // valid token rows in this tile
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                cmax = fmaxf(cmax, row < nrow ? v : -9999.0f);
            }
        }
        cmax = flmr_xhalf_max(cmax);
        if (h == 0) tr[i] = cmax;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const float sc = flmr_seq_sum(tr, nqc, f.f16_round);
            keys[(size_t)b * key_stride + w + j * W] = flmr_make_key(sc, pid);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = ncd[r];
    }
}

int flmr_launch_filter_stage2_mfma(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                   int32_t max_count, uint64_t* keys, int64_t key_stride, const _Float16* cen16,
                                   const _Float16* q_hi, const _Float16* q_lo, hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    if (f.ncol != 32) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 recompute needs one column tile");
    int G = (int)flmr_ceil_div(8192, 4 * (int64_t)f.nqueries);
    const int gmin = (int)flmr_ceil_div(max_count, 4 * 64);
    if (G < gmin) G = gmin;
    if (G > (int)flmr_ceil_div(max_count, 4)) G = (int)flmr_ceil_div(max_count, 4);
    if (G < 1) G = 1;
#ifdef FLMR_EXPERIMENTAL_VARIANTS
    // Measured losers, compiled only into a -DFLMR_EXPERIMENTAL_VARIANTS library (profiles/build_variant.py):
    // FLMR_S2_IMPL=regs: the first form of the kernel (per-lane half-row gathers into registers);
    // FLMR_S2_IMPL=ldsb: 16-wave blocks with the query operand in LDS (5.6 ms against 4.1 ms)
    if (flmr_opts().is(FLMR_OPT_S2_IMPL, "regs"))
        hipLaunchKernelGGL(filter_stage2_mfma_kernel, dim3(f.nqueries, G), dim3(256), 4 * 33 * sizeof(float), st, f, pids, pid_stride,
                           counts, keys, key_stride, cen16, q_hi, q_lo);
    else if (flmr_opts().is(FLMR_OPT_S2_IMPL, "ldsb")) {
        // 16-wave blocks sharing the query operand through LDS
        int G16 = (int)flmr_ceil_div(8192, 16 * (int64_t)f.nqueries);
        const int g16min = (int)flmr_ceil_div(max_count, 16 * 64);
        if (G16 < g16min) G16 = g16min;
        if (G16 > (int)flmr_ceil_div(max_count, 16)) G16 = (int)flmr_ceil_div(max_count, 16);
        if (G16 < 1) G16 = 1;
        const size_t lds16 = (16 * 33 * sizeof(float) + 15) / 16 * 16 + (size_t)16 * 32 * 256 + 16384;
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_stage2_lds_kernel<16, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
        hipLaunchKernelGGL((filter_stage2_lds_kernel<16, true>), dim3(f.nqueries, G16), dim3(1024), lds16, st, f, pids, pid_stride,
                           counts, keys, key_stride, cen16, q_hi, q_lo);
    } else
#endif
        hipLaunchKernelGGL((filter_stage2_lds_kernel<4, false>), dim3(f.nqueries, G), dim3(256), (4 * 33 * sizeof(float) + 15) / 16 * 16 + 4 * 32 * 256, st,
                           f, pids, pid_stride, counts, keys, key_stride, cen16, q_hi, q_lo);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// (The score rows of the surviving centroids -- round 3's qual_rows_kernel, for the query-split stage 0 only -- are computed by
// qualifying_kernel for every batch since round 4: flmr_candidates.hip.)

int flmr_launch_filter_stage2(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride,
                              const int32_t* counts, int32_t max_count, uint64_t* keys, int64_t key_stride,
                              hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    const size_t lds = (size_t)S2_WAVES * (((f.nq_cand + 31) >> 5) * 32 + 1) * sizeof(float);
    dim3 grid(f.nqueries, (unsigned)flmr_ceil_div(max_count, S2_WAVES)), block(S2_WAVES * 64);
    hipLaunchKernelGGL(filter_stage2_kernel, grid, block, lds, st, f, pids, pid_stride, counts, keys, key_stride);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// Digit pick of one radix-select pass, by the first wave: the largest digit d whose suffix count sum_{d' >= d} hist[d']
// reaches `rem`.  Lane l owns digits 4l..4l+3; a shuffle scan gives the counts above each lane's group.  (A single thread
// walking 256 LDS bins costs ~10 us per pass -- most of the kernel for short lists.)  Returns (digit, count of that digit,
// keys still needed inside it) in all lanes of the wave.
__device__ __forceinline__ void sel_pick_digit(const unsigned int* hist, int rem, int lane, int& dsel, int& csel, int& rem_out) {
    const int h0 = (int)hist[4 * lane], h1 = (int)hist[4 * lane + 1], h2 = (int)hist[4 * lane + 2], h3 = (int)hist[4 * lane + 3];
    const int mine = h0 + h1 + h2 + h3;
    int incl = mine;  // inclusive suffix sum over lanes lane..63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(incl, d, 64);
        if (lane + d < 64) incl += o;
    }
    const int above = incl - mine;                  // keys in digits above this lane's group
    const bool here = above < rem && incl >= rem;   // exactly one lane (when rem <= total)
    int d = 0, c = 0, r = 0;
    if (here) {
        int need = rem - above;
        if (h3 >= need) { d = 4 * lane + 3; c = h3; r = need; }
        else if (h3 + h2 >= need) { d = 4 * lane + 2; c = h2; r = need - h3; }
        else if (h3 + h2 + h1 >= need) { d = 4 * lane + 1; c = h1; r = need - h3 - h2; }
        else { d = 4 * lane; c = h0; r = need - h3 - h2 - h1; }
    }
    const unsigned long long m = __ballot(here);
    const int src = m ? __builtin_ctzll(m) : 0;
    dsel = __shfl(d, src, 64); csel = __shfl(c, src, 64); rem_out = __shfl(r, src, 64);
}

// ------------------------------------------------------------------------------------------------
// Radix select: the n largest of count[q] 64-bit keys, unordered.  grid = nqueries, block = 1024.
// Up to 8 passes of 8 bits from the top; the histogram lives in LDS; the leader digit of each wave is
// aggregated with a ballot so the heavily repeated high bytes do not serialise on one LDS address.
// The first KPT * 1024 keys of a query are loaded ONCE into registers -- one block per query means one block per CU, so
// a pass that re-reads global memory is pure exposed latency (30 dependent iterations x 9 passes before); only the tail
// of longer lists streams from global memory in every pass.  The passes stop once the selected bucket is taken whole.
// ------------------------------------------------------------------------------------------------
// BAND (the dense stage 1's approximate-then-refine, flmr_stage1_dense.hip): for the queries with band_mode[q] == FLMR_S1D_IMAGE the
// keys carry upper-bound scores U; instead of the n largest, EVERY key whose score is >= (the n-th largest U) - band_err[q] is
// emitted (no cap: out_stride >= count), and band_in[q] = how many -- the exact pass rescores them, the selection proper follows.
// The other queries only copy their count to band_in.
template <int KPT, bool BAND>
__global__ __launch_bounds__(1024) void select_topn_kernel(const uint64_t* keys, int64_t key_stride,
                                                           const int32_t* counts, int32_t n, int32_t* out_pids,
                                                           int64_t out_stride, int32_t* n_out, uint64_t* out_keys,
                                                           uint64_t key_add, int32_t fixed_count, const int32_t* band_mode,
                                                           const float* band_err, int32_t* band_in) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_remaining;
    __shared__ int s_out;
    __shared__ int s_done;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int P = counts ? counts[b] : fixed_count;
    const uint64_t* kb = keys + (size_t)b * key_stride;
    int32_t* ob = out_pids ? out_pids + (size_t)b * out_stride : nullptr;  // optional: the selected pids
    uint64_t* okb = out_keys ? out_keys + (size_t)b * out_stride : nullptr;  // optional: the selected keys (+ pid base), 0 padded
    if (BAND) {
        if (band_mode[b] != FLMR_S1D_IMAGE) {   // (block-uniform)
            if (tid == 0) band_in[b] = P;
            return;
        }
        if (P <= n) {   // every candidate is selected: all of them are rescored
            for (int i = tid; i < P; i += blockDim.x) ob[i] = flmr_key_pid(kb[i]);
            if (tid == 0) { n_out[b] = P; band_in[b] = P; }
            return;
        }
    }
    if (!BAND && P <= n) {
        for (int i = tid; i < n; i += blockDim.x) {
            if (ob && i < P) ob[i] = flmr_key_pid(kb[i]);
            if (okb) okb[i] = i < P ? kb[i] + key_add : 0ull;
        }
        if (tid == 0 && n_out) n_out[b] = P;
        return;
    }
    uint64_t kreg[KPT];  // keys [0, KPT*1024) of the list; anything beyond streams from global memory in every pass
#pragma unroll
    for (int j = 0; j < KPT; j++) kreg[j] = (j * 1024 + tid < P) ? kb[j * 1024 + tid] : 0ull;
    // once the bucket that still has to be split holds <= SEL_LCAP keys they are copied to LDS and the remaining passes
    // touch nothing else
    constexpr int SEL_LCAP = 4096;
    __shared__ unsigned long long s_list[SEL_LCAP];
    __shared__ int s_nlist, s_compact;
    if (tid == 0) { s_prefix = 0ull; s_remaining = n; s_out = 0; s_done = 0; s_nlist = 0; s_compact = 0; }
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if (s_done) break;  // block-uniform (written before the previous barrier pair)
        const unsigned long long prefix = s_prefix;
        if (s_compact) {  // block-uniform
            const int nl = s_nlist;
            for (int i0 = 0; i0 < nl; i0 += 1024) {
                const int i = i0 + tid;
                const bool valid = i < nl;
                const uint64_t key = valid ? s_list[i] : 0ull;
                const bool act = valid && ((key >> (shift + 8)) == (prefix >> (shift + 8)));
                if (act) atomicAdd(&hist[(unsigned int)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                int dsel, csel, rem;
                sel_pick_digit(hist, s_remaining, lane, dsel, csel, rem);
                if (tid == 0) {
                    s_remaining = rem;
                    s_prefix = prefix | ((unsigned long long)dsel << shift);
                    if (csel == rem) s_done = 1;
                }
            }
            __syncthreads();
            continue;
        }
        auto count = [&](bool valid, uint64_t key) {
            const bool act = valid && ((pass == 0) || ((key >> (shift + 8)) == (prefix >> (shift + 8))));
            const unsigned int dg = (unsigned int)(key >> shift) & 255u;
            const unsigned long long am = __ballot(act);
            if (am) {
                const int leader = __builtin_ctzll(am);
                const unsigned int ld = (unsigned int)__shfl((int)dg, leader, 64);
                const unsigned long long same = __ballot(act && dg == ld);
                if (lane == leader) atomicAdd(&hist[ld], (unsigned int)__popcll(same));
                if (act && dg != ld) atomicAdd(&hist[dg], 1u);
            }
        };
#pragma unroll
        for (int j = 0; j < KPT; j++) count(j * 1024 + tid < P, kreg[j]);
        for (int i0 = KPT * 1024; i0 < P; i0 += 4 * 1024) {  // 4 loads in flight per thread: the tail is latency-bound too
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kk[u] = (i0 + u * 1024 + tid < P) ? kb[i0 + u * 1024 + tid] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; u++) count(i0 + u * 1024 + tid < P, kk[u]);
        }
        __syncthreads();
        if (tid < 64) {
            int dsel, csel, rem;
            sel_pick_digit(hist, s_remaining, lane, dsel, csel, rem);
            if (tid == 0) {
                s_remaining = rem;  // how many keys with the selected digit (and prefix) are still needed
                s_prefix = prefix | ((unsigned long long)dsel << shift);
                if (csel == rem) s_done = 1;  // the whole bucket is taken: every key >= the prefix (low bits 0) is selected
                else if (csel <= SEL_LCAP && pass < 7) s_compact = 1;
            }
        }
        __syncthreads();
        if (s_compact && !s_done) {  // copy the bucket (keys whose top 8*(pass+1) bits equal the new prefix) to LDS
            const unsigned long long np = s_prefix;
            auto take = [&](bool valid, uint64_t key) {
                if (valid && ((key >> shift) == (np >> shift))) s_list[atomicAdd(&s_nlist, 1)] = key;
            };
#pragma unroll
            for (int j = 0; j < KPT; j++) take(j * 1024 + tid < P, kreg[j]);
            for (int i0 = KPT * 1024; i0 < P; i0 += 4 * 1024) {
                uint64_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) kk[u] = (i0 + u * 1024 + tid < P) ? kb[i0 + u * 1024 + tid] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++) take(i0 + u * 1024 + tid < P, kk[u]);
            }
            // (the barrier at the top of the next pass orders these writes before the list is read)
        }
    }
    // s_prefix is now the n-th largest key (or the floor of the last bucket, taken whole): keep everything >= it
    // Output positions come from a block scan of per-thread counts, not from an atomic counter: the selected SET is what
    // matters to the next stage, but a reproducible ORDER (a function of the input only) lets every rank of the sharded
    // protocol derive the same list from the same gathered keys.
    // Keys above the threshold are all kept; keys EQUAL to it (one key, or many empty keys = 0 when a row holds fewer than n
    // real keys) only fill what is left -- otherwise a run of empty keys met first in scan order could displace real ones.
    const unsigned long long thr = s_prefix;
    __shared__ int sel_scan[17];
    if (BAND) {
        // thr <= the n-th largest key; every key whose score reaches (its score - err) belongs to the band
        const unsigned long long cut = flmr_make_key(flmr_key_score(thr) - band_err[b], 0);
        int mine = 0;
#pragma unroll
        for (int j = 0; j < KPT; j++) mine += (j * 1024 + tid < P && kreg[j] >= cut) ? 1 : 0;
        for (int i0 = KPT * 1024 + tid; i0 < P; i0 += 1024) mine += kb[i0] >= cut ? 1 : 0;
        int total;
        int pos = flmr_block_exclusive_scan(mine, sel_scan, &total);
#pragma unroll
        for (int j = 0; j < KPT; j++)
            if (j * 1024 + tid < P && kreg[j] >= cut) ob[pos++] = flmr_key_pid(kreg[j]);
        for (int i0 = KPT * 1024 + tid; i0 < P; i0 += 1024) {
            const uint64_t key = kb[i0];
            if (key >= cut) ob[pos++] = flmr_key_pid(key);
        }
        if (tid == 0) { n_out[b] = total; band_in[b] = total; }
        return;
    }
    int mine_gt = 0, mine_eq = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const bool v = j * 1024 + tid < P;
        mine_gt += (v && kreg[j] > thr) ? 1 : 0;
        mine_eq += (v && kreg[j] == thr) ? 1 : 0;
    }
    for (int i0 = KPT * 1024; i0 < P; i0 += 4 * 1024) {  // (the tail: 4 loads in flight)
        uint64_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kk[u] = (i0 + u * 1024 + tid < P) ? kb[i0 + u * 1024 + tid] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool v = i0 + u * 1024 + tid < P;
            mine_gt += (v && kk[u] > thr) ? 1 : 0;
            mine_eq += (v && kk[u] == thr) ? 1 : 0;
        }
    }
    int total_gt, total_eq;
    int pos_gt = flmr_block_exclusive_scan(mine_gt, sel_scan, &total_gt);
    __syncthreads();
    int pos_eq = total_gt + flmr_block_exclusive_scan(mine_eq, sel_scan, &total_eq);
    auto emit = [&](bool valid, uint64_t key) {
        if (valid && key >= thr) {
            const int pos = key > thr ? pos_gt++ : pos_eq++;
            if (pos < n) { if (ob) ob[pos] = flmr_key_pid(key); if (okb) okb[pos] = key + key_add; }
        }
    };
#pragma unroll
    for (int j = 0; j < KPT; j++) emit(j * 1024 + tid < P, kreg[j]);
    for (int i0 = KPT * 1024; i0 < P; i0 += 4 * 1024) {
        uint64_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kk[u] = (i0 + u * 1024 + tid < P) ? kb[i0 + u * 1024 + tid] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) emit(i0 + u * 1024 + tid < P, kk[u]);
    }
    if (tid == 0 && n_out) n_out[b] = (total_gt + total_eq) < n ? (total_gt + total_eq) : n;
}

int flmr_launch_select_topn(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t nqueries,
                            int32_t n, int32_t* out_pids, int64_t out_stride, int32_t* n_out, hipStream_t st,
                            uint64_t* out_keys, uint64_t key_add) {
    // keys per thread held in registers: sized from the candidate capacity (the tail of longer lists streams)
    if (key_stride <= 16 * 1024)
        hipLaunchKernelGGL((select_topn_kernel<16, false>), dim3(nqueries), dim3(1024), 0, st, keys, key_stride, counts, n, out_pids,
                           out_stride, n_out, out_keys, key_add, 0, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((select_topn_kernel<24, false>), dim3(nqueries), dim3(1024), 0, st, keys, key_stride, counts, n, out_pids,
                           out_stride, n_out, out_keys, key_add, 0, nullptr, nullptr, nullptr);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_s1_band(const uint64_t* keys, int64_t key_stride, const int32_t* counts, const int32_t* mode, const float* err,
                        int32_t nqueries, int32_t n, int32_t* band, int32_t* band_count, int32_t* in_count, hipStream_t st) {
    if (key_stride <= 16 * 1024)
        hipLaunchKernelGGL((select_topn_kernel<16, true>), dim3(nqueries), dim3(1024), 0, st, keys, key_stride, counts, n, band,
                           key_stride, band_count, nullptr, 0ull, 0, mode, err, in_count);
    else
        hipLaunchKernelGGL((select_topn_kernel<24, true>), dim3(nqueries), dim3(1024), 0, st, keys, key_stride, counts, n, band,
                           key_stride, band_count, nullptr, 0ull, 0, mode, err, in_count);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// keys [nqueries, m] (0 = empty) -> the n largest of each row, UNORDERED, 0 padded (exchange steps of the sharded protocol
// whose consumers do not care about order: a radix select instead of a full bitonic sort of up to 8192 keys)
int flmr_launch_select_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out, hipStream_t st) {
    if (m <= 16 * 1024)
        hipLaunchKernelGGL((select_topn_kernel<16, false>), dim3(nqueries), dim3(1024), 0, st, keys, (int64_t)m, nullptr, n, nullptr,
                           (int64_t)n, nullptr, out, 0ull, m, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((select_topn_kernel<24, false>), dim3(nqueries), dim3(1024), 0, st, keys, (int64_t)m, nullptr, n, nullptr,
                           (int64_t)n, nullptr, out, 0ull, m, nullptr, nullptr, nullptr);
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
    flmr_sort_keys_desc(s, npow2);
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
    hipLaunchKernelGGL(sort_topn_kernel, dim3(nqueries), dim3(1024), flmr_sort_lds_bytes(npow2), st, keys, key_stride, counts,
                       npow2, n, out_pids, out_scores, out_stride, n_out, pid_base, fill);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Stage-2 survivor selection, approximate-then-refine (the default for whole batches on the sliced kernel).
// The sliced kernel's hi products alone give every passage's stage-2 score to within E = err_sum[query] (flmr_stage0.hip:
// s0_prepare_kernel) at 84 % of the full kernel's time.  With a* = the n-th largest approximate score, a passage with
// a > a* + 2E is certainly among the n best by FULL score and one with a < a* - 2E certainly is not (full scores differ from
// the approximate ones by <= E, and so does the n-th order statistic); only the band |a - a*| <= 2E (a few per cent of the
// survivors) is rescored with both products by the gather kernel, and the n - #certain best of the band by full
// (score, pid) complete the list -- the SET filter_pids.cpp:143-157 selects, not its order: stage 3 rescores every member
// anyway and the final ranking is by those scores.  grid = nqueries, block = 1024; LDS = the sorted keys.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void s2_refine_plan_kernel(const uint64_t* keys, int64_t key_stride, const int32_t* counts,
                                                              int32_t npow2, int32_t n, const float* __restrict__ err_sum,
                                                              int32_t* out_pids, int64_t out_stride, int32_t* band_pids,
                                                              int64_t band_stride, int32_t* band_count, int32_t* need,
                                                              int32_t* def_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);   // (no static LDS: npow2 = 8192 takes all 64 KB)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cnt = counts[b];
    for (int i = tid; i < npow2; i += blockDim.x) s[i] = (i < cnt) ? keys[(size_t)b * key_stride + i] : 0ull;
    __syncthreads();
    flmr_sort_keys_desc(s, npow2);
    if (cnt <= n) {   // nothing to select: every survivor goes on
        for (int i = tid; i < cnt; i += blockDim.x) out_pids[(size_t)b * out_stride + i] = flmr_key_pid(s[i]);
        if (tid == 0) { band_count[b] = 0; need[b] = 0; def_count[b] = cnt; }
        return;
    }
    // every thread runs the two searches itself (same addresses across the block: LDS broadcasts, no barrier, no shared pair).
    // The band edges are rounded OUTWARD: a* +- 2E evaluated in fp32 may land an ulp inside the exact interval.
    const float astar = flmr_key_score(s[n - 1]), E2 = 2.0f * err_sum[b];
    const float hi = nextafterf(astar + E2, INFINITY), lo = nextafterf(astar - E2, -INFINITY);
    // sorted descending: i0 = #keys with score > hi (certainly in), i1 = #keys with score >= lo (not certainly out)
    int a = 0, z = n - 1;             // score(s[n-1]) = a* <= hi, so i0 <= n - 1
    while (a < z) { const int mid = (a + z) >> 1; if (flmr_key_score(s[mid]) > hi) a = mid + 1; else z = mid; }
    const int i0 = a;
    a = n; z = cnt;                   // score(s[n-1]) = a* >= lo, so i1 >= n
    while (a < z) { const int mid = (a + z) >> 1; if (flmr_key_score(s[mid]) >= lo) a = mid + 1; else z = mid; }
    const int i1 = a;
    for (int i = tid; i < i0; i += blockDim.x) out_pids[(size_t)b * out_stride + i] = flmr_key_pid(s[i]);
    for (int i = tid; i < i1 - i0; i += blockDim.x) band_pids[(size_t)b * band_stride + i] = flmr_key_pid(s[i0 + i]);
    if (tid == 0) { band_count[b] = i1 - i0; need[b] = n - i0; def_count[b] = i0; }
}

__global__ __launch_bounds__(1024) void s2_refine_finish_kernel(const uint64_t* band_keys, int64_t key_stride,
                                                                const int32_t* band_count, const int32_t* need,
                                                                const int32_t* def_count, int32_t npow2, int32_t* out_pids,
                                                                int64_t out_stride, int32_t* n_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cnt = band_count[b], take = need[b], base = def_count[b];
    if (tid == 0) n_out[b] = base + take;
    if (cnt == 0) return;   // (block-uniform)
    // the band is usually a few dozen passages: sort only the power of two that holds it
    int np = 2;
    while (np < cnt) np <<= 1;
    if (np > npow2) np = npow2;
    for (int i = tid; i < np; i += blockDim.x) s[i] = (i < cnt) ? band_keys[(size_t)b * key_stride + i] : 0ull;
    __syncthreads();
    flmr_sort_keys_desc(s, np);
    for (int i = tid; i < take; i += blockDim.x) out_pids[(size_t)b * out_stride + base + i] = flmr_key_pid(s[i]);
}

int flmr_launch_s2_refine_plan(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t max_count, int32_t nqueries,
                               int32_t n, const float* err_sum, int32_t* out_pids, int64_t out_stride, int32_t* band_pids,
                               int64_t band_stride, int32_t* band_count, int32_t* need, int32_t* def_count, hipStream_t st) {
    if (max_count > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "s2 refine: %d keys > %d", max_count, FLMR_MAX_NDOCS);
    int npow2 = 2;
    while (npow2 < max_count) npow2 <<= 1;
    hipLaunchKernelGGL(s2_refine_plan_kernel, dim3(nqueries), dim3(1024), flmr_sort_lds_bytes(npow2), st, keys, key_stride, counts, npow2, n,
                       err_sum, out_pids, out_stride, band_pids, band_stride, band_count, need, def_count);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_s2_refine_finish(const uint64_t* band_keys, int64_t key_stride, const int32_t* band_count, const int32_t* need,
                                 const int32_t* def_count, int32_t max_count, int32_t nqueries, int32_t* out_pids,
                                 int64_t out_stride, int32_t* n_out, hipStream_t st) {
    int npow2 = 2;
    while (npow2 < max_count) npow2 <<= 1;
    hipLaunchKernelGGL(s2_refine_finish_kernel, dim3(nqueries), dim3(1024), flmr_sort_lds_bytes(npow2), st, band_keys, key_stride, band_count,
                       need, def_count, npow2, out_pids, out_stride, n_out);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Helpers of the exact sharded protocol (SURVEY 8e "exact-parity mode"): keys travel between ranks as
// u64 = order-preserving(score) << 32 | GLOBAL pid, 0 = empty slot.
// ------------------------------------------------------------------------------------------------
// keys [nqueries, m] -> the n largest, descending, 0 padded (m <= FLMR_MAX_NDOCS: in-LDS bitonic sort)
__global__ __launch_bounds__(1024) void sort_keys_topn_kernel(const uint64_t* keys, int m, int npow2, int n, uint64_t* out,
                                                              int32_t* out_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < npow2; i += blockDim.x) sk[i] = i < m ? keys[(size_t)b * m + i] : 0ull;
    __syncthreads();
    flmr_bitonic_sort_desc<unsigned long long>(sk, npow2);
    for (int i = tid; i < n; i += blockDim.x) out[(size_t)b * n + i] = i < npow2 ? sk[i] : 0ull;
    if (out_counts && tid == 0) {
        int c = 0;
        for (int i = 0; i < n && i < npow2; i++) c += sk[i] != 0ull;
        out_counts[b] = c;
    }
}

int flmr_launch_sort_keys_topn(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out,
                               int32_t* out_counts, hipStream_t st) {
    if (m > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "topn_keys: %d keys per query > %d", m, FLMR_MAX_NDOCS);
    int npow2 = 2;
    while (npow2 < m) npow2 <<= 1;
    hipLaunchKernelGGL(sort_keys_topn_kernel, dim3(nqueries), dim3(1024), (size_t)npow2 * 8, st, keys, m, npow2, n, out,
                       out_counts);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// global keys [nqueries, n_in] -> LOCAL pids of this shard (pid in [pid_base, pid_base + num_passages)), any order
__global__ __launch_bounds__(256) void filter_local_keys_kernel(const uint64_t* keys, int n_in, int64_t pid_base,
                                                                int64_t num_passages, int32_t* out_pids, int64_t out_stride,
                                                                int32_t* out_count, int32_t* out_slot) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_in; i += blockDim.x) {
        const uint64_t key = keys[(size_t)b * n_in + i];
        const int64_t pid = (int64_t)(uint32_t)key - pid_base;
        if (key != 0ull && pid >= 0 && pid < num_passages) {
            const int pos = atomicAdd(&cnt, 1);
            if (pos < out_stride) {
                out_pids[(size_t)b * out_stride + pos] = (int32_t)pid;
                if (out_slot) out_slot[(size_t)b * out_stride + pos] = i;  // where this passage sits in the global list
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_count[b] = cnt < out_stride ? cnt : (int)out_stride;
}

int flmr_launch_filter_local_keys(const uint64_t* keys, int32_t nqueries, int32_t n_in, int64_t pid_base,
                                  int64_t num_passages, int32_t* out_pids, int64_t out_stride, int32_t* out_count,
                                  hipStream_t st, int32_t* out_slot) {
    hipLaunchKernelGGL(filter_local_keys_kernel, dim3(nqueries), dim3(256), 0, st, keys, n_in, pid_base, num_passages, out_pids,
                       out_stride, out_count, out_slot);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// internal keys [nqueries, key_stride] (count[q] valid, local pids) -> out [nqueries, n] with global pids, 0 padded
__global__ __launch_bounds__(256) void export_keys_kernel(const uint64_t* keys, int64_t key_stride, const int32_t* counts,
                                                          uint64_t key_add, int n, uint64_t* out) {
    const int b = blockIdx.x;
    const int c = counts[b];
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[(size_t)b * n + i] = i < c ? keys[(size_t)b * key_stride + i] + key_add : 0ull;
}

// the same, SLOT-ALIGNED with the global list the passages were filtered from: out[q][slot[l]] = key of local passage l,
// 0 everywhere else -- so the shards' outputs can be combined by a SUM all-reduce (one non-zero contributor per slot)
// as well as by a gather
__global__ __launch_bounds__(256) void export_keys_slotted_kernel(const uint64_t* keys, int64_t key_stride, const int32_t* counts,
                                                                  const int32_t* slot, uint64_t key_add, int n, uint64_t* out) {
    const int b = blockIdx.x;
    const int c = counts[b];
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[(size_t)b * n + i] = 0ull;
    __syncthreads();
    for (int l = threadIdx.x; l < c; l += blockDim.x) {
        const int j = slot[(size_t)b * key_stride + l];
        if (j < n) out[(size_t)b * n + j] = keys[(size_t)b * key_stride + l] + key_add;
    }
}

int flmr_launch_export_keys_slotted(const uint64_t* keys, int64_t key_stride, const int32_t* counts, const int32_t* slot,
                                    int32_t nqueries, uint64_t key_add, int32_t n, uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(export_keys_slotted_kernel, dim3(nqueries), dim3(256), 0, st, keys, key_stride, counts, slot, key_add, n, out);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_export_keys(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t nqueries, uint64_t key_add,
                            int32_t n, uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(export_keys_kernel, dim3(nqueries), dim3(256), 0, st, keys, key_stride, counts, key_add, n, out);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// sorted keys [nqueries, n] -> (pids, scores, counts) of the first k
__global__ __launch_bounds__(256) void unpack_keys_kernel(const uint64_t* keys, int n, int k, int32_t* out_pids, float* out_scores,
                                                          int32_t* out_counts) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint64_t key = i < n ? keys[(size_t)b * n + i] : 0ull;
        out_pids[(size_t)b * k + i] = key ? flmr_key_pid(key) : -1;
        out_scores[(size_t)b * k + i] = key ? flmr_key_score(key) : 0.0f;
        mine += key != 0ull;
    }
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) out_counts[b] = cnt;
}

int flmr_launch_unpack_keys(const uint64_t* keys, int32_t nqueries, int32_t n, int32_t k, int32_t* out_pids, float* out_scores,
                            int32_t* out_counts, hipStream_t st) {
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(nqueries), dim3(256), 0, st, keys, n, k, out_pids, out_scores, out_counts);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
