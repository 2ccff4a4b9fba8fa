// Candidate generation after the cell probe: IVF union -> ascending candidate pids (+ stage-1 hit flags).
//
// Reference: TPC/search/candidate_generation.py:31-37,45-64 -- ivf.lookup(cells) (segmented_lookup.cpp), pids.sort(),
// unique_consecutive.  The union of sorted pid lists is order-free, so it is computed as a passage BITMAP and the
// ascending pid list is read off the bitmap (popcount ranks): no sort, no unique.
//
// MI355X design: the pid space is cut into chunks of 32768 passages.  A workgroup owns one (query, chunk) pair, keeps
// the chunk's 4 KB bitmap in LDS, and marks it from the probed cells' IVF lists; a per-index table
// ivf_chunk_tab[c][chunk] (built once at flmr_index_open) says where chunk boundaries fall inside every list, so a
// workgroup reads exactly the list slices it needs -- no global atomics, no bitmap memset, coalesced list reads.
// The same pass marks a second bitmap from the lists of the centroids that survive the score threshold (the
// "qualifying" centroids): a candidate outside it provably has no surviving centroid, its stage-1 score is the
// all-miss value, and stage 1 never reads its codes.  On the default path the same workgroup then COMPUTES stage 1 for
// the chunk from those lists (cand_mark_score_kernel below: per-passage accumulators in LDS, keys written directly), and
// the ascending pid list (cand_emit_kernel: bitmap + per-chunk counts -> pids + one hit flag per candidate) is only built
// for the queries that keep the code-scanning stage 1, or on demand for FLMR_TAP_CANDIDATES.
#include "flmr_device.h"

#define CAND_CHUNK_WORDS 1024                      // 32768 passages per chunk
#define CAND_CHUNK_PIDS (CAND_CHUNK_WORDS * 32)
#define CF_KCAP_LIMIT 3072     // = CF_KCAP, CF_QCAP, CF_DSLOTS below (static_assert there): what cand_plan_kernel plans against
#define CF_QCAP_LIMIT 256
#define CF_DSLOTS_LIMIT 256

// ---- per-index chunk table: position of the first pid >= chunk*32768 inside every IVF list ----------------------
__global__ void build_chunk_table_kernel(const int32_t* ivf_pids, const int64_t* ivf_offsets, int K, int nchunks,
                                         uint32_t* tab /* [K][nchunks+1] */) {
    const int64_t total = (int64_t)K * (nchunks + 1);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / (nchunks + 1)), ch = (int)(e % (nchunks + 1));
        const int64_t beg = ivf_offsets[c], end = ivf_offsets[c + 1];
        const int64_t target = (int64_t)ch * CAND_CHUNK_PIDS;
        int64_t lo = beg, hi = end;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ivf_pids[mid] < target) lo = mid + 1; else hi = mid;
        }
        tab[e] = (uint32_t)(lo - beg);
    }
}

int flmr_build_chunk_table(const int32_t* ivf_pids, const int64_t* ivf_offsets, int K, int64_t num_passages,
                           uint32_t** out_tab, int32_t* out_nchunks) {
    const int nchunks = (int)flmr_ceil_div(num_passages > 0 ? num_passages : 1, CAND_CHUNK_PIDS);
    uint32_t* tab = nullptr;
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(&tab), (size_t)K * (nchunks + 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(build_chunk_table_kernel, dim3(2048), dim3(256), 0, 0, ivf_pids, ivf_offsets, K, nchunks, tab);
    FLMR_HIP(hipDeviceSynchronize());
    *out_tab = tab;
    *out_nchunks = nchunks;
    return FLMR_OK;
}

// ---- qualifying centroids of each query: compact list of the set bits of idx + their score rows ------------------------------
// grid = nqueries, block = 1024.  qual[q][j] = the j-th surviving centroid (ascending), j < qmax (= the searcher's row capacity);
// idx_prefix[q][w] = number of surviving centroids before idx word w (so rank(c) = prefix[c >> 5] + popc(word & below(c)));
// hit_valid[q] = 1 when stage 1 can be computed from the surviving centroids' IVF lists: at most `smax` of them (the scatter
// kernel's list ids) whose lists are not longer than `ratio` times the probed-cell lists: 2 when the hit set only prefilters
// the code-scanning stage 1 (beyond that marking costs more than it saves), 8 when stage 1 itself is computed from those lists
// (32 LDS atomics per (centroid, passage) pair against ~128 code reads per candidate).
// rows_out != NULL (the sparse score table in its compact form): the 16 waves then compute the 32-column score row of every
// listed centroid with the stage-0 fp16-split MFMA sequence (bitwise the values stage 0 computes: an output element depends
// only on its A row and B column) into rows_out[q][j][0..31] -- the only score rows stages 1 of this query ever read.  Stage 0
// itself stores none: the per-query K x 32 table (16.8 MB at K = 131072) no longer exists on this path.
__global__ __launch_bounds__(1024) void qualifying_kernel(const uint32_t* idx_bits, int idx_words,
                                                          const int64_t* ivf_offsets, const int32_t* cells,
                                                          const int32_t* ncell, int max_cells, int32_t* qual,
                                                          int32_t* nqual, int qmax, int smax, int32_t* hit_valid,
                                                          int32_t* key_count, int ratio, uint32_t* idx_prefix,
                                                          float* rows_out, const _Float16* __restrict__ cen16,
                                                          const _Float16* __restrict__ q_hi, const _Float16* __restrict__ q_lo,
                                                          int32_t* row_ovf, int32_t* fast_state, int32_t* any_zero) {
    __shared__ int scan_lds[17];
    __shared__ unsigned long long tot_q, tot_c;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { tot_q = 0ull; tot_c = 0ull; }
    // (cand_plan_kernel, a later launch, ORs into [0] and [1]; [2 .. 9] are the exact pass's item counters, one per XCD)
    if (b == 0 && tid < 16 && any_zero) any_zero[tid] = 0;
    __syncthreads();
    int base = 0;
    unsigned long long mylen = 0;
    for (int w0 = 0; w0 < idx_words; w0 += 1024) {
        const int w = w0 + tid;
        uint32_t bits = (w < idx_words) ? idx_bits[(size_t)b * idx_words + w] : 0u;
        int total;
        int pos = base + flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
        if (idx_prefix && w < idx_words) idx_prefix[(size_t)b * idx_words + w] = (uint32_t)pos;
        while (bits) {
            const int c = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1;
            if (pos < qmax) qual[(size_t)b * qmax + pos] = c;
            pos++;
            mylen += (unsigned long long)(ivf_offsets[c + 1] - ivf_offsets[c]);
        }
        base += total;
    }
    unsigned long long clen = 0;
    for (int e = tid; e < ncell[b]; e += 1024) {
        const int c = cells[(size_t)b * max_cells + e];
        clen += (unsigned long long)(ivf_offsets[c + 1] - ivf_offsets[c]);
    }
    atomicAdd(&tot_q, mylen);
    atomicAdd(&tot_c, clen);
    __syncthreads();
    const int n = base < qmax ? base : qmax;
    if (tid == 0) {
        nqual[b] = n;
        hit_valid[b] = (base <= smax) && (base <= qmax) && (tot_q <= (unsigned long long)ratio * tot_c);
        if (key_count) key_count[b] = 0;
        if (fast_state) { fast_state[FLMR_FAST_HDR + b] = 1; fast_state[FLMR_FAST_HDR + gridDim.x + b] = 0; }   // (plan word: cand_plan_kernel; the fast forms' key counter)
        if (row_ovf) row_ovf[b] = (rows_out && base > qmax) ? 1 : 0;   // more surviving centroids than score rows: stage 1 is recomputed
    }
    if (!rows_out || n == 0) return;   // (block-uniform)
    // ---- the listed centroids' score rows, one 32-row MFMA tile per wave at a time ---------------------------------------
    // (the list was written by this block: read it back past the L1, which may still hold a line of an earlier batch's list)
    __threadfence_block();
    const int lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    const int ntiles = (n + 31) >> 5;
    if (wave >= ntiles) return;
    f16x8 bh[8], bl[8];
    {
        const f16x8* ph = reinterpret_cast<const f16x8*>(q_hi + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
        const f16x8* pl = reinterpret_cast<const f16x8*>(q_lo + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    float* const rows_b = rows_out + (size_t)b * qmax * 32;
    for (int t = wave; t < ntiles; t += 16) {
        const int e = t * 32 + i < n ? t * 32 + i : n - 1;
        const int c = __hip_atomic_load(qual + (size_t)b * qmax + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const f16x8* pc = reinterpret_cast<const f16x8*>(cen16 + (size_t)c * FLMR_DIM + 64 * h);
        f16x8 av[8];
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
            const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < n) rows_b[(size_t)row * 32 + i] = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
        }
    }
}

// ---- kernel A: mark the chunk bitmaps in LDS -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_mark_chunks_kernel(const int32_t* cells, const int32_t* ncell, int max_cells,
                                                               const int32_t* qual, const int32_t* nqual, int qmax,
                                                               const int32_t* hit_valid, const int32_t* ivf_pids,
                                                               const int64_t* ivf_offsets, const uint32_t* tab, int nchunks,
                                                               uint32_t* cand_bits, uint32_t* hit_bits, int64_t words,
                                                               int32_t* chunk_cnt) {
    __shared__ uint32_t cb[CAND_CHUNK_WORDS], hb[CAND_CHUNK_WORDS];
    __shared__ int cnt_lds;
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < CAND_CHUNK_WORDS; w += 256) { cb[w] = 0u; hb[w] = 0u; }
    if (tid == 0) cnt_lds = 0;
    __syncthreads();
    const int nl = ncell[b];
    const int nq = hit_valid[b] ? nqual[b] : 0;
    const int pid0 = ch * CAND_CHUNK_PIDS;
    for (int l = wave; l < nl + nq; l += 4) {  // one IVF list slice per wave at a time, lanes over its entries
        const bool is_cell = l < nl;
        const int c = is_cell ? cells[(size_t)b * max_cells + l] : qual[(size_t)b * qmax + (l - nl)];
        const int64_t beg = ivf_offsets[c];
        const uint32_t s = tab[(size_t)c * (nchunks + 1) + ch], e = tab[(size_t)c * (nchunks + 1) + ch + 1];
        uint32_t* dst = is_cell ? cb : hb;
        for (uint32_t x = s + lane; x < e; x += 64) {
            const int pid = ivf_pids[beg + x] - pid0;
            atomicOr(&dst[pid >> 5], 1u << (pid & 31));
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int w = tid; w < CAND_CHUNK_WORDS; w += 256) {
        const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + w;
        if (gw < words) {
            cand_bits[(size_t)b * words + gw] = cb[w];
            hit_bits[(size_t)b * words + gw] = hb[w];
            cnt += __popc(cb[w]);
        }
    }
    atomicAdd(&cnt_lds, cnt);
    __syncthreads();
    if (tid == 0) chunk_cnt[(size_t)b * nchunks + ch] = cnt_lds;
}

__device__ __forceinline__ int cf_wave_sum_early(int x) {   // (= cf_wave_sum below: row sums by DPP, the four rows on the scalar unit)
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ int64_t s1s_bcast64(int64_t v, int j);

// ---- which form of the list scatter takes a query: decided from the query's own data, before any form runs ----------------------
// The three forms (cand_fast_kernel's queue, cand_dense_small_kernel, cand_mark_score_kernel's slots) have hard per-chunk limits --
// staged hit candidates, queued pairs of passages with several surviving centroids, slots -- and a form that runs into one hands
// the whole query over after the work is done.  Which limits a query meets depends on how its surviving lists overlap inside the
// candidates, which list LENGTHS do not say (1 k pairs per chunk are 15 collisions on the planted corpus and 1.2 k on an index built
// from overlapping clusters); so the statistics are MEASURED on a sample: two of the query's 32768-passage chunks are marked exactly
// as the forms mark them (candidate bitmap from the probed cells' slices, hit / several bitmaps from the surviving lists' slices)
// and counted -- hit candidates H, pairs Q that belong to candidates with several surviving centroids.  The plan word
// fast_state[b] = 0 queue form | FLMR_PLAN_SMALL small-dense form | 1 slot form is a function of the query and the index alone: no
// searcher-lifetime counters, the first batch of a workload runs like the thousandth, and a batch may mix all forms.  (A form that
// still overflows in a chunk the sample did not see hands over as before: results never depend on the plan.)
// grid = nqueries, block = 1024: a wave takes the lists wave, wave + 16, ...; their (offset, chunk-table) entries are fetched one list
// per lane, and the slices' entries eight lists at a time into registers, where they stay from the marking to the count (a list a
// wave takes beyond its first eight is re-read): the kernel is a handful of dependent memory round trips per query, ~15 us a launch.
#define FLMR_PLAN_SMALL 8
#define PLAN_WAVES 16
#define PLAN_REGS 8
__global__ __launch_bounds__(64 * PLAN_WAVES) void cand_plan_kernel(flmr_cand_args a) {
    __shared__ uint32_t cb[CAND_CHUNK_WORDS], hb[CAND_CHUNK_WORDS], sb[CAND_CHUNK_WORDS];
    __shared__ int s_h, s_q;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = a.ncell[b], nq = a.nqual[b];
    int32_t* const plan = a.fast_state + FLMR_FAST_HDR + b;
    if (a.s1d_mode && tid == 0) {   // which dense form takes the query, if any (= s1_dense_modes_kernel: one launch less per batch)
        int m = 0 /* FLMR_S1D_SKIP */, sk = 1;
        if (!a.hit_valid[b] && !(a.row_ovf && a.row_ovf[b])) {
            if (nq <= a.s1d_img_rows) m = 1 /* FLMR_S1D_IMAGE */;
            else if (a.s1d_exact_too) m = 2 /* FLMR_S1D_EXACT */;
            else sk = 0;   // the scan takes the query
        }
        a.s1d_mode[b] = m;
        a.s1d_scan_skip[b] = sk;
        if (m == 1) atomicOr(a.s1d_any + 0, 1);
        if (m != 0) atomicOr(a.s1d_any + 1, 1);
    }
    if (!a.hit_valid[b] || nl > 512 || nq > 512) {   // (block-uniform) not a list-scatter query / more lists than the fast forms index
        if (tid == 0) *plan = 1;
        return;
    }
    const int ntot = nl + nq;
    const int mine = wave < ntot ? (ntot - wave + PLAN_WAVES - 1) / PLAN_WAVES : 0;   // lists of this wave (<= 64)
    // lane j: list wave + 16 j of the query (cells first, then the surviving lists)
    int64_t beg = 0;
    int is_cell = 0;
    size_t trow = 0;
    if (lane < mine) {
        const int l = wave + PLAN_WAVES * lane;
        is_cell = l < nl;
        const int c = is_cell ? a.cells[(size_t)b * a.max_cells + l] : a.qual[(size_t)b * a.qmax + (l - nl)];
        beg = a.ivf_offsets[c];
        trow = (size_t)c * (a.nchunks + 1);
    }
    const unsigned long long cellmask = __ballot(is_cell != 0);
    int hmax = 0, qmax = 0;
    const int nsample = a.nchunks < 2 ? a.nchunks : 2;
    for (int sidx = 0; sidx < nsample; sidx++) {
        const int ch = (int)(((unsigned)b * 2654435761u >> 8) % (unsigned)a.nchunks + (unsigned)sidx * ((unsigned)a.nchunks / 2u)) % a.nchunks;
        const int pid0 = ch * CAND_CHUNK_PIDS;
        uint32_t s0 = 0, e0 = 0;
        if (lane < mine) { s0 = a.chunk_tab[trow + ch]; e0 = a.chunk_tab[trow + ch + 1]; }
        __syncthreads();
        for (int w = tid; w < CAND_CHUNK_WORDS; w += 64 * PLAN_WAVES) { cb[w] = 0u; hb[w] = 0u; sb[w] = 0u; }
        if (tid == 0) { s_h = 0; s_q = 0; }
        __syncthreads();
        // the first 64 entries of the slices of the wave's first PLAN_REGS lists (-1: no entry)
        int r[PLAN_REGS];
#pragma unroll
        for (int u = 0; u < PLAN_REGS; u++) {
            r[u] = -1;
            if (u < mine) {   // (wave-uniform)
                const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)s0, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)e0, u);
                if (sv + lane < ev) r[u] = a.ivf_pids[s1s_bcast64(beg, u) + sv + lane] - pid0;
            }
        }
        auto mark = [&](int p, bool cell) {
            const uint32_t bit = 1u << (p & 31);
            if (cell) atomicOr(&cb[p >> 5], bit);
            else if (atomicOr(&hb[p >> 5], bit) & bit) atomicOr(&sb[p >> 5], bit);
        };
        auto rest = [&](int u, bool count, int& q) {   // entries of list u not held in registers: beyond the first 64, or lists beyond PLAN_REGS
            const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)s0, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)e0, u);
            const bool cell = (cellmask >> u) & 1ull;
            const int64_t bg = s1s_bcast64(beg, u);
            for (uint32_t x = sv + (u < PLAN_REGS ? 64u : 0u) + lane; x < ev; x += 64) {
                const int p = a.ivf_pids[bg + x] - pid0;
                if (!count) mark(p, cell);
                else if (!cell) q += (int)((cb[p >> 5] & sb[p >> 5]) >> (p & 31)) & 1;
            }
        };
        int q = 0;
#pragma unroll
        for (int u = 0; u < PLAN_REGS; u++)
            if (r[u] >= 0) mark(r[u], (cellmask >> u) & 1ull);
        for (int u = 0; u < mine; u++) rest(u, false, q);
        __syncthreads();
        int h = 0;
        for (int w = tid; w < CAND_CHUNK_WORDS; w += 64 * PLAN_WAVES) h += __popc(cb[w] & hb[w]);
        // the pairs the queue form would queue: (surviving list, passage) with the passage a candidate that holds several of them
#pragma unroll
        for (int u = 0; u < PLAN_REGS; u++)
            if (r[u] >= 0 && !((cellmask >> u) & 1ull)) q += (int)((cb[r[u] >> 5] & sb[r[u] >> 5]) >> (r[u] & 31)) & 1;
        for (int u = 0; u < mine; u++) rest(u, true, q);
        h = cf_wave_sum_early(h);
        q = cf_wave_sum_early(q);
        if (lane == 0) { atomicAdd(&s_h, h); atomicAdd(&s_q, q); }
        __syncthreads();
        hmax = hmax > s_h ? hmax : s_h;
        qmax = qmax > s_q ? qmax : s_q;
    }
    if (tid == 0) {
        // margins under the forms' limits (CF_KCAP staged keys, CF_QCAP queued pairs, CF_DSLOTS slots): the chunks not sampled vary
        int p = 1;
        if (hmax <= (CF_KCAP_LIMIT * 7) / 10 && qmax <= (CF_QCAP_LIMIT * 5) / 8) p = 0;
        else if (hmax <= (CF_DSLOTS_LIMIT * 7) / 10) p = FLMR_PLAN_SMALL;
        *plan = p;
    }
}

// ---- kernel A': mark + STAGE 1 BY SCATTER ---------------------------------------------------------------------------------
// The stage-1 score of a passage (filter_pids.cpp:27-69 with the idx mask of index_storage.py:116) depends only on the SET
// of surviving centroids that occur in it: sum_k max_{c in set} centroid_scores[c][k] (per-k maxima start at -9999, summed
// in ascending k).  "Centroid c occurs in passage p" is exactly "p is in ivf[c]" (the IVF is built from the codes,
// indexing/utils.py:8-53), so the per-passage sets can be read off the surviving centroids' IVF lists and the passages'
// codes never have to be scanned: this replaces the 4*sum(doclen) bytes/candidate code scan -- the largest term of the
// whole path's compulsory traffic in the reference formulation -- by the (c, pid) pairs of the surviving lists.
// Per (query, chunk) workgroup, after the bitmaps are marked: every candidate that is also in the hit set gets a slot
// (popcount rank inside the chunk).  The surviving lists' slices are walked a second time (from registers) and every
// (c, pid) pair adds (1, list id) to its slot's counter with ONE returning LDS atomic.  A passage with a single surviving
// centroid -- nearly all of them: 1.02 lists per hit passage on the bench corpus -- scores that list's constant
// sum_k max(-9999, cs[c][k]) (computed once per workgroup); only the pairs of passages with several surviving centroids are
// queued and fold their 32-column score rows into the slot's accumulator row with ds_max (scores order-encoded as ints, so
// the integer max is exact), and those slots sum their columns in ascending k.  A window whose queue overflows falls back
// to the dense form (every pair folds its row).  Keys go to keys[b][base + rank] with base from a per-query atomic
// counter: the top-ndocs selection is order-free.  More hits in a chunk than S1S_SLOTS are handled in windows of slots.
// Only candidates in the hit set get a key here (rank among the hits); a passage with one surviving centroid is scored
// inline in the key pass, the multi-centroid ones are written from the queue, which knows slot and passage of each.
// (Measured with the per-phase clock probe -DS1S_PROFILE: the dense form spent 8.4 k of 27 k clocks per chunk in the 32
// ds_max per pair and 3.3 k initialising accumulators; this form 21 k clocks per chunk, spread over seven
// barrier-separated phases of 1.5-6 k each -- the workgroup's 16 waves are issue- and barrier-bound, not LDS- or
// HBM-bound.)
#define S1S_SLOTS 1088   // slots per window: with the bitmaps, list constants and queue this fills the CU's 160 KB of LDS
#define S1S_STRIDE 33
#define S1S_QCAP 512     // (slot, list) pairs of passages with more than one surviving centroid, per window
#define S1S_IDBITS 10    // list index inside the query's surviving lists (< qmax = 1024)
#define S1S_WAVES 16     // one 1024-thread workgroup per CU

__device__ __forceinline__ int s1s_enc(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float s1s_dec(int e) { return __int_as_float(e ^ ((e >> 31) & 0x7fffffff)); }

// IVF list slices of this (query, chunk) owned by one wave: list ids wave, wave+16, ... (at most 64 of them: the
// launcher guarantees n <= 1024), one per lane, so that the dependent chain  id -> (offset, chunk table) -> pids
// costs its first two global round trips once per wave instead of once per list.
struct s1s_slices {
    int c;            // centroid id of this lane's list
    int64_t beg;      // ivf_offsets[c]
    uint32_t s, e;    // slice [s, e) of the list inside this chunk
    int n;            // number of lists of this wave (wave-uniform)
};
__device__ __forceinline__ s1s_slices s1s_load_slices(const int32_t* ids, int n_total, int wave, int lane,
                                                      const int64_t* ivf_offsets, const uint32_t* tab, int nchunks, int ch) {
    s1s_slices m;
    m.n = wave < n_total ? (n_total - wave + S1S_WAVES - 1) / S1S_WAVES : 0;
    m.c = 0; m.beg = 0; m.s = 0; m.e = 0;
    if (lane < m.n) {
        m.c = ids[wave + S1S_WAVES * lane];
        m.beg = ivf_offsets[m.c];
        m.s = tab[(size_t)m.c * (nchunks + 1) + ch];
        m.e = tab[(size_t)m.c * (nchunks + 1) + ch + 1];
    }
    return m;
}
__device__ __forceinline__ int64_t s1s_bcast64(int64_t v, int j) {
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, j), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// End of a RARE loop body that loads (lists beyond a wave's first four, slices longer than 64 entries): an explicit
// vmcnt(0) on every path.  The compiler's wait-count bookkeeping is per physical register and merges loop paths
// conservatively: without this, scratch registers that were load destinations inside such a loop count as pending at the
// loop exits, and every later write to them in the COMMON path gets a vmcnt(0) -- which would wait for the next chunk's
// prefetched groups.
#define S1S_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)

// barrier that orders LDS only: global loads (the next chunk's prefetch) and stores stay in flight across it
__device__ __forceinline__ void s1s_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// exclusive scan over the 1024-thread block with ONE barrier: every wave scans the 16 wave totals itself (`lds`: 16 ints,
// not reused before another barrier)
__device__ __forceinline__ int s1s_block_scan(int x, int* lds, int lane, int wave, int* total) {
    const int inc = flmr_wave_inclusive_scan(x, lane);
    if (lane == 63) lds[wave] = inc;
    s1s_sync();
    const int w = lane < S1S_WAVES ? lds[lane] : 0;
    const int winc = flmr_wave_inclusive_scan(w, lane);
    *total = __builtin_amdgcn_readlane(winc, S1S_WAVES - 1);
    return inc - x + __builtin_amdgcn_readlane(winc - w, wave);
}

#ifdef S1S_PROFILE   // development only: per-phase clocks of wave 0, summed over the grid and printed by the launcher
__device__ unsigned long long s1s_prof[12];
#define S1S_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define S1S_STAMP(k) do { } while (0)
#endif

// A workgroup takes `cpb` consecutive chunks of one query: the list ids and offsets are fetched once, and only the chunk
// table entry that ends the NEXT chunk's slice is loaded per chunk (a chunk's slice starts where the previous one ended),
// requested a whole chunk ahead.
// F16: FLMR_NUMERICS_GPU_FP16 (column maxima and sums rounded to fp16, flmr_device.h); a template parameter so that the
// default instantiation carries none of it (as a kernel argument the selects cost this register-bound kernel 0.8 ms per step)
template <bool F16>
__global__ __launch_bounds__(64 * S1S_WAVES) void cand_mark_score_kernel(flmr_cand_args a, int cpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cb = reinterpret_cast<uint32_t*>(smem);                   // candidate bitmap of the chunk
    uint32_t* hb = cb + CAND_CHUNK_WORDS;                               // hit-set bitmap
    uint16_t* cbase = reinterpret_cast<uint16_t*>(hb + CAND_CHUNK_WORDS);  // exclusive candidate count before word w
    uint16_t* hbase = cbase + CAND_CHUNK_WORDS;                         // exclusive (candidate & hit) count before word w
    int* acc = reinterpret_cast<int*>(hbase + CAND_CHUNK_WORDS);        // [S1S_SLOTS][S1S_STRIDE] (+ 96 scratch words)
    float* rsum = reinterpret_cast<float*>(acc + S1S_SLOTS * S1S_STRIDE + 96);   // [1024] stage-1 score of a passage whose only surviving centroid is list j
    int* queue = reinterpret_cast<int*>(rsum + 1024);                  // [S1S_QCAP] slot << S1S_IDBITS | list
    uint16_t* qpid = reinterpret_cast<uint16_t*>(queue + S1S_QCAP);    // [S1S_QCAP] the queued pair's passage (inside the chunk)
    __shared__ int scan_lds[S1S_WAVES];
    __shared__ int s_base, s_qn;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: slice bounds stay in SGPRs
    const int k = lane & 31;
    const int ch0 = blockIdx.y * cpb;
    const int ch_end = ch0 + cpb < a.nchunks ? ch0 + cpb : a.nchunks;
    if (a.fast_state) {   // the queue form ran first: this kernel takes the queries it handed over (block-uniform)
        const int r = a.fast_state[FLMR_FAST_HDR + b];
        if (blockIdx.y == 0 && tid == 0) {
            if (r == 0 || r == 3) a.key_count[b] = a.fast_state[FLMR_FAST_HDR + a.nqueries + b];
        }
        if (r == 0 || r == 3) return;   // done by the queue form / by the small-dense form
    }
    const int nl = a.ncell[b];
    const bool scatter = a.hit_valid[b] != 0;
    const int nq = scatter ? a.nqual[b] : 0;
    s1s_slices mc = s1s_load_slices(a.cells + (size_t)b * a.max_cells, nl, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
    s1s_slices mq = s1s_load_slices(a.qual + (size_t)b * a.qmax, nq, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
#ifdef S1S_PROFILE
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int init = s1s_enc(-9999.0f);
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq_cand;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;  // <= 32 on this path
    const float* cs_b = a.cs + (size_t)b * a.cs_query_stride;
    uint64_t* keys_b = a.keys + (size_t)b * a.cand_cap;

    // four lists at a time: slice bounds (wave-uniform) + the first 64 entries of each, relative to the chunk's first pid.
    // `ls` / `le` hold the slice bounds of list j in lane j (this chunk's, or the next chunk's when prefetching).
    // Every lane of a non-empty slice loads (lanes past the end repeat the last entry) and the values are only
    // interpreted where they are consumed: a load inside a DIVERGENT branch is waited for before the branch ends, which
    // would serialise the four round trips of a group.
    struct grp { int raw[4]; uint32_t sv[4], ev[4]; };
    struct begs { const int32_t* p[4]; };   // list starts of a group (wave-uniform, the same in every chunk)
    auto list_begs = [&](const s1s_slices& m, int j0) {
        begs b;
#pragma unroll
        for (int u = 0; u < 4; u++) b.p[u] = a.ivf_pids + s1s_bcast64(m.beg, (j0 + u < m.n) ? j0 + u : (j0 < m.n ? j0 : 0));
        return b;
    };
    auto issue = [&](const s1s_slices& m, const begs& bg, uint32_t ls, uint32_t le, int j0, grp& g) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = (j0 + u < m.n) ? j0 + u : (j0 < m.n ? j0 : 0);
            g.sv[u] = (uint32_t)__builtin_amdgcn_readlane((int)ls, j);
            g.ev[u] = (j0 + u < m.n) ? (uint32_t)__builtin_amdgcn_readlane((int)le, j) : g.sv[u];
            g.raw[u] = 0;
            if (g.sv[u] < g.ev[u]) {   // wave-uniform; lanes past the end repeat the slice's last entry
                const uint32_t x = g.sv[u] + lane;
                g.raw[u] = bg.p[u][x < g.ev[u] ? x : g.ev[u] - 1u];   // uniform base + 32-bit lane offset
            }
        }
    };
    // entry `lane` of slice u relative to the chunk's first pid, -1 past the slice's end
    auto pid_of = [&](const grp& g, int u, int pid0) { return (g.sv[u] + lane < g.ev[u]) ? g.raw[u] - pid0 : -1; };
    // The first group of the probed cells' lists and of the surviving lists of a chunk is requested a chunk ahead (during
    // the previous chunk's scatter phase), and the surviving group's entries are KEPT for the scatter phase, so in the
    // common case (<= 4 lists per wave, <= 64 entries per slice) a chunk waits for no global load of its own; the score
    // rows of the wave's first four surviving centroids do not depend on the chunk and are fetched once.
    grp gc, gq;
    const begs bc0 = list_begs(mc, 0), bq0 = list_begs(mq, 0);
    issue(mc, bc0, mc.s, mc.e, 0, gc);
    issue(mq, bq0, mq.s, mq.e, 0, gq);
    // Stage-1 score of a passage whose ONLY surviving centroid is list j (the usual case: ~1.02 lists per hit passage):
    // its 32 column maxima are that centroid's score row (floored at the accumulators' start value), so their
    // ascending-k sum is a per-list constant.  Lane l of wave w owns list w + 16 l.
    if (scatter && lane < mq.n) {
        const float4* r4 = reinterpret_cast<const float4*>(cs_b + (size_t)(a.cs_compact ? wave + S1S_WAVES * lane : mq.c) * 32);
        float sc = 0.0f;
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 v = r4[q4];
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = s1s_enc(x[i]);
                const float m = s1s_dec(e > init ? e : init);
                sc += q4 * 4 + i < nqc ? (F16 ? flmr_round_f16(m) : m) : 0.0f;
            }
        }
        rsum[wave + S1S_WAVES * lane] = F16 ? flmr_round_f16(sc) : sc;   // (read after the first chunk's barriers)
    }

    // Passages with MANY surviving centroids (a corpus whose clusters overlap: ~19 per hit passage on profiles/built_index_probe's
    // 4096-topic corpus, 1.02 on the planted one) overflow the queue of multi-centroid pairs in every window; the window then
    // takes the dense form (every pair folds its 32-column row).  Once a workgroup has seen that, its later chunks skip the
    // counting pass and go dense at once (dense_hint, block-uniform).  (Folding the first group from the registers it was
    // prefetched into, and keeping the first four lists' rows across chunks, cost the common path more in scalar spills --
    // this kernel is register-bound -- than they saved the dense one.)
    bool dense_hint = false;
    // The chunk bitmaps and the slots' pair counters start at zero and are LEFT at zero by whoever reads them last (the
    // bitmap words by their owner thread in the chunk's last key pass, a counter by the thread that turns it into a key),
    // so a chunk has no initialisation phase of its own.
    cb[tid] = 0u; hb[tid] = 0u;  // CAND_CHUNK_WORDS == blockDim.x
    if (scatter) {
        for (int sl = tid; sl < S1S_SLOTS; sl += 64 * S1S_WAVES) acc[sl * S1S_STRIDE + 32] = 0;
        if (tid == 0) s_qn = 0;
    }
    s1s_sync();
    S1S_STAMP(0);
    for (int ch = ch0; ch < ch_end; ch++) {
    const int pid0 = ch * CAND_CHUNK_PIDS;
    // the end of the NEXT chunk's slices, a chunk ahead of its use
    uint32_t mc_e2 = 0, mq_e2 = 0;
    if (ch + 1 < ch_end) {
        if (lane < mc.n) mc_e2 = a.chunk_tab[(size_t)mc.c * (a.nchunks + 1) + ch + 2];
        if (lane < mq.n) mq_e2 = a.chunk_tab[(size_t)mq.c * (a.nchunks + 1) + ch + 2];
    }
    S1S_STAMP(1);
    auto consume = [&](const grp& g, const begs& bg, uint32_t* dst) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p0 = pid_of(g, u, pid0);
            if (p0 >= 0) atomicOr(&dst[p0 >> 5], 1u << (p0 & 31));
            for (uint32_t x = g.sv[u] + 64 + lane; x < g.ev[u]; x += 64) {
                const int pid = bg.p[u][x] - pid0;
                atomicOr(&dst[pid >> 5], 1u << (pid & 31));
                S1S_DRAIN();
            }
        }
    };
    // the only loads in flight here are this chunk's first groups (and, in the first chunk, the score rows): wait for them
    // in ONE place on every path, so that no later use inside a divergent branch asks for vmcnt(0) again and thereby
    // waits for the NEXT chunk's groups requested below
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    S1S_STAMP(2);
    {
        grp gt;
        consume(gc, bc0, cb);
        for (int j0 = 4; j0 < mc.n; j0 += 4) { const begs bg = list_begs(mc, j0); issue(mc, bg, mc.s, mc.e, j0, gt); consume(gt, bg, cb); S1S_DRAIN(); }
        consume(gq, bq0, hb);
        for (int j0 = 4; j0 < mq.n; j0 += 4) { const begs bg = list_begs(mq, j0); issue(mq, bg, mq.s, mq.e, j0, gt); consume(gt, bg, hb); S1S_DRAIN(); }
    }
    s1s_sync();
    S1S_STAMP(3);
    // bitmaps out (the ascending candidate list is still produced by cand_emit_kernel) + the two popcount prefixes
    const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + tid;
    uint32_t cw = cb[tid];
    const uint32_t hw = hb[tid];
    if (gw < a.words) {
        a.cand_bits[(size_t)b * a.words + gw] = cw;
        a.hit_bits[(size_t)b * a.words + gw] = hw;
    } else {
        cw = 0u;
    }
    const uint32_t hcw = cw & hw;
    // one block scan for both ranks: candidates in the low half, candidate-and-hit in the high half (<= 32768 each)
    int tot;
    const int both = s1s_block_scan(__popc(cw) | (__popc(hcw) << 16), scan_lds, lane, wave, &tot);
    const int cpos = both & 0xffff, hpos = (int)((uint32_t)both >> 16);
    const int cnt = tot & 0xffff, nh = (int)((uint32_t)tot >> 16);
    hbase[tid] = (uint16_t)hpos;
    // where this chunk's keys go: a global atomic whose ~2 us round trip is only awaited right before the keys are written
    // (only the candidates in the hit set get a key here: the others all score the all-miss constant and are appended by
    // cand_emit_kernel for the rare query with fewer hits than the selection keeps)
    int my_base = 0;
    if (tid == 0) {
        a.chunk_cnt[(size_t)b * a.nchunks + ch] = cnt;
        a.chunk_hits[(size_t)b * a.nchunks + ch] = scatter ? nh : 0;
        if (scatter && nh) {
            // an address the compiler cannot prove uniform: keeps its atomic optimiser (wave reduction + readfirstlane of
            // the result right behind the atomic, i.e. an immediate wait) away from this single-lane atomic
            int zero = 0;
            asm volatile("" : "+v"(zero));
            my_base = atomicAdd(&a.key_count[b] + zero, nh);
        }
    }
    s1s_sync();
    S1S_STAMP(4);
    grp gcn = gc, gqn = gq;   // the next chunk's first groups: its slices start where this chunk's ended
    auto prefetch_next = [&]() {
        if (ch + 1 < ch_end) {
            issue(mc, bc0, mc.e, mc_e2, 0, gcn);
            issue(mq, bq0, mq.e, mq_e2, 0, gqn);
        }
    };
    prefetch_next();
    S1S_STAMP(5);
    // (queries without `scatter` leave stage 1 to the scanning kernel; both conditions are block-uniform)
    if (scatter && nh > 0) {
    for (int win0 = 0; win0 < nh; win0 += S1S_SLOTS) {
        const int nslot = (nh - win0) < S1S_SLOTS ? (nh - win0) : S1S_SLOTS;
        // (a) the slot's padding word counts the (list, passage) pairs that land on it: count << S1S_IDBITS + sum of list
        // ids (zero on entry: see the note before the chunk loop)
        auto slot_of = [&](int pid) {
            int slot = -1;
            if (pid >= 0) {
                const int w = pid >> 5, bit = pid & 31;
                const uint32_t cwd = cb[w];
                if ((cwd >> bit) & 1u) slot = (int)hbase[w] + __popc(cwd & hb[w] & ((1u << bit) - 1u)) - win0;
            }
            return (slot >= 0 && slot < nslot) ? slot : -1;
        };
        // (b) ONE returning LDS atomic per pair.  A pair that finds others before it puts itself on the queue of passages
        // with several surviving centroids; the second pair also enqueues the first, whose list id it reads off the
        // count, and starts the slot's row of column maxima (nothing folds into rows before the barrier below)
        auto count_pid = [&](int pid, int j) {
            const int slot = slot_of(pid);
            if (slot >= 0) {
                const int old = atomicAdd(&acc[slot * S1S_STRIDE + 32], (1 << S1S_IDBITS) | j);
                const int before = old >> S1S_IDBITS;
                if (before >= 1) {
                    const int n = before == 1 ? 2 : 1;
                    if (before == 1) {
#pragma unroll
                        for (int q = 0; q < 32; q++) acc[slot * S1S_STRIDE + q] = init;
                    }
                    const int at = atomicAdd(&s_qn, n);
                    if (at + n <= S1S_QCAP) {
                        queue[at] = (slot << S1S_IDBITS) | j;
                        qpid[at] = (uint16_t)pid;
                        if (before == 1) {
                            queue[at + 1] = (slot << S1S_IDBITS) | (old & ((1 << S1S_IDBITS) - 1));
                            qpid[at + 1] = (uint16_t)pid;
                        }
                    }
                }
            }
        };
        auto count_group = [&](const grp& g, const begs& bg, int j0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (g.sv[u] >= g.ev[u]) continue;  // wave-uniform
                const int j = wave + S1S_WAVES * (j0 + u);
                count_pid(pid_of(g, u, pid0), j);
                for (uint32_t x0 = g.sv[u] + 64; x0 < g.ev[u]; x0 += 64) {
                    count_pid(x0 + lane < g.ev[u] ? bg.p[u][x0 + lane] - pid0 : -1, j);
                    S1S_DRAIN();
                }
            }
        };
#ifdef S1S_NO_HINT   // development A/B (the hint costs the planted-centroid path 0.03 ms per step in scalar spills)
        if (true) {
#else
        if (!dense_hint) {   // (block-uniform)
#endif
            if (mq.n > 0) count_group(gq, bq0, 0);   // the entries kept from the marking pass
            for (int j0 = 4; j0 < mq.n; j0 += 4) {
                grp gt;
                const begs bg = list_begs(mq, j0);
                issue(mq, bg, mq.s, mq.e, j0, gt);
                count_group(gt, bg, j0);
                S1S_DRAIN();
            }
        }
        if (tid == 0 && win0 == 0) s_base = my_base;   // the chunk's key base: the atomic issued before the counting pass
        s1s_sync();
        S1S_STAMP(6);
#ifdef S1S_NO_HINT
        const int qn = s_qn;
#else
        const int qn = dense_hint ? S1S_QCAP + 1 : s_qn;
#endif
        const bool dense = qn > S1S_QCAP;   // block-uniform
        if (dense) {
            // More multi-centroid pairs than the queue holds: every pair of the window walks the 32 columns itself --
            // 32 x (v_readlane, ds_max) per 64 pairs into fully initialised rows; lanes without a slot aim at the scratch
            // words behind the accumulators (one word per lane and column, so they do not collide).
            for (int e = tid; e < nslot * S1S_STRIDE; e += 64 * S1S_WAVES) acc[e] = init;
            s1s_sync();
            auto scatter_pids = [&](int pid, int rowk) {
                const int slot = slot_of(pid);
                int* dst = slot >= 0 ? acc + slot * S1S_STRIDE : acc + S1S_SLOTS * S1S_STRIDE + lane;
#pragma unroll
                for (int q = 0; q < 32; q++) atomicMax(dst + q, __builtin_amdgcn_readlane(rowk, q));
            };
            for (int j0 = 0; j0 < mq.n; j0 += 4) {
                grp gt;
                int rowv[4];
                const begs bg = list_begs(mq, j0);
                issue(mq, bg, mq.s, mq.e, j0, gt);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    rowv[u] = s1s_enc(cs_b[(size_t)(a.cs_compact ? wave + S1S_WAVES * ((j0 + u < mq.n) ? j0 + u : j0)
                                                                     : __builtin_amdgcn_readlane(mq.c, (j0 + u < mq.n) ? j0 + u : j0)) * 32 + k]);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (gt.sv[u] >= gt.ev[u]) continue;  // wave-uniform
                    scatter_pids(pid_of(gt, u, pid0), rowv[u]);
                    for (uint32_t x0 = gt.sv[u] + 64; x0 < gt.ev[u]; x0 += 64)
                        scatter_pids(x0 + lane < gt.ev[u] ? bg.p[u][x0 + lane] - pid0 : -1, rowv[u]);
                }
                S1S_DRAIN();
            }
            s1s_sync();
        } else if (qn > 0) {
            // (c) the queued pairs: half a wave per pair, one column per lane -- fold the pair's score row into the slot's
            // row (ds_max on the order-preserving int encoding, exact)
            for (int e = tid >> 5; e < qn; e += 2 * S1S_WAVES) {
                const int ent = queue[e];
                const int lj = ent & ((1 << S1S_IDBITS) - 1);
                const int c = a.cs_compact ? lj : a.qual[(size_t)b * a.qmax + lj];
                atomicMax(&acc[(ent >> S1S_IDBITS) * S1S_STRIDE + k], s1s_enc(cs_b[(size_t)c * 32 + k]));
            }
            S1S_DRAIN();
            s1s_sync();
        }
        // ascending-k sum of a slot's column maxima (filter_pids.cpp:59-63)
        auto column_sum = [&](int sl) {
            float sc = 0.0f;
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 8) {   // eight LDS reads in flight at a time (registers are scarce here)
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = s1s_dec(acc[sl * S1S_STRIDE + q0 + q]);
#pragma unroll
                for (int q = 0; q < 8; q++) sc += q0 + q < nqc ? (F16 ? flmr_round_f16(v[q]) : v[q]) : 0.0f;
            }
            return F16 ? flmr_round_f16(sc) : sc;
        };
        if (dense) {   // every slot holds a row: scores into the padding words, one thread per slot
            for (int sl = tid; sl < nslot; sl += 64 * S1S_WAVES) acc[sl * S1S_STRIDE + 32] = __float_as_int(column_sum(sl));
            s1s_sync();
        }
        S1S_STAMP(7);
        // (d) keys at the chunk's base + rank among the hits.  One thread per bitmap word takes its hit candidates of this
        // window: a passage with one surviving centroid scores its list's constant (two LDS reads); the multi-centroid ones
        // are skipped here and written from the queue below, which knows slot and passage of each (a slot queued several
        // times is written several times with the same key)
        {
            const int64_t kbase = s_base;
            uint32_t bits = hcw;
            while (bits) {
                const int bit = __ffs(bits) - 1;
                bits &= bits - 1;
                const int rank = hpos + __popc(hcw & ((1u << bit) - 1u));
                const int slot = rank - win0;
                if (slot >= 0 && slot < nslot) {
                    const int info = acc[slot * S1S_STRIDE + 32];
                    acc[slot * S1S_STRIDE + 32] = 0;   // the counter's last reader leaves it ready for the next window / chunk
                    if (kbase + rank < a.cand_cap) {
                        if (dense) keys_b[kbase + rank] = flmr_make_key(__int_as_float(info), pid0 + tid * 32 + bit);
                        else if ((info >> S1S_IDBITS) == 1) keys_b[kbase + rank] = flmr_make_key(rsum[info & ((1 << S1S_IDBITS) - 1)], pid0 + tid * 32 + bit);
                    }
                }
            }
            if (tid == 0 && qn > 0) s_qn = 0;   // (qn > 0: a barrier lies between every thread's read of it and this pass)
            if (win0 + S1S_SLOTS >= nh) { cb[tid] = 0u; hb[tid] = 0u; }   // last window: nobody reads the bitmaps any more
            if (!dense) {
                for (int e = tid; e < qn; e += 64 * S1S_WAVES) {
                    const int slot = queue[e] >> S1S_IDBITS;
                    const int64_t pos = kbase + win0 + slot;
                    if (pos < a.cand_cap) keys_b[pos] = flmr_make_key(column_sum(slot), pid0 + (int)qpid[e]);
                }
            }
        }
        s1s_sync();
        S1S_STAMP(8);
        dense_hint = dense;
    }
    }
    // next chunk
    mc.s = mc.e; mc.e = mc_e2;
    mq.s = mq.e; mq.e = mq_e2;
    gc = gcn; gq = gqn;
    if (!(scatter && nh > 0)) {  // (the window loop ends with the bitmaps cleared and a barrier)
        cb[tid] = 0u; hb[tid] = 0u;
        s1s_sync();
    }
    S1S_STAMP(9);
    }
#ifdef S1S_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 10; i++) atomicAdd(&s1s_prof[i], (unsigned long long)pt[i]);
        atomicAdd(&s1s_prof[10], (unsigned long long)(ch_end - ch0));
        atomicAdd(&s1s_prof[11], 1ull);
    }
#endif
}

// ---- kernel A'': the QUEUE form of the scatter kernel -----------------------------------------------------------------------
// Same outputs as cand_mark_score_kernel (bitmaps, per-chunk counts, one stage-1 key per candidate of the hit set) for the corpus
// shape the index is built for -- nearly every hit passage holds ONE surviving centroid -- with a third of the instructions:
// the slot kernel is VALU-issue bound (~1250 wave instructions per wave and chunk in seven barrier-separated phases).
//   mark     the probed cells' slices set the candidate bitmap; a surviving list's pair sets its hit bit with a RETURNING
//            ds_or: the pair that finds the bit set marks the passage in a third bitmap ("several surviving centroids").
//   barrier  (the only one of a chunk)
//   words    bitmaps out, candidate / hit counts by one LDS atomic per wave; the wave that arrives last owns the chunk's
//            totals: it writes the counts and reserves the chunk's key range with the global atomic, whose result is first
//            read a whole chunk later.  The next chunk's slices are requested here.
//   pairs    every (list, passage) pair again, from the registers it was loaded into: candidate and single -> the list's
//            constant (see the slot kernel) as a key into an LDS buffer; candidate and several -> (passage, list) on a queue.
//   deferred (one chunk later, after that chunk's barrier, when the key range is known) the staged keys go out in one coalesced
//            pass; the queue is folded by half-waves: the entry with the lowest index among those of its passage gathers the
//            others' score rows (column maxima on the order-preserving int encoding, ascending-k sum: the slot kernel's
//            arithmetic, bit for bit) and writes the key at the END of the chunk's range.
// Bitmaps, queue and counters rotate over three sets and the key buffer over two, so that a wave may run ahead into the next
// chunk's marking while others still read this chunk's: each set is cleared / reset in the period after its last reader.
// Eight waves and 77 KB of LDS: two workgroups per CU hide each other's barrier and LDS round trips.
// What it does not handle it hands over whole, per query (fast_state[FLMR_FAST_HDR + b] != 0 -> the later kernels redo the query): more
// staged keys or queued pairs than fit (a corpus whose clusters overlap: the slot kernel's dense form), more than 512 lists.
// Two cumulative counters (queries handed over after trying / tried) switch a searcher whose queries mostly overflow to trying
// one query in 64 of one batch in 16.
#define CF_WAVES 8
#define CF_THREADS (64 * CF_WAVES)
#define CF_FC 16          // probed-cell lists per wave whose slices are requested a chunk ahead (ncells = 4: 128 cells)
#define CF_FQ 8           // surviving lists per wave requested a chunk ahead and kept in registers for the pair pass
#define CF_KCAP 3072      // keys of a chunk staged in LDS (as list << 15 | passage inside the chunk: 4 bytes each)
#define CF_QCAP 256       // queued pairs per chunk
#define CF_MAXLISTS (64 * CF_WAVES)
static_assert(CF_KCAP == CF_KCAP_LIMIT && CF_QCAP == CF_QCAP_LIMIT && CF_MAXLISTS == 512, "cand_plan_kernel plans against these limits");
#define CF_RC 64          // surviving lists whose score rows are kept in LDS for the queued pairs (the others are read from memory)

struct cf_chunk_state { int qn, ks, km, nh, base, pad0, pad1, pad2; };
struct cf_slices { int c; int64_t beg; uint32_t s, e; int n; };   // one list per lane: list wave + CF_WAVES * lane
__device__ __forceinline__ cf_slices cf_load_slices(const int32_t* ids, int n_total, int wave, int lane, const int64_t* ivf_offsets,
                                                    const uint32_t* tab, int nchunks, int ch) {
    cf_slices m;
    m.n = wave < n_total ? (n_total - wave + CF_WAVES - 1) / CF_WAVES : 0;
    m.c = 0; m.beg = 0; m.s = 0; m.e = 0;
    if (lane < m.n) {
        m.c = ids[wave + CF_WAVES * lane];
        m.beg = ivf_offsets[m.c];
        m.s = tab[(size_t)m.c * (nchunks + 1) + ch];
        m.e = tab[(size_t)m.c * (nchunks + 1) + ch + 1];
    }
    return m;
}
// sum over the wave: row sums by DPP (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), the four rows on the scalar unit
__device__ __forceinline__ int cf_wave_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}

#ifdef CF_PROFILE   // development only: per-phase clocks of wave 0, summed over the grid and printed by the launcher
__device__ unsigned long long cf_prof[12];
#define CF_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define CF_STAMP(k) do { } while (0)
#endif

template <bool F16>
__global__ __launch_bounds__(CF_THREADS, 2) void cand_fast_kernel(flmr_cand_args a, int cpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem);                          // [3][candidate | hit | several][CAND_CHUNK_WORDS]
    uint32_t* kbuf = bm + 9 * CAND_CHUNK_WORDS;                                // [2][CF_KCAP] list << 15 | passage inside the chunk
    uint32_t* queue = kbuf + 2 * CF_KCAP;                                      // [3][CF_QCAP] passage (inside the chunk) | list << 16
    int* rows = reinterpret_cast<int*>(queue + 3 * CF_QCAP);                   // [CF_RC][32] score rows of the query's first surviving lists (order-encoded, floored)
    uint32_t* rtab = reinterpret_cast<uint32_t*>(rows + CF_RC * 32);           // [CF_MAXLISTS] score half of the key of a passage whose only surviving centroid is list j
    __shared__ cf_chunk_state st[3];
    __shared__ int s_tot, s_arr, s_abort[2];   // s_abort[round & 1]: set during a round, read after the NEXT round's barrier
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, k = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t* const redo = a.fast_state + FLMR_FAST_HDR + b;
    int32_t* const kcount = a.fast_state + FLMR_FAST_HDR + a.nqueries + b;
    const int nl = a.ncell[b], nq = a.nqual[b];
    {
        // (the plan word, cand_plan_kernel: 0 = this form; anything else is another form's query)
        const bool ok = *redo == 0 && a.hit_valid[b] != 0 && nl <= CF_MAXLISTS && nq <= CF_MAXLISTS;
        if (!ok) return;   // (block-uniform; the word is 0 or 2 -- given up by another workgroup of this launch: its chunks are redone anyway)
    }
    const int ch0 = blockIdx.y * cpb;
    const int ch_end = ch0 + cpb < a.nchunks ? ch0 + cpb : a.nchunks;
    cf_slices mc = cf_load_slices(a.cells + (size_t)b * a.max_cells, nl, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
    cf_slices mq = cf_load_slices(a.qual + (size_t)b * a.qmax, nq, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
    const int init = s1s_enc(-9999.0f);
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq_cand;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;  // <= 32 on this path
    const float* cs_b = a.cs + (size_t)b * a.cs_query_stride;
    uint64_t* keys_b = a.keys + (size_t)b * a.cand_cap;

    auto list_ptr = [&](const cf_slices& m, int u) { return a.ivf_pids + s1s_bcast64(m.beg, u); };
    // the first 64 entries of list u's slice [ls, le) (bounds of list u in lane u); every lane loads (lanes past the end repeat the
    // last entry: a load inside a divergent branch would be waited for before the branch ends)
    auto issue1 = [&](const cf_slices& m, int n, int u, uint32_t ls, uint32_t le) {
        int r = 0;
        if (u < n) {   // wave-uniform
            const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)ls, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)le, u);
            if (sv < ev) {
                const uint32_t x = sv + lane;
                r = list_ptr(m, u)[x < ev ? x : ev - 1u];
            }
        }
        return r;
    };
    int raw_c[CF_FC], raw_q[CF_FQ], raw_qn[CF_FQ];
#pragma unroll
    for (int u = 0; u < CF_FC; u++) raw_c[u] = issue1(mc, mc.n, u, mc.s, mc.e);
#pragma unroll
    for (int u = 0; u < CF_FQ; u++) { raw_q[u] = issue1(mq, mq.n, u, mq.s, mq.e); raw_qn[u] = 0; }
    // the stage-1 score of a passage whose only surviving centroid is this lane's list (see the slot kernel)
    float rconst = 0.0f;
    if (lane < mq.n) {
        const float4* r4 = reinterpret_cast<const float4*>(cs_b + (size_t)(a.cs_compact ? wave + CF_WAVES * lane : mq.c) * 32);
        float sc = 0.0f;
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 v = r4[q4];
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = s1s_enc(x[i]);
                const float m = s1s_dec(e > init ? e : init);
                sc += q4 * 4 + i < nqc ? (F16 ? flmr_round_f16(m) : m) : 0.0f;
            }
        }
        rconst = F16 ? flmr_round_f16(sc) : sc;
    }
    if (lane < mq.n) rtab[wave + CF_WAVES * lane] = flmr_f2ord(rconst);   // (read after the first barrier)
    for (int j = tid >> 5; j < nq && j < CF_RC; j += 2 * CF_WAVES) {
        const int c = a.cs_compact ? j : a.qual[(size_t)b * a.qmax + j];
        const int v = s1s_enc(cs_b[(size_t)c * 32 + k]);
        rows[j * 32 + k] = v > init ? v : init;
    }
    for (int e = tid; e < 9 * CAND_CHUNK_WORDS; e += CF_THREADS) bm[e] = 0u;
    if (tid < 3) { st[tid].qn = 0; st[tid].ks = 0; st[tid].km = 0; st[tid].nh = 0; st[tid].base = 0; }
    if (tid == 0) { s_tot = 0; s_arr = 0; s_abort[0] = 0; s_abort[1] = 0; }
    s1s_sync();

#ifdef CF_PROFILE
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
    int my_base = 0;
    bool issuer = false;          // this wave arrived last in the previous chunk's count and holds its key base (wave-uniform)
    int r3 = 0;                   // the chunk's set of bitmaps / queue / counters
    for (int ch = ch0; ch <= ch_end; ch++) {   // (the last round only drains the last chunk's deferred work)
        const bool live = ch < ch_end, prev = ch > ch0;
        const int rp = r3 == 0 ? 2 : r3 - 1, rn = r3 == 2 ? 0 : r3 + 1;
        const int pid0 = ch * CAND_CHUNK_PIDS;
        uint32_t* const cb = bm + (r3 * 3 + 0) * CAND_CHUNK_WORDS;
        uint32_t* const hb = bm + (r3 * 3 + 1) * CAND_CHUNK_WORDS;
        uint32_t* const mb = bm + (r3 * 3 + 2) * CAND_CHUNK_WORDS;
        uint32_t* const kb = kbuf + (ch & 1) * CF_KCAP;
        uint32_t* const qu = queue + r3 * CF_QCAP;
        uint32_t mc_e2 = 0, mq_e2 = 0;
        if (ch + 1 < ch_end) {   // the end of the NEXT chunk's slices
            if (lane < mc.n) mc_e2 = a.chunk_tab[(size_t)mc.c * (a.nchunks + 1) + ch + 2];
            if (lane < mq.n) mq_e2 = a.chunk_tab[(size_t)mq.c * (a.nchunks + 1) + ch + 2];
        }
        CF_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this chunk's slices (requested a chunk ago) and the previous chunk's key base
        CF_STAMP(1);
        if (issuer) {
            if (lane == 0) st[rp].base = my_base;
            issuer = false;
        }
        if (live) {
            // ---- mark ----
            // (the list counts are made opaque once per phase: otherwise the 24 "u < n" tests of the unrolled loops are computed
            // once before the chunk loop and kept in spilled scalar registers, two v_readlane per use)
            int ncl = mc.n, nql = mq.n;
            asm volatile("" : "+s"(ncl), "+s"(nql));
#pragma unroll
            for (int u = 0; u < CF_FC; u++) {
                if (u < ncl) {
                    const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mc.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mc.e, u);
                    if (sv + lane < ev) {
                        const int p = raw_c[u] - pid0;
                        atomicOr(&cb[p >> 5], 1u << (p & 31));
                    }
                    if (ev - sv > 64u && sv < ev) {   // rare: the slice's entries beyond the first 64
                        const int32_t* ptr = list_ptr(mc, u);
                        for (uint32_t x = sv + 64 + lane; x < ev; x += 64) {
                            const int p = ptr[x] - pid0;
                            atomicOr(&cb[p >> 5], 1u << (p & 31));
                        }
                        S1S_DRAIN();
                    }
                }
            }
            for (int u = CF_FC; u < mc.n; u++) {   // rare: more than CF_FC lists per wave
                const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mc.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mc.e, u);
                const int32_t* ptr = list_ptr(mc, u);
                for (uint32_t x = sv + lane; x < ev; x += 64) {
                    const int p = ptr[x] - pid0;
                    atomicOr(&cb[p >> 5], 1u << (p & 31));
                }
                S1S_DRAIN();
            }
            auto mark_hit = [&](int p) {
                const uint32_t bit = 1u << (p & 31);
                const uint32_t old = atomicOr(&hb[p >> 5], bit);
                if (old & bit) atomicOr(&mb[p >> 5], bit);
            };
#pragma unroll
            for (int u = 0; u < CF_FQ; u++) {
                if (u < nql) {
                    const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                    if (sv + lane < ev) mark_hit(raw_q[u] - pid0);
                    if (ev - sv > 64u && sv < ev) {
                        const int32_t* ptr = list_ptr(mq, u);
                        for (uint32_t x = sv + 64 + lane; x < ev; x += 64) mark_hit(ptr[x] - pid0);
                        S1S_DRAIN();
                    }
                }
            }
            for (int u = CF_FQ; u < mq.n; u++) {
                const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                const int32_t* ptr = list_ptr(mq, u);
                for (uint32_t x = sv + lane; x < ev; x += 64) mark_hit(ptr[x] - pid0);
                S1S_DRAIN();
            }
        }
        CF_STAMP(2);
        s1s_sync();
        CF_STAMP(3);
        if (s_abort[(ch + 1) & 1]) return;   // (block-uniform: written only before the barrier, never cleared)
        if (live) {
            // ---- words: bitmaps out, counts, clear the previous chunk's set ----
            int mine = 0;
#pragma unroll
            for (int h = 0; h < CAND_CHUNK_WORDS / CF_THREADS; h++) {
                const int w = tid + h * CF_THREADS;
                const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + w;
                uint32_t cw = cb[w];
                const uint32_t hw = hb[w];
                if (gw < a.words) {
                    a.cand_bits[(size_t)b * a.words + gw] = cw;
                    a.hit_bits[(size_t)b * a.words + gw] = hw;
                } else {
                    cw = 0u;
                }
                mine += __popc(cw) | (__popc(cw & hw) << 16);
                bm[(rp * 3 + 0) * CAND_CHUNK_WORDS + w] = 0u;
                bm[(rp * 3 + 1) * CAND_CHUNK_WORDS + w] = 0u;
                bm[(rp * 3 + 2) * CAND_CHUNK_WORDS + w] = 0u;
            }
            const int wsum = cf_wave_sum(mine);   // candidates | hit candidates << 16 (each <= 32768)
            int last = 0;
            if (lane == 0) {
                atomicAdd(&s_tot, wsum);
                if (atomicAdd(&s_arr, 1) == CF_WAVES - 1) {   // every wave's total is in
                    const uint32_t tot = (uint32_t)atomicExch(&s_tot, 0);
                    s_arr = 0;
                    const int cnt = (int)(tot & 0xffffu), nh = (int)(tot >> 16);
                    a.chunk_cnt[(size_t)b * a.nchunks + ch] = cnt;
                    a.chunk_hits[(size_t)b * a.nchunks + ch] = nh;
                    st[r3].nh = nh;
                    if (nh > CF_KCAP) { s_abort[ch & 1] = 1; *redo = 2; }
                    int zero = 0;
                    asm volatile("" : "+v"(zero));   // (see the slot kernel: keeps the compiler's atomic optimiser off this atomic)
                    my_base = atomicAdd(kcount + zero, nh);
                    last = 1;
                }
            }
            issuer = __builtin_amdgcn_readfirstlane(last) != 0;
            CF_STAMP(4);
            // ---- the next chunk's slices ----
            if (ch + 1 < ch_end) {
                int ncp = mc.n, nqp = mq.n;
                asm volatile("" : "+s"(ncp), "+s"(nqp));
#pragma unroll
                for (int u = 0; u < CF_FC; u++) raw_c[u] = issue1(mc, ncp, u, mc.e, mc_e2);
#pragma unroll
                for (int u = 0; u < CF_FQ; u++) raw_qn[u] = issue1(mq, nqp, u, mq.e, mq_e2);
            }
            CF_STAMP(5);
            // ---- pairs ----
            auto emit = [&](int p, int u) {
                bool single = false, several = false;
                if (p >= 0) {
                    const uint32_t bit = 1u << (p & 31);
                    const bool cand = (cb[p >> 5] & bit) != 0u, sev = (mb[p >> 5] & bit) != 0u;
                    single = cand && !sev;
                    several = cand && sev;
                }
                const uint64_t ms = __builtin_amdgcn_ballot_w64(single);
                if (ms) {   // wave-uniform
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&st[r3].ks, __popcll(ms));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
                    if (single && pos < CF_KCAP)
                        kb[pos] = ((uint32_t)(wave + CF_WAVES * u) << 15) | (uint32_t)p;
                }
                if (__builtin_amdgcn_ballot_w64(several)) {
                    if (several) {
                        const int at = atomicAdd(&st[r3].qn, 1);
                        if (at < CF_QCAP) qu[at] = (uint32_t)p | ((uint32_t)(wave + CF_WAVES * u) << 16);
                        else { s_abort[ch & 1] = 1; *redo = 2; }
                    }
                }
            };
            {   // the first groups, from the registers they were loaded into: every list's LDS reads first, ONE position atomic per wave
                int nqe = mq.n;
                asm volatile("" : "+s"(nqe));
                int pp[CF_FQ];
                uint32_t cw_[CF_FQ], mw_[CF_FQ];
#pragma unroll
                for (int u = 0; u < CF_FQ; u++) {
                    pp[u] = -1; cw_[u] = 0u; mw_[u] = 0u;
                    if (u < nqe) {
                        const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                        const int p = raw_q[u] - pid0;
                        const bool ok = sv + lane < ev;
                        pp[u] = ok ? p : -1;
                        const int w = ok ? p >> 5 : 0;
                        cw_[u] = cb[w]; mw_[u] = mb[w];
                    }
                }
                uint64_t ms[CF_FQ];
                int total = 0;
                bool anysev = false;
#pragma unroll
                for (int u = 0; u < CF_FQ; u++) {
                    ms[u] = 0ull;
                    if (u < nqe) {
                        const uint32_t bit = 1u << (pp[u] & 31);
                        const bool cand = pp[u] >= 0 && (cw_[u] & bit) != 0u, sev = (mw_[u] & bit) != 0u;
                        ms[u] = __builtin_amdgcn_ballot_w64(cand && !sev);
                        total += __popcll(ms[u]);
                        anysev |= cand && sev;
                    }
                }
                if (total) {   // wave-uniform
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&st[r3].ks, total);
                    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                    for (int u = 0; u < CF_FQ; u++) {
                        if (u < nqe) {
                            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ms[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms[u], 0u));
                            const bool single = (ms[u] >> lane) & 1ull;
                            if (single && pos < CF_KCAP)
                                kb[pos] = ((uint32_t)(wave + CF_WAVES * u) << 15) | (uint32_t)pp[u];
                            base += __popcll(ms[u]);
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(anysev)) {   // wave-uniform; rare per list
#pragma unroll
                    for (int u = 0; u < CF_FQ; u++) {
                        const bool sev = u < nqe && pp[u] >= 0 && (cw_[u] & mw_[u] & (1u << (pp[u] & 31))) != 0u;
                        if (sev) {
                            const int at = atomicAdd(&st[r3].qn, 1);
                            if (at < CF_QCAP) qu[at] = (uint32_t)pp[u] | ((uint32_t)(wave + CF_WAVES * u) << 16);
                            else { s_abort[ch & 1] = 1; *redo = 2; }
                        }
                    }
                }
            }
            for (int u = 0; u < mq.n; u++) {   // rare: slices longer than 64 entries, lists beyond the first CF_FQ of a wave
                const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                const uint32_t from = u < CF_FQ ? sv + 64u : sv;
                if (from < ev) {
                    const int32_t* ptr = list_ptr(mq, u);
                    for (uint32_t x0 = from; x0 < ev; x0 += 64) {
                        emit(x0 + lane < ev ? ptr[x0 + lane] - pid0 : -1, u);
                        S1S_DRAIN();
                    }
                }
            }
        }
        CF_STAMP(6);
        if (prev) {
            // ---- deferred: the previous chunk's keys ----
            const cf_chunk_state& S = st[rp];
            const int base = S.base, nh = S.nh, qn = S.qn;
            const int ns = S.ks < CF_KCAP ? S.ks : CF_KCAP;
            const uint32_t* kp = kbuf + ((ch - 1) & 1) * CF_KCAP;
            const uint32_t pid0q = (uint32_t)(ch - 1) * CAND_CHUNK_PIDS;
            for (int t = tid; t < ns; t += CF_THREADS) {
                const uint32_t e = kp[t];
                if ((int64_t)base + t < a.cand_cap) keys_b[(int64_t)base + t] = ((uint64_t)rtab[e >> 15] << 32) | (pid0q + (e & 0x7fffu));
            }
            if (qn > 0) {   // block-uniform
                const uint32_t* qp = queue + rp * CF_QCAP;
                const int pid0p = (ch - 1) * CAND_CHUNK_PIDS;
                const int hw = tid >> 5, hi = lane >> 5;
                for (int e0 = 0; e0 < qn; e0 += 2 * CF_WAVES) {
                    const int e = e0 + hw;
                    const bool act = e < qn;
                    const uint32_t pid = act ? (qp[e] & 0xffffu) : 0xffffffffu;
                    // leader = the lowest entry of the passage; it folds the rows of all of them
                    bool leader = act;
                    int menc = init;
                    for (int i0 = 0; i0 < qn; i0 += 32) {   // (block-uniform bounds)
                        const int i = i0 + k;
                        const uint32_t ent = i < qn ? qp[i] : 0xffffffffu;
                        const bool same = (ent & 0xffffu) == pid && i < qn;
                        const uint64_t bal = __builtin_amdgcn_ballot_w64(same);
                        uint32_t hm = hi ? (uint32_t)(bal >> 32) : (uint32_t)bal;   // this half-wave's members in [i0, i0 + 32)
                        if (hm && i0 + (int)__ffs(hm) - 1 < e) leader = false;
                        if (!leader) hm = 0u;
                        while (hm) {
                            const int m = i0 + (int)__ffs(hm) - 1;
                            hm &= hm - 1u;
                            const int lj = (int)(qp[m] >> 16);
                            int v;
                            if (lj < CF_RC) {
                                v = rows[lj * 32 + k];
                            } else {
                                const int c = a.cs_compact ? lj : a.qual[(size_t)b * a.qmax + lj];
                                v = s1s_enc(cs_b[(size_t)c * 32 + k]);
                            }
                            menc = v > menc ? v : menc;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(leader)) {   // wave-uniform
                        // ascending-k sum (filter_pids.cpp:59-63) across the half-wave's 32 lanes: step t makes lane t's prefix from
                        // lane t-1's (v_add_f32 with a wave_shr:1 operand); later steps overwrite it, only lane 31's total is kept
                        const float mv = s1s_dec(menc);
                        const float x = k < nqc ? (F16 ? flmr_round_f16(mv) : mv) : 0.0f;
                        float acc = k == 0 ? 0.0f + x : x;
#pragma unroll
                        for (int t = 1; t < 32; t++)
                            acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x138, 0xF, 0xF, false)) + x;
                        if (leader && k == 31) {
                            const float sc = F16 ? flmr_round_f16(acc) : acc;
                            const int64_t pos = (int64_t)base + nh - 1 - atomicAdd(&st[rp].km, 1);
                            if (pos >= base && pos < a.cand_cap) keys_b[pos] = flmr_make_key(sc, pid0p + (int)pid);
                        }
                    }
                }
            }
        }
        CF_STAMP(7);
        // the counters of the set the NEXT chunk uses: their last readers (the deferred pass of the previous round) are behind this
        // round's barrier
        if (tid == 0) { st[rn].qn = 0; st[rn].ks = 0; st[rn].km = 0; }
        // next chunk
        mc.s = mc.e; mc.e = mc_e2;
        mq.s = mq.e; mq.e = mq_e2;
#pragma unroll
        for (int u = 0; u < CF_FQ; u++) raw_q[u] = raw_qn[u];
        r3 = rn;
    }
#ifdef CF_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&cf_prof[i], (unsigned long long)pt[i]);
        atomicAdd(&cf_prof[10], (unsigned long long)(ch_end - ch0));
        atomicAdd(&cf_prof[11], 1ull);
    }
#endif
}

// ---- kernel A-3: the SMALL-DENSE form of the scatter kernel ------------------------------------------------------------------
// The regime an index built from real embeddings is expected to be in (profiles/built_index_probe.py: ~95 surviving centroids
// per query, ~19 of them in every hit passage, ~70 hit candidates and ~1.3 k (list, passage) pairs per chunk): every hit passage
// folds several score rows, so the queue form gives the query up at once, and the slot form spends ~19 us per chunk in seven
// barrier-separated phases on very little work.  This form is the queue form's frame -- eight waves, two workgroups per CU, the
// slices of the next chunk requested a chunk ahead, counts by the wave that arrives last -- with the slot form's dense fold:
//   mark / barrier / words: bitmaps out, counts, and a SLOT for every hit candidate (a wave scan of the words' popcounts + one
//   LDS atomic per wave: slots are unique, not ordered) / barrier / pairs: every (list, passage) pair folds its list's row
//   into the slot's 32 column maxima (ds_max on the order-preserving encoding) / barrier / sums: half a wave per slot, the
//   ascending-k sum as a wave_shr DPP chain, the key straight to base + slot.
// Three barriers a chunk.  It runs after the queue form, for the queries that one left untried because the searcher's counters
// say its queries mostly overflow the queue (fast_state: 1 -> 3 "done here" / 4 "given up": more than CF_DSLOTS hit candidates
// in a chunk); the slot kernel, launched last, takes what is left.
#define CF_DSLOTS 256
static_assert(CF_DSLOTS == CF_DSLOTS_LIMIT, "cand_plan_kernel plans against this limit");
#define CF_DRC 128         // surviving lists whose score rows are kept in LDS
#define CF_DFQ 16          // surviving lists per wave kept in registers from the marking to the pair pass

template <bool F16>
__global__ __launch_bounds__(CF_THREADS, 2) void cand_dense_small_kernel(flmr_cand_args a, int cpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem);                          // [2][candidate | hit][CAND_CHUNK_WORDS]
    int* acc = reinterpret_cast<int*>(bm + 4 * CAND_CHUNK_WORDS);              // [CF_DSLOTS][33] column maxima (+ 64 scratch words)
    int* rows = acc + CF_DSLOTS * S1S_STRIDE + 64;                             // [CF_DRC][32]
    uint16_t* wslot = reinterpret_cast<uint16_t*>(rows + CF_DRC * 32);         // [CAND_CHUNK_WORDS] slot of a word's first hit candidate
    uint16_t* spid = wslot + CAND_CHUNK_WORDS;                                 // [CF_DSLOTS] the slot's passage (inside the chunk)
    __shared__ int s_tot, s_arr, s_slots, s_nh, s_base, s_abort;
    __shared__ __attribute__((aligned(8))) int vs_all[CF_WAVES][66];           // a list's slots, packed, per wave (see the pair pass)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, k = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t* const redo = a.fast_state + FLMR_FAST_HDR + b;
    int32_t* const kcount = a.fast_state + FLMR_FAST_HDR + a.nqueries + b;
    const int nl = a.ncell[b], nq = a.nqual[b];
    {
        const int r = *redo;   // FLMR_PLAN_SMALL: planned for this form (cand_plan_kernel); 3: another workgroup of this launch was here first
        const bool ok = (r == FLMR_PLAN_SMALL || r == 3) && a.hit_valid[b] != 0 && nl <= CF_MAXLISTS && nq <= CF_MAXLISTS;
        if (!ok) return;   // (block-uniform)
        if (tid == 0) atomicCAS(redo, FLMR_PLAN_SMALL, 3);
    }
    const int ch0 = blockIdx.y * cpb;
    const int ch_end = ch0 + cpb < a.nchunks ? ch0 + cpb : a.nchunks;
    cf_slices mc = cf_load_slices(a.cells + (size_t)b * a.max_cells, nl, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
    cf_slices mq = cf_load_slices(a.qual + (size_t)b * a.qmax, nq, wave, lane, a.ivf_offsets, a.chunk_tab, a.nchunks, ch0);
    const int init = s1s_enc(-9999.0f);
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq_cand;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;  // <= 32 on this path
    const float* cs_b = a.cs + (size_t)b * a.cs_query_stride;
    uint64_t* keys_b = a.keys + (size_t)b * a.cand_cap;
    auto list_ptr = [&](const cf_slices& m, int u) { return a.ivf_pids + s1s_bcast64(m.beg, u); };
    auto issue1 = [&](const cf_slices& m, int n, int u, uint32_t ls, uint32_t le) {
        int r = 0;
        if (u < n) {   // wave-uniform
            const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)ls, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)le, u);
            if (sv < ev) {
                const uint32_t x = sv + lane;
                r = list_ptr(m, u)[x < ev ? x : ev - 1u];
            }
        }
        return r;
    };
    int* const vs = vs_all[wave];
    auto row_of = [&](int j) {   // list j's score row, column k, order-encoded and floored
        if (j < CF_DRC) return rows[j * 32 + k];
        const int c = a.cs_compact ? j : a.qual[(size_t)b * a.qmax + j];
        const int v = s1s_enc(cs_b[(size_t)c * 32 + k]);
        return v > init ? v : init;
    };
    // (CF_DFQ surviving lists per wave stay in registers from the marking to the pair pass: ~100 surviving centroids are 12 per
    // wave here, and a list that is reloaded costs a memory round trip in both passes.  Their next slices are requested behind
    // the pair pass, the probed cells' -- dead after the marking -- a phase earlier.)
    int raw_c[CF_FC], raw_q[CF_DFQ];
#pragma unroll
    for (int u = 0; u < CF_FC; u++) raw_c[u] = issue1(mc, mc.n, u, mc.s, mc.e);
#pragma unroll
    for (int u = 0; u < CF_DFQ; u++) raw_q[u] = issue1(mq, mq.n, u, mq.s, mq.e);
    for (int j = tid >> 5; j < nq && j < CF_DRC; j += 2 * CF_WAVES) {
        const int c = a.cs_compact ? j : a.qual[(size_t)b * a.qmax + j];
        const int v = s1s_enc(cs_b[(size_t)c * 32 + k]);
        rows[j * 32 + k] = v > init ? v : init;
    }
    for (int e = tid; e < 4 * CAND_CHUNK_WORDS; e += CF_THREADS) bm[e] = 0u;
    for (int e = tid; e < CF_DSLOTS * S1S_STRIDE + 64; e += CF_THREADS) acc[e] = init;
    if (tid == 0) { s_tot = 0; s_arr = 0; s_slots = 0; s_nh = 0; s_base = 0; s_abort = 0; }
    s1s_sync();

#ifdef CF_PROFILE
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
    int my_base = 0;
    bool issuer = false;   // this wave arrived last in the chunk's count and holds its key base (wave-uniform)
    for (int ch = ch0; ch < ch_end; ch++) {
        const int pid0 = ch * CAND_CHUNK_PIDS;
        uint32_t* const cb = bm + ((ch & 1) * 2 + 0) * CAND_CHUNK_WORDS;
        uint32_t* const hb = bm + ((ch & 1) * 2 + 1) * CAND_CHUNK_WORDS;
        uint32_t mc_e2 = 0, mq_e2 = 0;
        if (ch + 1 < ch_end) {   // the end of the NEXT chunk's slices
            if (lane < mc.n) mc_e2 = a.chunk_tab[(size_t)mc.c * (a.nchunks + 1) + ch + 2];
            if (lane < mq.n) mq_e2 = a.chunk_tab[(size_t)mq.c * (a.nchunks + 1) + ch + 2];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this chunk's slices, requested a chunk ago
        CF_STAMP(0);
        // ---- mark ----
        int ncl = mc.n, nql = mq.n;
        asm volatile("" : "+s"(ncl), "+s"(nql));
        auto mark_list = [&](const cf_slices& m, int u, int raw, bool first_group, uint32_t* dst) {
            const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)m.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)m.e, u);
            uint32_t from = sv;
            if (first_group) {
                if (sv + lane < ev) { const int p = raw - pid0; atomicOr(&dst[p >> 5], 1u << (p & 31)); }
                from = sv + 64u;
            }
            if (from < ev) {   // rare: entries beyond the first 64, lists beyond the first group
                const int32_t* ptr = list_ptr(m, u);
                for (uint32_t x = from + lane; x < ev; x += 64) { const int p = ptr[x] - pid0; atomicOr(&dst[p >> 5], 1u << (p & 31)); }
                S1S_DRAIN();
            }
        };
#pragma unroll
        for (int u = 0; u < CF_FC; u++)
            if (u < ncl) mark_list(mc, u, raw_c[u], true, cb);
        for (int u = CF_FC; u < mc.n; u++) mark_list(mc, u, 0, false, cb);
#pragma unroll
        for (int u = 0; u < CF_DFQ; u++)
            if (u < nql) mark_list(mq, u, raw_q[u], true, hb);
        for (int u = CF_DFQ; u < mq.n; u++) mark_list(mq, u, 0, false, hb);
        CF_STAMP(1);
        s1s_sync();
        CF_STAMP(2);
        // ---- words: bitmaps out, counts, a slot for every hit candidate ----
        {
            uint32_t hw_[CAND_CHUNK_WORDS / CF_THREADS];
            int mine = 0, nslot = 0;
#pragma unroll
            for (int h = 0; h < CAND_CHUNK_WORDS / CF_THREADS; h++) {
                const int w = tid + h * CF_THREADS;
                const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + w;
                uint32_t cw = cb[w];
                const uint32_t hw = hb[w];
                if (gw < a.words) {
                    a.cand_bits[(size_t)b * a.words + gw] = cw;
                    a.hit_bits[(size_t)b * a.words + gw] = hw;
                } else {
                    cw = 0u;
                }
                hw_[h] = cw & hw;
                mine += __popc(cw) | (__popc(cw & hw) << 16);
                nslot += __popc(cw & hw);
            }
            const int incl = flmr_wave_inclusive_scan(nslot, lane);
            int wbase = 0;
            if (lane == 63) wbase = atomicAdd(&s_slots, incl);
            wbase = __builtin_amdgcn_readlane(wbase, 63);
            int slot = wbase + incl - nslot;
#pragma unroll
            for (int h = 0; h < CAND_CHUNK_WORDS / CF_THREADS; h++) {
                const int w = tid + h * CF_THREADS;
                wslot[w] = (uint16_t)slot;
                uint32_t bits = hw_[h];
                while (bits) {
                    const int bit = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    if (slot < CF_DSLOTS) spid[slot] = (uint16_t)(w * 32 + bit);
                    slot++;
                }
            }
            const int wsum = cf_wave_sum(mine);
            int last = 0;
            if (lane == 0) {
                atomicAdd(&s_tot, wsum);
                if (atomicAdd(&s_arr, 1) == CF_WAVES - 1) {   // every wave's total is in
                    const uint32_t tot = (uint32_t)atomicExch(&s_tot, 0);
                    s_arr = 0;
                    const int cnt = (int)(tot & 0xffffu), nh = (int)(tot >> 16);
                    a.chunk_cnt[(size_t)b * a.nchunks + ch] = cnt;
                    a.chunk_hits[(size_t)b * a.nchunks + ch] = nh;
                    s_nh = nh;
                    if (nh > CF_DSLOTS) { s_abort = 1; atomicMax(redo, 4); }
                    int zero = 0;
                    asm volatile("" : "+v"(zero));
                    my_base = atomicAdd(kcount + zero, nh);   // (first read behind the pair pass)
                    last = 1;
                }
            }
            issuer = __builtin_amdgcn_readfirstlane(last) != 0;
        }
        CF_STAMP(3);
        // ---- the next chunk's slices ----
        if (ch + 1 < ch_end) {
            int ncp = mc.n;
            asm volatile("" : "+s"(ncp));
#pragma unroll
            for (int u = 0; u < CF_FC; u++) raw_c[u] = issue1(mc, ncp, u, mc.e, mc_e2);
        }
        s1s_sync();
        CF_STAMP(4);
        if (s_abort) return;   // (block-uniform: written before this barrier only)
        // ---- pairs: every (list, passage) pair of a hit candidate folds its list's row into the slot's column maxima ----
        {
            auto fold = [&](int p, int rowk) {
                int slot = -1;
                if (p >= 0) {
                    const uint32_t hbits = cb[p >> 5] & hb[p >> 5], bit = 1u << (p & 31);
                    if (hbits & bit) slot = (int)wslot[p >> 5] + __popc(hbits & (bit - 1u));
                }
                // Lane = COLUMN for the fold: one ds_max takes two pairs (a half-wave each, 32 consecutive words: no bank conflicts)
                // and only pairs that have a slot are visited.  With lane = pair it was 32 ds_max per LIST whatever its ~14 entries:
                // ~3 k LDS atomic instructions per chunk and workgroup, and the CU's LDS pipe -- not the VALUs -- set the pace
                // (24 us per chunk, as in the slot kernel's dense path); this is ~650.  The list's slots are first packed into a
                // per-wave LDS array (ballot ranks), so that an iteration is one broadcast read of two slots, a select, an address
                // and the ds_max (with v_readlane from the scattered lanes it was ~12 instructions per two pairs: 22 k of a
                // chunk's 41 k clocks).
                const unsigned long long m = __builtin_amdgcn_ballot_w64(slot >= 0);
                const int n = __popcll(m);
                if (n) {   // (wave-uniform)
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (slot >= 0) vs[rank] = slot;
                    if (slot >= 0 && rank == n - 1) vs[n] = slot;   // an odd pair out: the second half-wave repeats it (max is idempotent)
                    for (int t = 0; t < n; t += 2) {
                        const int2 s01 = *reinterpret_cast<const int2*>(vs + t);
                        atomicMax(acc + (lane < 32 ? s01.x : s01.y) * S1S_STRIDE + k, rowk);
                    }
                }
            };
            int nqe = mq.n;
            asm volatile("" : "+s"(nqe));
#pragma unroll
            for (int u = 0; u < CF_DFQ; u++) {
                if (u < nqe) {
                    const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                    if (sv < ev) fold(sv + lane < ev ? raw_q[u] - pid0 : -1, row_of(wave + CF_WAVES * u));
                }
            }
            for (int u = 0; u < mq.n; u++) {   // slices longer than 64 entries, lists beyond the first CF_DFQ of a wave
                const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)mq.s, u), ev = (uint32_t)__builtin_amdgcn_readlane((int)mq.e, u);
                const uint32_t from = u < CF_DFQ ? sv + 64u : sv;
                if (from < ev) {
                    const int32_t* ptr = list_ptr(mq, u);
                    const int rowk = row_of(wave + CF_WAVES * u);
                    for (uint32_t x0 = from; x0 < ev; x0 += 64) {
                        fold(x0 + lane < ev ? ptr[x0 + lane] - pid0 : -1, rowk);
                        S1S_DRAIN();
                    }
                }
            }
        }
        CF_STAMP(5);
        if (ch + 1 < ch_end) {   // the surviving lists' next slices (their registers are free now)
            int nqp = mq.n;
            asm volatile("" : "+s"(nqp));
#pragma unroll
            for (int u = 0; u < CF_DFQ; u++) raw_q[u] = issue1(mq, nqp, u, mq.e, mq_e2);
        }
        if (issuer) {
            if (lane == 0) s_base = my_base;
            issuer = false;
        }
        s1s_sync();
        CF_STAMP(6);
        // ---- sums: half a wave per slot; the slot's row is left at its start value, this thread's bitmap words at zero ----
        {
            const int nh = s_nh, base = s_base;
            for (int slot = tid >> 5; slot < nh; slot += 2 * CF_WAVES) {   // (nh <= CF_DSLOTS)
                const int v = acc[slot * S1S_STRIDE + k];
                acc[slot * S1S_STRIDE + k] = init;
                const float mv = s1s_dec(v);
                const float x = k < nqc ? (F16 ? flmr_round_f16(mv) : mv) : 0.0f;
                float sum = k == 0 ? 0.0f + x : x;
#pragma unroll
                for (int t = 1; t < 32; t++)
                    sum = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x138, 0xF, 0xF, false)) + x;
                if (k == 31) {
                    const float sc = F16 ? flmr_round_f16(sum) : sum;
                    const int64_t pos = (int64_t)base + slot;
                    if (pos < a.cand_cap) keys_b[pos] = flmr_make_key(sc, pid0 + (int)spid[slot]);
                }
            }
#pragma unroll
            for (int h = 0; h < CAND_CHUNK_WORDS / CF_THREADS; h++) { cb[tid + h * CF_THREADS] = 0u; hb[tid + h * CF_THREADS] = 0u; }
            if (tid == 0) s_slots = 0;
        }
        CF_STAMP(7);
        // next chunk (its marking uses the other pair of bitmaps; slots, counters and accumulators are next touched behind its first barrier)
        mc.s = mc.e; mc.e = mc_e2;
        mq.s = mq.e; mq.e = mq_e2;
    }
#ifdef CF_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&cf_prof[i], (unsigned long long)pt[i]);
        atomicAdd(&cf_prof[10], (unsigned long long)(ch_end - ch0));
        atomicAdd(&cf_prof[11], 1ull);
    }
#endif
}

// ---- kernel B: bitmap chunk -> ascending pids at the chunk's global rank, one hit flag per candidate --------------------------
__global__ __launch_bounds__(1024) void cand_emit_kernel(const uint32_t* cand_bits, const uint32_t* hit_bits, int64_t words,
                                                         const int32_t* chunk_cnt, int nchunks, int32_t* cand, int64_t cand_cap,
                                                         uint8_t* cand_hit, int32_t* cand_count, int32_t* overflow,
                                                         const int32_t* skip, const int32_t* key_count, const int32_t* chunk_hits,
                                                         uint64_t* keys, const int32_t* q_lens, int nq_cand, int n_select, int f16_round) {
    __shared__ int scan_lds[17];
    __shared__ int base_lds;
    const int b = blockIdx.x, tid = threadIdx.x;
    // (gridDim.y <= nchunks: a workgroup walks chunks blockIdx.y, + gridDim.y, ... -- on the scatter path nearly every query
    // leaves at the first test below, and 8 k workgroups that only do that cost 11 us per launch against 4 for 1 k)
    for (int ch = blockIdx.y; ch < nchunks; ch += gridDim.y) {
    __syncthreads();   // (base_lds / scan_lds of the previous chunk have been read)
    if (skip && skip[b]) {
        // Stage 1 of this query was done by scatter: nobody reads the ascending list (taps rebuild it), and the keys of its
        // key_count[b] hit candidates are in place.  The other candidates all score the all-miss constant, below every hit:
        // they only matter when the selection keeps more than there are hits (or the query is empty and everything ties) --
        // then they are appended here, chunk by chunk at deterministic positions.
        const int nhit = key_count[b];
        const int qlen = q_lens ? q_lens[b] : nq_cand;
        const int nqc = qlen < nq_cand ? qlen : nq_cand;
        if (nhit >= n_select && nqc > 0) {
            if (ch == 0 && tid == 0) cand_count[b] = (int32_t)(nhit < cand_cap ? nhit : cand_cap);
            return;   // (block-uniform, and the same for every chunk of this query)
        }
        if (tid == 0) base_lds = 0;
        __syncthreads();
        int part = 0;
        for (int c = tid; c < ch; c += 1024) part += chunk_cnt[(size_t)b * nchunks + c] - chunk_hits[(size_t)b * nchunks + c];
        if (part) atomicAdd(&base_lds, part);
        __syncthreads();
        const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + tid;
        uint32_t bits = 0;
        if (gw < words) bits = cand_bits[(size_t)b * words + gw] & ~hit_bits[(size_t)b * words + gw];
        int total;
        int64_t pos = (int64_t)nhit + base_lds + flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
        const float miss_score = flmr_miss_score(nqc, f16_round);
        uint64_t* kb = keys + (size_t)b * cand_cap;
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            if (pos < cand_cap) kb[pos] = flmr_make_key(miss_score, (int32_t)(gw * 32 + bit));
            pos++;
        }
        if (ch == nchunks - 1 && tid == 0) {
            int64_t n = (int64_t)nhit + base_lds + total;
            if (n > cand_cap) { atomicExch(overflow, 1); n = cand_cap; }
            cand_count[b] = (int32_t)n;
        }
        continue;
    }
    if (tid == 0) base_lds = 0;
    __syncthreads();
    int part = 0;
    for (int c = tid; c < ch; c += 1024) part += chunk_cnt[(size_t)b * nchunks + c];
    if (part) atomicAdd(&base_lds, part);
    __syncthreads();
    const int64_t gw = (int64_t)ch * CAND_CHUNK_WORDS + tid;
    uint32_t bits = 0, hbits = 0;
    if (gw < words) { bits = cand_bits[(size_t)b * words + gw]; hbits = hit_bits[(size_t)b * words + gw]; }
    int total;
    int64_t pos = (int64_t)base_lds + flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
    int32_t* out = cand + (size_t)b * cand_cap;
    uint8_t* outh = cand_hit + (size_t)b * cand_cap;
    while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        if (pos < cand_cap) { out[pos] = (int32_t)(gw * 32 + bit); outh[pos] = (uint8_t)((hbits >> bit) & 1u); }
        pos++;
    }
    if (ch == nchunks - 1 && tid == 0) {
        int64_t n = (int64_t)base_lds + total;
        if (n > cand_cap) { atomicExch(overflow, 1); n = cand_cap; }
        cand_count[b] = (int32_t)n;
    }
    }
}

// the surviving centroids' list, ranks and compact score rows alone (FLMR_CAND_IMPL=atomic: the first candidate generation has no
// use for the list, but stage 1 reads the rows)
int flmr_launch_qualifying(const flmr_cand_args& a, hipStream_t st) {
    hipLaunchKernelGGL(qualifying_kernel, dim3(a.nqueries), dim3(1024), 0, st, a.idx_bits, a.idx_words, a.ivf_offsets, a.cells,
                       a.ncell, a.max_cells, a.qual, a.nqual, a.qmax, 1 << S1S_IDBITS, a.hit_valid, nullptr, 2, a.idx_prefix, a.rows_out,
                       a.cen16, a.q_hi, a.q_lo, a.row_ovf, nullptr, nullptr);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_candidates_chunked(const flmr_cand_args& a, hipStream_t st) {
    hipLaunchKernelGGL(qualifying_kernel, dim3(a.nqueries), dim3(1024), 0, st, a.idx_bits, a.idx_words, a.ivf_offsets, a.cells,
                       a.ncell, a.max_cells, a.qual, a.nqual, a.qmax, 1 << S1S_IDBITS, a.hit_valid, a.scatter ? a.key_count : nullptr,
                       a.scatter ? 8 : 2, a.idx_prefix, a.rows_out, a.cen16, a.q_hi, a.q_lo, a.row_ovf, a.scatter ? a.fast_state : nullptr,
                       (a.scatter && a.fast_state && a.s1d_mode) ? a.s1d_any : nullptr);
    if (a.scatter) {
        const size_t lds = (size_t)CAND_CHUNK_WORDS * (2 * sizeof(uint32_t) + 2 * sizeof(uint16_t)) +
                           ((size_t)S1S_SLOTS * S1S_STRIDE + 96 + 1024 + S1S_QCAP) * sizeof(int) + S1S_QCAP * sizeof(uint16_t);   // + scratch words of slot-less lanes, list constants, queue (+ its passages)
        const void* kfn = a.f16_round ? reinterpret_cast<const void*>(cand_mark_score_kernel<true>) : reinterpret_cast<const void*>(cand_mark_score_kernel<false>);
        FLMR_HIP(hipFuncSetAttribute(kfn,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // chunks per workgroup: as many as still leave two workgroups per CU (the per-workgroup set-up -- list ids, offsets,
        // chunk table, list constants -- costs about a third of a chunk; 8 vs 4 measured -3 % at 256 queries x 31 chunks)
        int cpb = 8;
        while (cpb > 1 && (int64_t)a.nqueries * ((a.nchunks + cpb - 1) / cpb) < 512) cpb >>= 1;
        if (a.fast_state) {   // the plan per query (measured on a sample of its chunks), then the queue form; what it hands over is done by the slot kernel below
            hipLaunchKernelGGL(cand_plan_kernel, dim3(a.nqueries), dim3(64 * PLAN_WAVES), 0, st, a);
            const size_t flds = (size_t)9 * CAND_CHUNK_WORDS * sizeof(uint32_t) + (size_t)2 * CF_KCAP * sizeof(uint32_t) + (size_t)CF_MAXLISTS * sizeof(uint32_t) +
                                (size_t)3 * CF_QCAP * sizeof(uint32_t) + (size_t)CF_RC * 32 * sizeof(int);
            const void* ffn = a.f16_round ? reinterpret_cast<const void*>(cand_fast_kernel<true>) : reinterpret_cast<const void*>(cand_fast_kernel<false>);
            FLMR_HIP(hipFuncSetAttribute(ffn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
            int fcpb = 8;   // two resident workgroups per CU: twice the slot kernel's workgroups
            while (fcpb > 1 && (int64_t)a.nqueries * ((a.nchunks + fcpb - 1) / fcpb) < 1024) fcpb >>= 1;
            if (a.f16_round)
                hipLaunchKernelGGL(cand_fast_kernel<true>, dim3(a.nqueries, (a.nchunks + fcpb - 1) / fcpb), dim3(CF_THREADS), flds, st, a, fcpb);
            else
                hipLaunchKernelGGL(cand_fast_kernel<false>, dim3(a.nqueries, (a.nchunks + fcpb - 1) / fcpb), dim3(CF_THREADS), flds, st, a, fcpb);
            {   // the small-dense form: the queries planned for it
                const size_t dlds = (size_t)4 * CAND_CHUNK_WORDS * sizeof(uint32_t) + ((size_t)CF_DSLOTS * S1S_STRIDE + 64) * sizeof(int) +
                                    (size_t)CF_DRC * 32 * sizeof(int) + ((size_t)CAND_CHUNK_WORDS + CF_DSLOTS) * sizeof(uint16_t);
                const void* dfn = a.f16_round ? reinterpret_cast<const void*>(cand_dense_small_kernel<true>) : reinterpret_cast<const void*>(cand_dense_small_kernel<false>);
                FLMR_HIP(hipFuncSetAttribute(dfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                if (a.f16_round)
                    hipLaunchKernelGGL(cand_dense_small_kernel<true>, dim3(a.nqueries, (a.nchunks + fcpb - 1) / fcpb), dim3(CF_THREADS), dlds, st, a, fcpb);
                else
                    hipLaunchKernelGGL(cand_dense_small_kernel<false>, dim3(a.nqueries, (a.nchunks + fcpb - 1) / fcpb), dim3(CF_THREADS), dlds, st, a, fcpb);
#ifdef CF_PROFILE
                {
                    unsigned long long h[12];
                    (void)hipDeviceSynchronize();
                    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(cf_prof), sizeof(h));
                    if (h[10]) fprintf(stderr, "[cfd] blocks %llu chunks %llu; ticks per chunk: top %.0f mark %.0f barrier1 %.0f words %.0f prefetch+barrier2 %.0f pairs %.0f prefetch+barrier3 %.0f sums %.0f\n",
                            h[11], h[10], (double)h[0] / h[10], (double)h[1] / h[10], (double)h[2] / h[10], (double)h[3] / h[10], (double)h[4] / h[10],
                            (double)h[5] / h[10], (double)h[6] / h[10], (double)h[7] / h[10]);
                }
#endif
            }
#ifdef CF_PROFILE
            {
                unsigned long long h[12];
                (void)hipDeviceSynchronize();
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(cf_prof), sizeof(h));
                fprintf(stderr, "[cf] blocks %llu chunks %llu; ticks per chunk: loop-top %.0f wait %.0f mark %.0f barrier %.0f words %.0f prefetch %.0f pairs %.0f deferred %.0f\n",
                        h[11], h[10], (double)h[0] / h[10], (double)h[1] / h[10], (double)h[2] / h[10], (double)h[3] / h[10], (double)h[4] / h[10],
                        (double)h[5] / h[10], (double)h[6] / h[10], (double)h[7] / h[10]);
                unsigned long long z[12] = {};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(cf_prof), z, sizeof(z));
            }
#endif
        }
        if (a.f16_round)
            hipLaunchKernelGGL(cand_mark_score_kernel<true>, dim3(a.nqueries, (a.nchunks + cpb - 1) / cpb), dim3(64 * S1S_WAVES), lds, st, a, cpb);
        else
            hipLaunchKernelGGL(cand_mark_score_kernel<false>, dim3(a.nqueries, (a.nchunks + cpb - 1) / cpb), dim3(64 * S1S_WAVES), lds, st, a, cpb);
#ifdef S1S_PROFILE
        {
            unsigned long long h[12];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(s1s_prof), sizeof(h));
            fprintf(stderr, "[s1s] blocks %llu chunks %llu; 100 MHz ticks per chunk: prologue/blk %.1f init %.1f wait %.1f mark %.1f scan %.1f prefetch %.1f scatter %.1f sum %.1f keys %.1f end %.1f\n",
                    h[11], h[10], (double)h[0] / h[11], (double)h[1] / h[10], (double)h[2] / h[10], (double)h[3] / h[10], (double)h[4] / h[10],
                    (double)h[5] / h[10], (double)h[6] / h[10], (double)h[7] / h[10], (double)h[8] / h[10], (double)h[9] / h[10]);
            unsigned long long z[12] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(s1s_prof), z, sizeof(z));
            int nqh[1024], nch[1024];
            const int nb = a.nqueries < 1024 ? a.nqueries : 1024;
            (void)hipMemcpy(nqh, a.nqual, nb * sizeof(int), hipMemcpyDeviceToHost);
            (void)hipMemcpy(nch, a.ncell, nb * sizeof(int), hipMemcpyDeviceToHost);
            long sq = 0, sc = 0; int mq = 0;
            for (int i = 0; i < nb; i++) { sq += nqh[i]; sc += nch[i]; mq = nqh[i] > mq ? nqh[i] : mq; }
            fprintf(stderr, "[s1s] queries %d cpb %d nchunks %d: surviving lists mean %.1f max %d, probed cells mean %.1f\n", a.nqueries, cpb, a.nchunks,
                    (double)sq / nb, mq, (double)sc / nb);
        }
#endif
    } else {
        hipLaunchKernelGGL(cand_mark_chunks_kernel, dim3(a.nqueries, a.nchunks), dim3(256), 0, st, a.cells, a.ncell, a.max_cells,
                           a.qual, a.nqual, a.qmax, a.hit_valid, a.ivf_pids, a.ivf_offsets, a.chunk_tab, a.nchunks, a.cand_bits,
                           a.hit_bits, a.words, a.chunk_cnt);
    }
    int ey = (int)flmr_ceil_div(1024, a.nqueries);   // ~1024 workgroups; every chunk its own workgroup for small batches
    if (ey > a.nchunks) ey = a.nchunks;
    hipLaunchKernelGGL(cand_emit_kernel, dim3(a.nqueries, a.scatter ? ey : a.nchunks), dim3(1024), 0, st, a.cand_bits, a.hit_bits, a.words,
                       a.chunk_cnt, a.nchunks, a.cand, a.cand_cap, a.cand_hit, a.cand_count, a.overflow,
                       a.scatter ? a.hit_valid : nullptr, a.key_count, a.chunk_hits, a.keys, a.q_lens, a.nq_cand, a.n_select, a.f16_round);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// the ascending candidate lists of every query of the last batch (FLMR_TAP_CANDIDATES after a scatter-mode search)
int flmr_launch_cand_emit_all(const flmr_cand_args& a, hipStream_t st) {
    hipLaunchKernelGGL(cand_emit_kernel, dim3(a.nqueries, a.nchunks), dim3(1024), 0, st, a.cand_bits, a.hit_bits, a.words,
                       a.chunk_cnt, a.nchunks, a.cand, a.cand_cap, a.cand_hit, a.cand_count, a.overflow, nullptr, nullptr, nullptr, nullptr,
                       nullptr, 0, 0, 0);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
