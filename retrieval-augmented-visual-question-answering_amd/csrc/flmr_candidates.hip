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
// all-miss value, and stage 1 never reads its codes.  A second kernel turns (bitmap, per-chunk counts) into the
// ascending pid list with one hit flag per candidate.
#include "flmr_device.h"

#define CAND_CHUNK_WORDS 1024                      // 32768 passages per chunk
#define CAND_CHUNK_PIDS (CAND_CHUNK_WORDS * 32)

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

// ---- qualifying centroids of each query: compact list of the set bits of idx (if few enough) -------------------------
// grid = nqueries, block = 1024.  hit_valid[q] = 1 when the list fits and the qualifying lists are not longer than
// twice the probed-cell lists (otherwise marking would cost more than it saves and stage 1 scans every candidate).
__global__ __launch_bounds__(1024) void qualifying_kernel(const uint32_t* idx_bits, int idx_words,
                                                          const int64_t* ivf_offsets, const int32_t* cells,
                                                          const int32_t* ncell, int max_cells, int32_t* qual,
                                                          int32_t* nqual, int qmax, int32_t* hit_valid) {
    __shared__ int scan_lds[17];
    __shared__ unsigned long long tot_q, tot_c;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { tot_q = 0ull; tot_c = 0ull; }
    __syncthreads();
    int base = 0;
    unsigned long long mylen = 0;
    for (int w0 = 0; w0 < idx_words; w0 += 1024) {
        const int w = w0 + tid;
        uint32_t bits = (w < idx_words) ? idx_bits[(size_t)b * idx_words + w] : 0u;
        int total;
        int pos = base + flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
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
    if (tid == 0) {
        nqual[b] = base < qmax ? base : qmax;
        hit_valid[b] = (base <= qmax) && (tot_q <= 2ull * tot_c);
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

// ---- kernel B: bitmap chunk -> ascending pids at the chunk's global rank, one hit flag per candidate --------------------------
__global__ __launch_bounds__(1024) void cand_emit_kernel(const uint32_t* cand_bits, const uint32_t* hit_bits, int64_t words,
                                                         const int32_t* chunk_cnt, int nchunks, int32_t* cand, int64_t cand_cap,
                                                         uint8_t* cand_hit, int32_t* cand_count, int32_t* overflow) {
    __shared__ int scan_lds[17];
    __shared__ int base_lds;
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
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

int flmr_launch_candidates_chunked(const flmr_cand_args& a, hipStream_t st) {
    hipLaunchKernelGGL(qualifying_kernel, dim3(a.nqueries), dim3(1024), 0, st, a.idx_bits, a.idx_words, a.ivf_offsets, a.cells,
                       a.ncell, a.max_cells, a.qual, a.nqual, a.qmax, a.hit_valid);
    hipLaunchKernelGGL(cand_mark_chunks_kernel, dim3(a.nqueries, a.nchunks), dim3(256), 0, st, a.cells, a.ncell, a.max_cells,
                       a.qual, a.nqual, a.qmax, a.hit_valid, a.ivf_pids, a.ivf_offsets, a.chunk_tab, a.nchunks, a.cand_bits,
                       a.hit_bits, a.words, a.chunk_cnt);
    hipLaunchKernelGGL(cand_emit_kernel, dim3(a.nqueries, a.nchunks), dim3(1024), 0, st, a.cand_bits, a.hit_bits, a.words,
                       a.chunk_cnt, a.nchunks, a.cand, a.cand_cap, a.cand_hit, a.cand_count, a.overflow);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
