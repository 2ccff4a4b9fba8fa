// Stage 2 as a query-stationary dense walk over the centroid table ("walk" kernel), plus the per-passage sorted code
// copy it reads.
//
// Reference: TPC/search/filter_pids.cpp:27-69 with idx == all ones (second call of filter_pids_helper, :143-157): per
// surviving passage, per query token k, the maximum over the passage's tokens of centroid_scores[code, k], summed over k
// in ascending order.
//
// Why another stage-2 kernel.  The gather form (filter_stage2_lds_kernel) fetches one 256-byte fp16 centroid row per
// survivor token: 131072 random rows per query at BASELINE's shape, 34 GB per 1024-query launch, nearly all of them L2
// misses served by the Infinity Cache -- and random 256-byte gathers from a 32 MB table top out at 9.3-9.6 TB/s on this
// chip (profiles/microbench/s2_design_probe: 25-32 TB/s when the table fits an XCD's 4 MB L2, 34 TB/s for sequential tiles
// that the CUs of an XCD walk together).  A query's 1024 survivors hold as many tokens as there are centroids, so the
// dense product `centroids x Q^T` for ALL K rows costs the same MFMA work as the gathered one -- and its operand stream is
// sequential and shared: every workgroup walks the table in the same order at the same pace, each XCD's L2 loads a slice
// once and serves it to its 32 CUs.
//
// One workgroup (8 waves) owns a (query, block of 1024 survivors) item.  Every lane owns TWO passages and keeps their 32
// running column maxima in registers (2 x 32 VGPRs) -- no atomics, no LDS accumulators.  The table is walked in slices of
// 1024 centroids, each in two strict phases separated by barriers:
//   produce  every wave computes four 32-row tiles of the slice's [1024 x 32] fp32 score tile with exactly the MFMA sequence
//            of stage 0 (2 x v_mfma_f32_32x32x16_f16 per 16 dims against q_hi / q_lo, then fma(lo, 2^-11, hi): bitwise the
//            values stage 0 produces) and stores them to the 144 KB LDS buffer; the A operand comes from a copy of the
//            table laid out in MFMA order (one contiguous 1 KB load per k-step), tile t+1 in flight behind tile t's MFMAs;
//   consume  a lane walks its passages' codes in ASCENDING order (`codes_sorted`, built once at flmr_index_open) and, for
//            every code inside the slice, reads that row from LDS (8 x ds_read_b128 at immediate offsets from one address;
//            rows are 144 bytes apart so that random rows spread over the banks) and folds it into its maxima.
// max is exact, so the token order does not matter: the maxima -- and the k-ascending fp32 sum -- are bit-identical to the
// gather kernels'.  (A double-buffered variant with 512-row slices that overlaps the phases fits the 160 KB of LDS too, but
// a lane then meets half as many of its codes per slice and the consumer loop -- whose trip count is the LONGEST per-lane
// run in the wave -- runs at 15 % lane utilisation instead of 22 %; measured slower.)
//
// Bound: the L2 -> CU stream of the table in the produce phase (one pass over all K rows per item: 256 KB per 1024-row slice
// per CU; 34 TB/s chip-wide is the measured ceiling of that stream, = the MFMA time of the slice) and LDS instruction issue in
// the consume phase.  Measured with profiles/microbench/s2_walk_probe (K = 131072, 1024 queries x 1024 survivors x 128
// codes): 5.1 ms per launch -- per 1024-row slice and wave 11.4 k cycles produce, 6.1 k consume, 3.5 k barrier wait --
// against 4.06 ms for the gather kernel on the same shape, so at BASELINE's shape (tokens per query == K) the gather stays
// the default.  The walk's cost is proportional to K, the gather's to the survivors' token count: the cost model below
// picks the walk when the table is the shorter of the two (the 10 k / 160 k-passage corpora with K = 16384 / 65536).
#include "flmr_device.h"

typedef _Float16 w2h8 __attribute__((ext_vector_type(8)));
typedef int w2i4 __attribute__((ext_vector_type(4)));
typedef int w2i4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address

// The running maxima are kept as order-preserving SIGNED-INTEGER images of the fp32 scores (positive floats keep their bits,
// negative ones have the low 31 bits flipped), so that the consumer's inner operation is one v_max_i32 per column.
// fmaxf() lowers to three VALU ops (both operands are canonicalised first for IEEE signalling-NaN semantics) and drags
// register copies through the loop; the integer form costs three VALU ops per PRODUCED score instead -- a sixth as many.
// The values are MFMA results and -inf, never NaN; -0.0 orders just below +0.0, which no sum can tell apart.
__device__ __forceinline__ int w2_enc(float v) {
    const int b = __float_as_int(v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float w2_dec(int e) { return __int_as_float(e ^ ((e >> 31) & 0x7fffffff)); }

// -DW2_PROFILE (profiles/microbench/s2_walk_probe.hip): per-wave cycle counts of the three phases of a slice
#ifdef W2_PROFILE
#define W2_CLK(x) const long long x = __builtin_readcyclecounter()
#define W2_ACC(k, v) prof_acc[k] += (v)
#else
#define W2_CLK(x)
#define W2_ACC(k, v)
#endif

#define W2_WAVES 8
#define W2_SLICE 1024                // centroid rows per slice (the LDS score buffer)
#define W2_TPW (W2_SLICE / 32 / W2_WAVES)   // tiles per wave and slice
#define W2_STRIDE 36                 // floats per staged score row: 144 B, so that the rows random lanes read start in 16
                                     // different bank groups and every ds_read_b128 stays 16-byte aligned (144 KB of LDS)
#define W2_DOCS (W2_WAVES * 128)     // passages per work item: two per lane
#define W2_BUF ((W2_SLICE + 1) * W2_STRIDE)   // words per buffer: the score rows + one dummy row (the smallest image)
#define W2_INF 0x7fffffff
#define W2_MAX_DOCLEN 2048           // longest passage the in-LDS code sort handles

// ------------------------------------------------------------------------------------------------
// codes_sorted[off[p] .. off[p+1]) = ascending copy of codes[off[p] .. off[p+1]) with the DISTINCT values first: the first
// ulen[p] entries are the passage's distinct codes in ascending order, the rest repeat the largest (the run stays ascending
// and holds the same set: every reader that takes all of it -- the walk -- computes the same maxima; the readers that take
// only the distinct prefix -- the sliced stage 2, the dense stage 1 -- do less than half the work on a real index, where a
// passage repeats its codes: 2.2 tokens per distinct (passage, centroid) pair on the built 1 M-passage index.  The
// reference's own IVF is built from the same distinct pairs, indexing/utils.py:8-53, and filter_pids.cpp:50-63 skips repeats).
// One wave per passage, bitonic in LDS.  `total_ulen`: sum of ulen over the passages.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sort_doc_codes_kernel(const int32_t* __restrict__ codes, const int64_t* __restrict__ offsets,
                                                            int64_t num_passages, int32_t* __restrict__ out,
                                                            uint16_t* __restrict__ ulen, unsigned long long* __restrict__ total_ulen) {
    __shared__ int32_t s[W2_MAX_DOCLEN];
    unsigned long long mine = 0;
    for (int64_t p = blockIdx.x; p < num_passages; p += gridDim.x) {
        const int64_t off = offsets[p];
        const int len = (int)(offsets[p + 1] - off);
        if (len <= 0) { if (threadIdx.x == 0) ulen[p] = 0; continue; }
        int n = 1;
        while (n < len) n <<= 1;
        for (int t = threadIdx.x; t < n; t += 64) s[t] = t < len ? codes[off + t] : W2_INF;
        __syncthreads();
        for (int k = 2; k <= n; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < n; t += 64) {
                    const int q = t ^ j;
                    if (q > t) {
                        const int32_t a = s[t], b = s[q];
                        const bool asc = (t & k) == 0;
                        if (asc ? (a > b) : (a < b)) { s[t] = b; s[q] = a; }
                    }
                }
                __syncthreads();
            }
        }
        // distinct values to the front (positions by a running wave scan), the largest repeated behind them
        int base = 0;
        const int32_t last = s[len - 1];
        for (int t0 = 0; t0 < len; t0 += 64) {
            const int t = t0 + threadIdx.x;
            const bool first = t < len && (t == 0 || s[t] != s[t - 1]);
            const unsigned long long m = __ballot(first);
            if (first) out[off + base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = s[t];
            base += __popcll(m);
        }
        for (int t = base + threadIdx.x; t < len; t += 64) out[off + t] = last;
        if (threadIdx.x == 0) { ulen[p] = (uint16_t)base; mine += (unsigned long long)base; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && mine) atomicAdd(total_ulen, mine);
}

// ------------------------------------------------------------------------------------------------
// centroids_f16_tiled: the fp16 centroid table re-laid in the MFMA A-operand order.  Lane (i, h) of a wave needs dims
// 64h .. 64h+63 of row i of a 32-row tile -- its own 128-byte line of the row-major table -- so loading a tile straight into
// registers takes 8 instructions that EACH touch all 64 lines of the tile: the texture-address unit retires about one line per
// cycle and the first version of the walk spent 9 000 of its 14 000 cycles per slice there (the LDS-DMA form of the same
// tile is no faster: ~6.4 TB/s chip-wide, MI355X_MICROARCH.md).  In this copy 16-byte unit [(tile*8 + s)*64 + lane] holds
// dims 64h+8s .. 64h+8s+7 of row tile*32 + i (lane = 32h + i), so step s of a tile is ONE fully contiguous 1 KB load that
// lands in MFMA layout.  33.5 MB at K = 131072, built once at flmr_index_open.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_centroids_kernel(const _Float16* __restrict__ cen16, int K, uint4* __restrict__ out) {
    const size_t n = (size_t)K * 16;  // 16-byte units
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(u & 63), s = (int)((u >> 6) & 7);
        const size_t tile = u >> 9;
        const int i = lane & 31, h = lane >> 5;
        out[u] = *reinterpret_cast<const uint4*>(cen16 + (tile * 32 + i) * FLMR_DIM + 64 * h + 8 * s);
    }
}

// Built only where the walk can be chosen at all: the cost model below (K * ceil(ndocs / 1024) <= 0.6 x survivor tokens) bounds
// K by ~0.6 * 1024 * mean passage length whatever ndocs is -- BASELINE's 1 M / 6 M-passage indexes (K = 2^17, 2^18) never take
// the walk and do not pay for the copy -- or where FLMR_S2_IMPL=walk forces it.  A failed allocation only makes the walk
// unavailable.
int flmr_build_tiled_centroids(flmr_index* ix) {
    ix->centroids_f16_tiled = nullptr;
    if (!ix->centroids_f16 || ix->K % 32 != 0 || !ix->codes_sorted) return FLMR_OK;
    const double mean_len = (double)ix->N / (double)(ix->num_passages > 0 ? ix->num_passages : 1);
    const bool can_pay = (double)ix->K <= 0.6 * W2_DOCS * mean_len;
    if (!can_pay && !flmr_process_options().is(FLMR_OPT_S2_IMPL, "walk")) return FLMR_OK;
    if (hipMalloc(reinterpret_cast<void**>(&ix->centroids_f16_tiled), (size_t)ix->K * FLMR_DIM * sizeof(_Float16)) != hipSuccess) {
        (void)hipGetLastError();
        ix->centroids_f16_tiled = nullptr;
        return FLMR_OK;
    }
    hipLaunchKernelGGL(tile_centroids_kernel, dim3(2048), dim3(256), 0, 0, ix->centroids_f16, ix->K,
                       reinterpret_cast<uint4*>(ix->centroids_f16_tiled));
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipDeviceSynchronize());
    return FLMR_OK;
}

int flmr_build_sorted_codes(flmr_index* ix) {
    ix->codes_sorted = nullptr; ix->doc_ulen = nullptr; ix->dup_share = 0.0; ix->mean_ulen = 0.0;
    if (ix->max_doclen > W2_MAX_DOCLEN || ix->N >= 0x7fffffffLL || ix->num_passages <= 0 || ix->N <= 0) return FLMR_OK;
    // Read by the sliced / walking stage-2 forms and by the dense stage 1 (flmr_stage1_dense.hip); a failed allocation only
    // makes those forms unavailable (4 bytes per token + 2 per passage)
    if (hipMalloc(reinterpret_cast<void**>(&ix->codes_sorted), ((size_t)ix->N + FLMR_CODE_PAD) * sizeof(int32_t)) != hipSuccess) {
        (void)hipGetLastError();
        ix->codes_sorted = nullptr;
        return FLMR_OK;
    }
    unsigned long long* d_total = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&ix->doc_ulen), (size_t)ix->num_passages * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_total), sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ix->codes_sorted); (void)hipFree(ix->doc_ulen); (void)hipFree(d_total);
        ix->codes_sorted = nullptr; ix->doc_ulen = nullptr;
        return FLMR_OK;
    }
    FLMR_HIP(hipMemset(ix->codes_sorted + ix->N, 0x7f, FLMR_CODE_PAD * sizeof(int32_t)));
    FLMR_HIP(hipMemset(d_total, 0, sizeof(unsigned long long)));
    const int64_t grid = ix->num_passages < 262144 ? ix->num_passages : 262144;
    hipLaunchKernelGGL(sort_doc_codes_kernel, dim3((unsigned)grid), dim3(64), 0, 0, ix->codes, ix->doc_offsets, ix->num_passages,
                       ix->codes_sorted, ix->doc_ulen, d_total);
    FLMR_LAUNCH_CHECK();
    unsigned long long total = 0;
    FLMR_HIP(hipMemcpy(&total, d_total, sizeof(total), hipMemcpyDeviceToHost));
    (void)hipFree(d_total);
    ix->dup_share = 1.0 - (double)total / (double)ix->N;
    ix->mean_ulen = (double)total / (double)ix->num_passages;
    if (ix->dup_share < 0.10) {   // nothing to gain: every token counts (the readers then take the whole run)
        (void)hipFree(ix->doc_ulen);
        ix->doc_ulen = nullptr;
        ix->mean_ulen = (double)ix->N / (double)ix->num_passages;
    }
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// The walk.  grid = min(#items, #CUs) persistent workgroups of 512 threads, item = (query, block of 1024 survivors);
// dynamic LDS = the score buffer (144 KB) + 8 KB for q_lo.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * W2_WAVES, 2) void filter_stage2_walk_kernel(
    flmr_filter_args f, const int32_t* __restrict__ pids, int64_t pid_stride, const int32_t* __restrict__ counts,
    uint64_t* __restrict__ keys, int64_t key_stride, const _Float16* __restrict__ cen16t, const _Float16* __restrict__ q_hi,
    const _Float16* __restrict__ q_lo, const int32_t* __restrict__ codes_sorted, int nchunks, int nitems
#ifdef W2_PROFILE
    , long long* prof
#endif
    ) {
#ifdef W2_PROFILE
    long long prof_acc[3] = {0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const buf = reinterpret_cast<int*>(smem);  // [W2_SLICE + 1][W2_STRIDE] score images (w2_enc)
    w2h8* const blds = reinterpret_cast<w2h8*>(smem + (size_t)W2_BUF * sizeof(int));  // [8 steps][64 lanes] q_lo in MFMA order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int K = f.K;
    const int nslices = (K + W2_SLICE - 1) / W2_SLICE;
    if (threadIdx.x < W2_STRIDE) buf[W2_SLICE * W2_STRIDE + threadIdx.x] = (int)0x80000000;  // the dummy row behind the score rows

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / nchunks, chunk = item - b * nchunks;
        const int cnt = counts[b];
        const int base = chunk * W2_DOCS;
        if (base >= cnt) continue;  // block-uniform
        const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
        const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;  // <= 32 on this path

        // B operand: this query's fp16 split, lane (i, h) holds dims 64h .. 64h+63 of query token i.  q_hi stays in registers
        // (32 VGPRs); q_lo lives in LDS in MFMA order ([step][lane] 16-byte units, 8 KB, the same for every wave) and is read
        // one step ahead of its MFMA -- with both halves in registers the kernel does not fit the 256-VGPR budget of two
        // waves per SIMD next to the two A tiles, the accumulators and the 64 running maxima.
        w2h8 bh[8];
        {
            const w2h8* ph = reinterpret_cast<const w2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
            for (int s = 0; s < 8; s++) bh[s] = ph[s];
            const int u = threadIdx.x, us = u >> 6, ui = u & 31, uh = (u >> 5) & 1;   // 512 threads <-> 512 units
            blds[u] = *reinterpret_cast<const w2h8*>(q_lo + ((size_t)b * f.ncol + ui) * FLMR_DIM + 64 * uh + 8 * us);
        }
        __syncthreads();

        // this lane's two passages: slots base + wave*128 + {0, 64} + lane of the survivor list.  Their ascending codes are
        // streamed through 4-entry register windows.  `ps` is the array index of the next unread code (= cw.x), cw holds
        // [ps, cwe), nw holds [cwe, nwe) (nwe == cwe: none).  A third window `st` is in flight: it is requested once per
        // slice, BEFORE the slice's MFMA work, from the position behind nw, and becomes nw at the end of the slice if nw was
        // used up meanwhile -- so the consumer loops contain no memory loads (a load inside their divergent flow costs a
        // full-wave gather, merge copies and a vmcnt(0) that also waits for the prefetched A tile).  Windows are fetched
        // with one unconditional 16-byte load; entries past the passage's end are garbage and are never looked at
        // (`ps < end` guards every use); codes_sorted carries 8 words of padding.
        int pid[2];
        uint32_t ps[2], cwe[2], nwe[2], end[2];
        int cwx[2], cwy[2], cwz[2], cww[2], nwx[2], nwy[2], nwz[2], nww[2];  // (scalars: element-wise selects on vector types
                                                                             //  compile to index-compare chains)
        int m[2][32];              // running column maxima, w2_enc images (filter_pids.cpp:30-33: they start at -9999)
        auto load4 = [&](uint32_t p) { return *reinterpret_cast<const w2i4u*>(codes_sorted + p); };
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const int slot = base + wave * 128 + d * 64 + lane;
            pid[d] = -1;
            ps[d] = end[d] = 0u;
            if (slot < cnt) {
                pid[d] = pids[(size_t)b * pid_stride + slot];
                const int64_t off = f.offsets[pid[d]];
                ps[d] = (uint32_t)off;
                end[d] = (uint32_t)(off + (f.doclens ? f.doclens[pid[d]] : (f.offsets[pid[d] + 1] - off)));
            }
            const w2i4 w0 = load4(ps[d]);
            cwx[d] = w0.x; cwy[d] = w0.y; cwz[d] = w0.z; cww[d] = w0.w;
            cwe[d] = ps[d] + 4u;
            const w2i4 w1 = load4(cwe[d]);
            nwx[d] = w1.x; nwy[d] = w1.y; nwz[d] = w1.z; nww[d] = w1.w;
            nwe[d] = cwe[d] + 4u;
#pragma unroll
            for (int c = 0; c < 32; c++) m[d][c] = w2_enc(-9999.0f);
        }

        // ---- producer pieces ------------------------------------------------------------------------------------
        w2h8 a[8], a1[8];  // A operands of two tiles (ping-pong): lane (i, h) holds dims 64h .. 64h+63 of centroid row i
        auto load_a = [&](w2h8* a, int slice, int tile) {  // from the tiled copy: step s is one contiguous 1 KB load (see above)
            int t = slice * (W2_SLICE / 32) + tile;
            t = t * 32 < K ? t : 0;  // tiles past the table are never referenced by a code
            const w2h8* pc = reinterpret_cast<const w2h8*>(cen16t) + (size_t)t * 512 + lane;
#pragma unroll
            for (int s = 0; s < 8; s++) a[s] = pc[s * 64];
        };
        f32x16 ah, al;
        auto mfma_tile = [&](const w2h8* a) {
#pragma unroll
            for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
            w2h8 blo[2] = {blds[lane], blds[64 + lane]};   // q_lo, two steps ahead of its MFMA
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const w2h8 cur_lo = blo[s & 1];
                if (s + 2 < 8) blo[s & 1] = blds[(s + 2) * 64 + lane];
                ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bh[s], ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], cur_lo, al, 0, 0, 0);
            }
        };
        // C/D layout of the 32x32 MFMA: register r of lane (i, h) is row (r&3) + 8*(r>>2) + 4h, column i.  One lane-dependent
        // address, everything else is an immediate offset.
        const int wbase = ((W2_TPW * wave) * 32 + 4 * h) * W2_STRIDE + i;
        auto store_tile = [&](int tile_rel) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                buf[wbase + (tile_rel * 32 + (r & 3) + 8 * (r >> 2)) * W2_STRIDE] = w2_enc(fmaf(al[r], 1.0f / 2048.0f, ah[r]));
        };
        // this wave's four tiles of slice `s`.  `a` already holds the first (requested before the previous consumer phase);
        // every further tile is requested as soon as the registers it lands in have been read by the MFMAs of the tile
        // before last, and the first tile of slice s+1 goes out behind the third.
        auto produce = [&](int s) {
            load_a(a1, s, W2_TPW * wave + 1);
            mfma_tile(a);
            load_a(a, s, W2_TPW * wave + 2);
            store_tile(0);
            mfma_tile(a1);
            load_a(a1, s, W2_TPW * wave + 3);
            store_tile(1);
            mfma_tile(a);
            if (s + 1 < nslices) load_a(a, s + 1, W2_TPW * wave);
            store_tile(2);
            mfma_tile(a1);
            store_tile(3);
        };

        // ---- consumer: fold the rows of both passages' codes that lie in [sb, se) --------------------------------------
        // ONE wave-uniform loop serves both passages of a lane per trip (two independent LDS read streams in flight, half the
        // trips of two loops).  No control flow around the maxima: a passage without a token in the slice reads the buffer's
        // dummy row (row W2_SLICE, the smallest image) instead, so the running maxima are updated in place on every trip
        // (with an `if (active)` around them the compiler carries two copies of all 64 maxima through the loop's phi nodes;
        // with the test at the loop head it splits their live ranges and copies them on every trip -- hence guard + do/while).
        // The loop contains no memory loads: a load inside its divergent flow costs a full-wave gather, merge copies and a
        // vmcnt(0) that also waits for the prefetched A tile.
#define W2_ACT(d) ((ps[d] < cwe[d]) & (ps[d] < end[d]) & (cwx[d] < se))   /* `&`, not `&&`: no short-circuit branches */
#define W2_STEP(d, act)                                                                                          \
    {                                                                                                            \
        cwx[d] = (act) ? cwy[d] : cwx[d];                                                                        \
        cwy[d] = (act) ? cwz[d] : cwy[d];                                                                        \
        cwz[d] = (act) ? cww[d] : cwz[d];                                                                        \
        ps[d] += (act) ? 1u : 0u;                                                                                \
        const bool promote = (ps[d] == cwe[d]) & (nwe[d] != cwe[d]); /* cw used up, nw present: nw takes over */ \
        cwx[d] = promote ? nwx[d] : cwx[d];                                                                      \
        cwy[d] = promote ? nwy[d] : cwy[d];                                                                      \
        cwz[d] = promote ? nwz[d] : cwz[d];                                                                      \
        cww[d] = promote ? nww[d] : cww[d];                                                                      \
        cwe[d] = promote ? nwe[d] : cwe[d];                                                                      \
    }
        auto consume = [&](const int* src, int sb, int se) {
            if (__ballot(W2_ACT(0) | W2_ACT(1)) != 0ull) do {
                const bool act0 = W2_ACT(0), act1 = W2_ACT(1);
                const int* r0 = src + (act0 ? (cwx[0] - sb) : W2_SLICE) * W2_STRIDE;
                const int* r1 = src + (act1 ? (cwx[1] - sb) : W2_SLICE) * W2_STRIDE;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const w2i4 t = *reinterpret_cast<const w2i4*>(r0 + 4 * j);
                    m[0][4 * j + 0] = max(m[0][4 * j + 0], t.x);
                    m[0][4 * j + 1] = max(m[0][4 * j + 1], t.y);
                    m[0][4 * j + 2] = max(m[0][4 * j + 2], t.z);
                    m[0][4 * j + 3] = max(m[0][4 * j + 3], t.w);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const w2i4 t = *reinterpret_cast<const w2i4*>(r1 + 4 * j);
                    m[1][4 * j + 0] = max(m[1][4 * j + 0], t.x);
                    m[1][4 * j + 1] = max(m[1][4 * j + 1], t.y);
                    m[1][4 * j + 2] = max(m[1][4 * j + 2], t.z);
                    m[1][4 * j + 3] = max(m[1][4 * j + 3], t.w);
                }
                W2_STEP(0, act0)
                W2_STEP(1, act1)
            } while (__ballot(W2_ACT(0) | W2_ACT(1)) != 0ull);
        };

        // ---- the walk ------------------------------------------------------------------------------------------------
        load_a(a, 0, W2_TPW * wave);
        for (int s = 0; s < nslices; s++) {
            W2_CLK(t1);
            // request the window behind nw (position nwe) for both passages: it lands during the MFMA work below and becomes
            // nw at the end of the consumer phase if nw was used up meanwhile
            const uint32_t sta0 = nwe[0], sta1 = nwe[1];
            const w2i4 st0 = load4(sta0), st1 = load4(sta1);
            produce(s);
            W2_CLK(t2);
            __syncthreads();  // the slice is complete
            W2_CLK(t3);
            const int sb = s * W2_SLICE, se = sb + W2_SLICE;
            while (true) {
                consume(buf, sb, se);
                // a passage with more than 4 .. 8 codes inside one slice runs out of buffered codes (rare): fetch its next
                // window now (exposed latency) and go round again
                const bool starved0 = (ps[0] == cwe[0]) & (ps[0] < end[0]), starved1 = (ps[1] == cwe[1]) & (ps[1] < end[1]);
                if (__ballot(starved0 | starved1) == 0ull) break;
                const w2i4 x0 = load4(cwe[0]), x1 = load4(cwe[1]);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), here: keeps the wait out of the consumer loop
                cwx[0] = starved0 ? x0.x : cwx[0]; cwy[0] = starved0 ? x0.y : cwy[0];
                cwz[0] = starved0 ? x0.z : cwz[0]; cww[0] = starved0 ? x0.w : cww[0];
                cwx[1] = starved1 ? x1.x : cwx[1]; cwy[1] = starved1 ? x1.y : cwy[1];
                cwz[1] = starved1 ? x1.z : cwz[1]; cww[1] = starved1 ? x1.w : cww[1];
                cwe[0] += starved0 ? 4u : 0u; cwe[1] += starved1 ? 4u : 0u;
                nwe[0] = starved0 ? cwe[0] : nwe[0]; nwe[1] = starved1 ? cwe[1] : nwe[1];
            }
            // nw was promoted during this slice -> the staged window (it starts where nw ended) is the new nw
            {
                const bool take0 = (nwe[0] == cwe[0]) & (sta0 == cwe[0]), take1 = (nwe[1] == cwe[1]) & (sta1 == cwe[1]);
                nwx[0] = take0 ? st0.x : nwx[0]; nwy[0] = take0 ? st0.y : nwy[0];
                nwz[0] = take0 ? st0.z : nwz[0]; nww[0] = take0 ? st0.w : nww[0];
                nwx[1] = take1 ? st1.x : nwx[1]; nwy[1] = take1 ? st1.y : nwy[1];
                nwz[1] = take1 ? st1.z : nwz[1]; nww[1] = take1 ? st1.w : nww[1];
                nwe[0] += take0 ? 4u : 0u;
                nwe[1] += take1 ? 4u : 0u;
            }
            W2_CLK(t4);
            __syncthreads();  // the buffer may be overwritten
            W2_CLK(t5);
            W2_ACC(0, t2 - t1); W2_ACC(1, t4 - t3); W2_ACC(2, (t3 - t2) + (t5 - t4));
        }
#undef W2_ACT
#undef W2_STEP

        // ---- k-ascending fp32 sum (filter_pids.cpp:59-63) and the (score, pid) key, slot-aligned with the survivor list
#pragma unroll
        for (int d = 0; d < 2; d++) {
            if (pid[d] >= 0) {
                float sc = 0.0f;
#pragma unroll
                for (int k = 0; k < 32; k++)
                    if (k < nqc) sc += w2_dec(m[d][k]);
                keys[(size_t)b * key_stride + base + wave * 128 + d * 64 + lane] = flmr_make_key(sc, pid[d]);
            }
        }
    }
#ifdef W2_PROFILE
    if (lane == 0 && prof)
        for (int k = 0; k < 3; k++) atomicAdd(reinterpret_cast<unsigned long long*>(prof) + k, (unsigned long long)prof_acc[k]);
#endif
}

static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// true when the walk is the cheaper form for this launch: it needs the sorted code copy, a chip's worth of items (an item
// occupies one CU for the whole table walk), and a table that is not much longer than the token runs it replaces
bool flmr_stage2_walk_pays(const flmr_index* ix, int nqueries, int max_count) {
    if (!ix->codes_sorted || !ix->centroids_f16_tiled || max_count <= 0) return false;
    const int nchunks = (int)flmr_ceil_div(max_count, W2_DOCS);
    const int64_t nitems = (int64_t)nqueries * nchunks;
    if (nitems * 2 < cu_count()) return false;
    const double tokens = (double)max_count * ((double)ix->N / (double)(ix->num_passages > 0 ? ix->num_passages : 1));
    // measured break-even against the gather form: K = 0.8 x tokens (see the header).  Where the XCD-sliced kernel runs -- and with
    // it the approximate-then-refine selection -- the walk has to beat 2.95 ms instead of 3.8: at 1024 survivors x 128 tokens the
    // walk takes 2.43 ms at K = 32768 and 3.51 ms at K = 65536 (linear in K), the sliced form 2.79 and 2.95: break-even K = 46 k
    const double factor = flmr_stage2_xcd_pays(ix) ? 0.35 : 0.6;
    return (double)ix->K * nchunks <= factor * tokens;
}

int flmr_launch_filter_stage2_walk(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                   int32_t max_count, uint64_t* keys, int64_t key_stride, const _Float16* cen16_tiled,
                                   const _Float16* q_hi, const _Float16* q_lo, const int32_t* codes_sorted, hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    if (f.ncol != 32) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 walk needs one column tile");
    if (!codes_sorted) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 walk needs the sorted code copy");
    const int nchunks = (int)flmr_ceil_div(max_count, W2_DOCS);
    const int64_t nitems = (int64_t)f.nqueries * nchunks;
    const int grid = (int)(nitems < cu_count() ? nitems : cu_count());
    const size_t lds = (size_t)W2_BUF * sizeof(int) + 8192;
    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_stage2_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(filter_stage2_walk_kernel, dim3(grid), dim3(64 * W2_WAVES), lds, st, f, pids, pid_stride, counts, keys,
                       key_stride, cen16_tiled, q_hi, q_lo, codes_sorted, nchunks, (int)nitems
#ifdef W2_PROFILE
                       , nullptr
#endif
    );
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
