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
// 512 centroids: the waves compute the slice's [512 x 32] fp32 score tile with exactly the MFMA sequence of stage 0
// (2 x v_mfma_f32_32x32x16_f16 per 16 dims against q_hi / q_lo, then fma(lo, 2^-11, hi): bitwise the values stage 0
// produces) into one of two 72 KB LDS buffers while the lanes consume the previous slice from the other one: a lane walks
// its passage's codes in ASCENDING order (`codes_sorted`, built once at flmr_index_open) and, for every code inside the
// slice, reads that row from LDS (8 x ds_read_b128 at immediate offsets from one address; rows are 144 bytes apart so that
// random rows spread over the banks) and folds it into its maxima.  max is exact, so the token order does not matter: the maxima -- and the
// k-ascending fp32 sum -- are bit-identical to the gather kernels'.  One barrier per slice.
//
// Bound: the consumer's VALU work (32 v_max per token, issued for the longest per-lane token run of each wave and slice)
// and the MFMA pipe are of the same size; HBM is not involved (the 33.5 MB table is L2 / Infinity-Cache resident, the
// sorted codes of the survivors are 0.5 MB per query).
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

#define W2_SLICE 512                 // centroid rows per LDS buffer
#define W2_STRIDE 36                 // floats per staged score row: 144 B, so that the rows random lanes read start in 16
                                     // different bank groups and every ds_read_b128 stays 16-byte aligned (2 x 72 KB of LDS)
#define W2_WAVES 8
#define W2_DOCS (W2_WAVES * 128)     // passages per work item: two per lane
#define W2_BUF ((W2_SLICE + 1) * W2_STRIDE)   // words per buffer: the score rows + one dummy row (the smallest image)
#define W2_INF 0x7fffffff
#define W2_MAX_DOCLEN 2048           // longest passage the in-LDS code sort handles

// ------------------------------------------------------------------------------------------------
// codes_sorted[off[p] .. off[p+1]) = ascending copy of codes[off[p] .. off[p+1]).  One wave per passage, bitonic in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sort_doc_codes_kernel(const int32_t* __restrict__ codes, const int64_t* __restrict__ offsets,
                                                            int64_t num_passages, int32_t* __restrict__ out) {
    __shared__ int32_t s[W2_MAX_DOCLEN];
    for (int64_t p = blockIdx.x; p < num_passages; p += gridDim.x) {
        const int64_t off = offsets[p];
        const int len = (int)(offsets[p + 1] - off);
        if (len <= 0) continue;
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
        for (int t = threadIdx.x; t < len; t += 64) out[off + t] = s[t];
        __syncthreads();
    }
}

int flmr_build_sorted_codes(flmr_index* ix) {
    ix->codes_sorted = nullptr;
    if (ix->max_doclen > W2_MAX_DOCLEN || ix->N >= 0x7fffffffLL || ix->num_passages <= 0 || ix->N <= 0) return FLMR_OK;
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(&ix->codes_sorted), ((size_t)ix->N + 8) * sizeof(int32_t)));  // + window padding
    FLMR_HIP(hipMemset(ix->codes_sorted + ix->N, 0x7f, 8 * sizeof(int32_t)));
    const int64_t grid = ix->num_passages < 262144 ? ix->num_passages : 262144;
    hipLaunchKernelGGL(sort_doc_codes_kernel, dim3((unsigned)grid), dim3(64), 0, 0, ix->codes, ix->doc_offsets, ix->num_passages,
                       ix->codes_sorted);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipDeviceSynchronize());
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// The walk.  grid = min(#items, #CUs) persistent workgroups of 512 threads, item = (query, block of 1024 survivors);
// dynamic LDS = 2 x W2_SLICE x W2_STRIDE floats (144 KB).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * W2_WAVES, 2) void filter_stage2_walk_kernel(
    flmr_filter_args f, const int32_t* __restrict__ pids, int64_t pid_stride, const int32_t* __restrict__ counts,
    uint64_t* __restrict__ keys, int64_t key_stride, const _Float16* __restrict__ cen16, const _Float16* __restrict__ q_hi,
    const _Float16* __restrict__ q_lo, const int32_t* __restrict__ codes_sorted, int nchunks, int nitems
#ifdef W2_PROFILE
    , long long* prof
#endif
    ) {
#ifdef W2_PROFILE
    long long prof_acc[3] = {0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const buf = reinterpret_cast<int*>(smem);  // [2][W2_SLICE + 1][W2_STRIDE] score images (w2_enc)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int K = f.K;
    const int nslices = (K + W2_SLICE - 1) / W2_SLICE;
    if (threadIdx.x < 2 * W2_STRIDE)  // the dummy row behind each buffer's 512 score rows
        buf[(size_t)(threadIdx.x / W2_STRIDE) * W2_BUF + W2_SLICE * W2_STRIDE + threadIdx.x % W2_STRIDE] = (int)0x80000000;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / nchunks, chunk = item - b * nchunks;
        const int cnt = counts[b];
        const int base = chunk * W2_DOCS;
        if (base >= cnt) continue;  // block-uniform
        const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
        const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;  // <= 32 on this path

        // B operand: this query's fp16 split, lane (i, h) holds dims 64h .. 64h+63 of query token i
        w2h8 bh[8], bl[8];
        {
            const w2h8* ph = reinterpret_cast<const w2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
            const w2h8* pl = reinterpret_cast<const w2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
            for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
        }

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
        w2i4 cw[2], nw[2];
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
            cw[d] = load4(ps[d]);
            cwe[d] = ps[d] + 4u;
            nw[d] = load4(cwe[d]);
            nwe[d] = cwe[d] + 4u;
#pragma unroll
            for (int c = 0; c < 32; c++) m[d][c] = w2_enc(-9999.0f);
        }

        // ---- producer pieces ------------------------------------------------------------------------------------
        w2h8 a[8];  // A operand of the next tile to multiply: lane (i, h) holds dims 64h .. 64h+63 of centroid row i of the tile
        auto load_a = [&](int slice, int tile) {
            int row = slice * W2_SLICE + tile * 32 + i;
            row = row < K ? row : 0;  // rows past the table are never referenced by a code
            const w2h8* pc = reinterpret_cast<const w2h8*>(cen16 + (size_t)row * FLMR_DIM + 64 * h);
#pragma unroll
            for (int s = 0; s < 8; s++) a[s] = pc[s];
        };
        f32x16 ah, al;
        auto mfma_tile = [&]() {
#pragma unroll
            for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bh[s], ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bl[s], al, 0, 0, 0);
            }
        };
        // C/D layout of the 32x32 MFMA: register r of lane (i, h) is row (r&3) + 8*(r>>2) + 4h, column i.  One lane-dependent
        // address, everything else is an immediate offset.
        const int wbase = ((2 * wave) * 32 + 4 * h) * W2_STRIDE + i;
        auto store_tile = [&](int* dst, int tile_rel) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                dst[wbase + (tile_rel * 32 + (r & 3) + 8 * (r >> 2)) * W2_STRIDE] = w2_enc(fmaf(al[r], 1.0f / 2048.0f, ah[r]));
        };
        // slice `s` into buffer `dst`; on return `a` holds tile 2*wave of slice s+1 (prefetched across the consumer phase)
        auto produce = [&](int s, int* dst) {
            mfma_tile();
            load_a(s, 2 * wave + 1);
            store_tile(dst, 0);
            mfma_tile();
            if (s + 1 < nslices) load_a(s + 1, 2 * wave);
            store_tile(dst, 1);
        };

        // ---- consumer: fold the rows of this passage's codes that lie in [sb, se) -----------------------------------
        // No control flow around the maxima: a lane without a token in the slice reads the buffer's dummy row (row W2_SLICE,
        // the smallest image) instead, so the 32 running maxima are updated in place by every trip of the wave-uniform loop
        // (with an `if (active)` around them the compiler carries two copies of all 64 maxima through the loop's phi nodes;
        // with the test at the loop head it splits their live ranges and copies them on every trip -- hence guard + do/while).
        auto consume = [&](const int* src, int sb, int se, w2i4& cwd, const w2i4& nwd, uint32_t& p, uint32_t& ce, const uint32_t ne,
                           const uint32_t e, int* md) {
            if (__ballot(p < ce && p < e && cwd.x < se) != 0ull) do {
                const bool act = p < ce && p < e && cwd.x < se;
                const int* rowp = src + (act ? (cwd.x - sb) : W2_SLICE) * W2_STRIDE;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const w2i4 t = *reinterpret_cast<const w2i4*>(rowp + 4 * j);
                    md[4 * j + 0] = max(md[4 * j + 0], t.x);
                    md[4 * j + 1] = max(md[4 * j + 1], t.y);
                    md[4 * j + 2] = max(md[4 * j + 2], t.z);
                    md[4 * j + 3] = max(md[4 * j + 3], t.w);
                }
                p += act ? 1u : 0u;
                const bool promote = p == ce && ne != ce;   // cw used up and nw present: nw becomes the current window
                cwd.x = promote ? nwd.x : (act ? cwd.y : cwd.x);
                cwd.y = promote ? nwd.y : (act ? cwd.z : cwd.y);
                cwd.z = promote ? nwd.z : (act ? cwd.w : cwd.z);
                cwd.w = promote ? nwd.w : cwd.w;
                ce = promote ? ne : ce;
            } while (__ballot(p < ce && p < e && cwd.x < se) != 0ull);
        };

        // ---- the walk ------------------------------------------------------------------------------------------------
        W2_CLK(t0);
        load_a(0, 2 * wave);
        produce(0, buf);
        __syncthreads();
        for (int s = 0; s < nslices; s++) {
            int* const cur = buf + (size_t)(s & 1) * W2_BUF;
            int* const nxt = buf + (size_t)((s + 1) & 1) * W2_BUF;
            W2_CLK(t1);
            // request the window behind nw (position nwe) for both passages: it lands during the MFMA work below
            const uint32_t sta0 = nwe[0], sta1 = nwe[1];
            const w2i4 st0 = load4(sta0), st1 = load4(sta1);
            if (s + 1 < nslices) produce(s + 1, nxt);
            W2_CLK(t2);
            const int sb = s * W2_SLICE, se = sb + W2_SLICE;
            while (true) {
                consume(cur, sb, se, cw[0], nw[0], ps[0], cwe[0], nwe[0], end[0], m[0]);
                consume(cur, sb, se, cw[1], nw[1], ps[1], cwe[1], nwe[1], end[1], m[1]);
                // a passage with more than 4 .. 8 codes inside one slice runs out of buffered codes (rare): fetch its next
                // window now (exposed latency) and go round again
                const bool starved0 = ps[0] == cwe[0] && ps[0] < end[0], starved1 = ps[1] == cwe[1] && ps[1] < end[1];
                if (__ballot(starved0 || starved1) == 0ull) break;
                const w2i4 x0 = load4(cwe[0]), x1 = load4(cwe[1]);
                if (starved0) { cw[0] = x0; cwe[0] += 4u; nwe[0] = cwe[0]; }
                if (starved1) { cw[1] = x1; cwe[1] += 4u; nwe[1] = cwe[1]; }
            }
            // nw was promoted during this slice -> the staged window (it starts where nw ended) is the new nw
            {
                const bool take0 = nwe[0] == cwe[0] && sta0 == cwe[0], take1 = nwe[1] == cwe[1] && sta1 == cwe[1];
                nw[0].x = take0 ? st0.x : nw[0].x; nw[0].y = take0 ? st0.y : nw[0].y;
                nw[0].z = take0 ? st0.z : nw[0].z; nw[0].w = take0 ? st0.w : nw[0].w;
                nw[1].x = take1 ? st1.x : nw[1].x; nw[1].y = take1 ? st1.y : nw[1].y;
                nw[1].z = take1 ? st1.z : nw[1].z; nw[1].w = take1 ? st1.w : nw[1].w;
                nwe[0] += take0 ? 4u : 0u;
                nwe[1] += take1 ? 4u : 0u;
            }
            W2_CLK(t3);
            __syncthreads();  // `nxt` is complete, `cur` may be overwritten
            W2_CLK(t4);
            W2_ACC(0, t2 - t1); W2_ACC(1, t3 - t2); W2_ACC(2, t4 - t3);
        }

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
    if (!ix->codes_sorted || max_count <= 0) return false;
    const int nchunks = (int)flmr_ceil_div(max_count, W2_DOCS);
    const int64_t nitems = (int64_t)nqueries * nchunks;
    if (nitems * 2 < cu_count()) return false;
    const double tokens = (double)max_count * ((double)ix->N / (double)(ix->num_passages > 0 ? ix->num_passages : 1));
    return (double)ix->K * nchunks <= 1.5 * tokens;
}

int flmr_launch_filter_stage2_walk(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                   int32_t max_count, uint64_t* keys, int64_t key_stride, const _Float16* cen16,
                                   const _Float16* q_hi, const _Float16* q_lo, const int32_t* codes_sorted, hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    if (f.ncol != 32) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 walk needs one column tile");
    if (!codes_sorted) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 walk needs the sorted code copy");
    const int nchunks = (int)flmr_ceil_div(max_count, W2_DOCS);
    const int64_t nitems = (int64_t)f.nqueries * nchunks;
    const int grid = (int)(nitems < cu_count() ? nitems : cu_count());
    const size_t lds = (size_t)2 * W2_BUF * sizeof(float);
    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_stage2_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(filter_stage2_walk_kernel, dim3(grid), dim3(64 * W2_WAVES), lds, st, f, pids, pid_stride, counts, keys,
                       key_stride, cen16, q_hi, q_lo, codes_sorted, nchunks, (int)nitems
#ifdef W2_PROFILE
                       , nullptr
#endif
    );
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
