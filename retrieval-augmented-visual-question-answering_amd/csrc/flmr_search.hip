// Batched search orchestration: workspace planning + the S0..S4 launch sequence on one HIP stream.
// Replaces Searcher._search_all_Q / dense_search / IndexScorer.rank (TPC/searcher.py:73-132,
// TPC/search/index_storage.py:67-182) for a whole batch of queries at once.
#include <new>

#include "flmr_common.h"

static const char* kStageNames[FLMR_NUM_STAGES] = {
    "s0_centroid_scores", "s0_select_cells", "s0_candidates", "s1_hitmap", "s1_filter", "s1_select", "s2_filter_sort",
    "s3_maxsim", "s4_topk"};

extern "C" const char* flmr_stage_name(int32_t s) { return (s >= 0 && s < FLMR_NUM_STAGES) ? kStageNames[s] : "?"; }

#define FLMR_PROF_RING 8

struct flmr_searcher {
    const flmr_index* ix;
    int32_t max_queries, max_nq;
    flmr_search_params_t maxp;
    int32_t ncol_max, idx_words, nblk, max_cells, nc_bucket;
    int64_t bitmap_words, cand_cap, bytes;
    float* cs; uint32_t* idx_bits; float* part_val; int32_t* part_idx; int32_t* cells; int32_t* ncell;
    uint32_t* bitmap; int32_t* cand; int32_t* cand_count; uint64_t* keys1; int32_t* s1_pids; int32_t* s1_count;
    uint64_t* keys2; int32_t* s2_pids; int32_t* s2_count; uint64_t* keys3; float* doc_scores;
    int32_t* overflow;          // = status: device flags [0] candidate capacity exceeded, [1] q_lens outside [0, nq] ([2], [3] unused)
    int32_t* row_ovf;           // [max_queries] 1 = more surviving centroids than compact score rows (row_cap): the query's stage 1 is
                                // recomputed from the fp16 centroids inside the same batch (no error, no host round trip)
    int32_t* status_host;       // pinned copy of the flags, refreshed asynchronously after every batch
    float* rows;                // compact score rows [max_queries, row_cap, 32]: the sparse path's whole "score table"
    uint32_t* idx_prefix;       // [max_queries, idx_words] ranks of the surviving centroids (qualifying_kernel)
    int32_t row_cap;            // score rows per query: min(K, FLMR_ROW_CAP (default 16384))
    hipEvent_t status_ev; bool status_pending;
    int32_t* q_lens_ws;         // [max_queries] query lengths clamped to [0, nq]: what every kernel reads
    flmr_options opt;           // variant switches, snapshot taken at flmr_searcher_create
    _Float16* q_hi; _Float16* q_lo;
    float* q_err;               // [max_queries, ncol_max] per-column error bound of the hi-only stage-0 scores
    float* q_err_sum;           // [max_queries] error bound of a passage's hi-only stage-2 score
    int32_t* s2_band; int32_t* s2_band_count; int32_t* s2_need; int32_t* s2_def; uint64_t* keys2b;   // stage 2, approximate-then-refine
    uint32_t* hit_bits; int32_t* hit_valid; int32_t* key_count; int32_t* chunk_hits;
    int32_t* s1_slot; int32_t* s2_slot;   // sharded protocol: position of each local survivor / finalist in the global list
    float* s2_part;             // XCD-sliced stage 2: per (query, slice, survivor) column maxima (NULL when the index has no split table)
    _Float16* q3_hi; _Float16* q3_lo;
    uint2* s3_desc; int64_t s3_desc_stride; int32_t* s3_wbeg; int32_t s3_wcap;   // planned-tile S3 (NULL: the passage-walking kernel)
    float* s3_colmax; int64_t s3_colmax_cap;   // long queries: per (query, finalist, column) maxima of the query-stationary S3 kernel
    int32_t* qual; int32_t* nqual; int32_t* chunk_cnt; uint8_t* cand_hit; int32_t qmax;
    // stage 1 for dense survivor sets (flmr_stage1_dense.hip): per query which form, the band of an image query, its error bound,
    // how many keys the selection reads, and which queries are left to the round-5 scan
    int32_t* s1d_any; int32_t* s1d_mode; int32_t* s1d_band; int32_t* s1d_band_count; float* s1d_err; int32_t* s1d_in_count; int32_t* s1d_scan_skip;
    int32_t s1d_img_rows;       // score-row images the LDS form holds (0: no dense form on this index)
    int32_t* cand_fast;         // flmr_cand_args::fast_state ([FLMR_FAST_HDR + 2 * max_queries]; the header words live as long as the searcher)
    // last call (for taps)
    int32_t last_nqueries, last_ncol, last_ndocs, last_full_table;
    flmr_cand_args last_ca{};   // candidate-stage arguments of the last batch (lazy FLMR_TAP_CANDIDATES in scatter mode)
    bool last_scatter = false;
    bool last_dense = false;    // the last batch ran the dense stage-1 forms for the queries the list-scatter forms left
    bool last_hi_first = false; // the last batch's stage 0 wrote q_err / q_err_sum (FLMR_TAP_Q_ERR*)
    hipStream_t last_stream;
    bool profiling;
    bool full_table;  // keep the whole centroid-score table (needed by the CENTROID_SCORES tap / retrieve())
    int32_t numerics; // FLMR_NUMERICS_CPU (default) | FLMR_NUMERICS_GPU_FP16
    // stage timing: a ring of event sets, one per profiled call, folded into acc_ms when read (or when the ring wraps)
    hipEvent_t ev[FLMR_PROF_RING][FLMR_NUM_STAGES + 1];
    int ev_first = 0, ev_pending = 0;   // oldest unread set, number of unread sets
    float acc_ms[FLMR_NUM_STAGES] = {};
    bool have_ms;
};

static int nc_bucket_of(int ncells) { return ncells <= 1 ? 1 : ncells <= 2 ? 2 : ncells <= 4 ? 4 : 8; }

template <typename T>
static int ws_alloc(flmr_searcher* s, T** p, size_t count) {
    const size_t bytes = (count ? count : 1) * sizeof(T);
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(p), bytes));
    s->bytes += (int64_t)bytes;
    return FLMR_OK;
}

static int check_params(const flmr_search_params_t* p) {
    if (!p) FLMR_FAIL(FLMR_ERR_INVALID, "params is NULL");
    if (p->ncells < 1 || p->ncells > FLMR_MAX_NCELLS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "ncells=%d (1..%d)", p->ncells, FLMR_MAX_NCELLS);
    if (p->ndocs < 4 || p->ndocs > FLMR_MAX_NDOCS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "ndocs=%d (4..%d)", p->ndocs, FLMR_MAX_NDOCS);
    if (p->nq_cand < 1 || p->nq_cand > FLMR_MAX_NQ_CAND) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq_cand=%d (1..%d)", p->nq_cand, FLMR_MAX_NQ_CAND);
    if (p->k < 1) FLMR_FAIL(FLMR_ERR_INVALID, "k=%d", p->k);
    return FLMR_OK;
}

extern "C" int flmr_searcher_create(const flmr_index_t* ix, int32_t max_queries, int32_t max_nq,
                                    const flmr_search_params_t* maxp, flmr_searcher_t** out) {
    if (!ix || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = check_params(maxp);
    if (rc) return rc;
    if (max_queries < 1 || max_nq < 1) FLMR_FAIL(FLMR_ERR_INVALID, "max_queries/max_nq must be >= 1");
    flmr_searcher* s = new (std::nothrow) flmr_searcher();
    if (!s) FLMR_FAIL(FLMR_ERR_NOMEM, "host allocation failed");
    memset(s, 0, sizeof(*s));
    s->ix = ix; s->max_queries = max_queries; s->max_nq = max_nq; s->maxp = *maxp;
    s->opt = flmr_process_options();
    const int nqc = maxp->nq_cand < max_nq ? maxp->nq_cand : max_nq;
    s->ncol_max = (int32_t)flmr_round_up(nqc, 32);
    s->idx_words = (int32_t)flmr_ceil_div(ix->K, 32);
    s->nblk = (int32_t)flmr_ceil_div(ix->K, 64);  // upper bound on partial-list blocks (64 rows each on the fp16 kernels, 128 on the fp32 ones)
    s->max_cells = nqc * maxp->ncells;
    s->nc_bucket = nc_bucket_of(maxp->ncells);
    s->bitmap_words = flmr_ceil_div(ix->num_passages > 0 ? ix->num_passages : 1, 32);
    const int mc = s->max_cells < ix->K ? s->max_cells : ix->K;
    s->cand_cap = ix->ivf_len_prefix[mc] < ix->num_passages ? ix->ivf_len_prefix[mc] : ix->num_passages;
    if (s->cand_cap < 1) s->cand_cap = 1;
    const size_t B = (size_t)max_queries;
    const int nd = maxp->ndocs, nd4 = maxp->ndocs / 4;
#define WS(ptr, count)                          \
    do {                                        \
        rc = ws_alloc(s, &s->ptr, (count));     \
        if (rc) { flmr_searcher_destroy(s); return rc; } \
    } while (0)
    // The full K x ncol score table per query (16.8 MB at K = 131072) is allocated on first use by a batch that needs it
    // (full-table mode: taps / retrieve(), centroids that are not fp16-exact, nq_cand > 32); the default path keeps only the
    // rows of each query's surviving centroids, at most row_cap of them (2 MB per query at the default).
    s->cs = nullptr;
    {
        const char* rc_opt = s->opt.has(FLMR_OPT_ROW_CAP) ? s->opt.v[FLMR_OPT_ROW_CAP] : nullptr;
        long cap = rc_opt ? atol(rc_opt) : 16384;
        if (cap < 64) cap = 64;
        if (cap > 65535) cap = 65535;   // (the code-scanning stage 1 keeps 16-bit ranks in LDS)
        s->row_cap = (int32_t)(cap < ix->K ? cap : ix->K);
    }
    WS(rows, B * (size_t)s->row_cap * 32);
    WS(idx_prefix, B * (size_t)s->idx_words);
    WS(idx_bits, B * (size_t)s->idx_words);
    WS(part_val, B * (size_t)s->nblk * s->ncol_max * s->nc_bucket);
    WS(part_idx, B * (size_t)s->nblk * s->ncol_max * s->nc_bucket);
    WS(cells, B * (size_t)s->max_cells);
    WS(ncell, B);
    WS(bitmap, B * (size_t)s->bitmap_words);
    WS(cand, B * (size_t)s->cand_cap);
    WS(cand_count, B);
    WS(keys1, B * (size_t)s->cand_cap);
    WS(s1_pids, B * (size_t)nd);
    WS(s1_count, B);
    WS(keys2, B * (size_t)nd);
    WS(s2_pids, B * (size_t)nd4);
    WS(s2_count, B);
    WS(keys3, B * (size_t)nd4);
    WS(doc_scores, B * (size_t)nd4);
    WS(overflow, 4);
    WS(q_lens_ws, B);
    WS(q_hi, B * (size_t)s->ncol_max * FLMR_DIM);
    WS(q_lo, B * (size_t)s->ncol_max * FLMR_DIM);
    WS(q_err, B * (size_t)s->ncol_max);
    WS(q_err_sum, B);
    WS(s2_band, B * (size_t)nd);
    WS(s2_band_count, B);
    WS(s2_need, B);
    WS(s2_def, B);
    WS(keys2b, B * (size_t)nd);
    WS(hit_bits, B * (size_t)s->bitmap_words);
    WS(hit_valid, B);
    WS(key_count, B);
    WS(s1_slot, B * (size_t)nd);
    WS(s2_slot, B * (size_t)nd4);
    if (ix->doc_splits && s->ncol_max == 32 && (flmr_stage2_xcd_pays(ix) || s->opt.is(FLMR_OPT_S2_IMPL, "xcd"))) {
        // optional: without it (1 KB per query and survivor; e.g. out of memory) stage 2 runs in its gather form
        const size_t bytes = flmr_stage2_xcd_part_floats(ix, (int64_t)B, nd) * sizeof(float);
        if (hipMalloc(reinterpret_cast<void**>(&s->s2_part), bytes) == hipSuccess) s->bytes += (int64_t)bytes;
        else { s->s2_part = nullptr; (void)hipGetLastError(); }
    }
    s->qmax = s->row_cap;
    WS(qual, B * (size_t)s->qmax);
    WS(nqual, B);
    WS(chunk_cnt, B * (size_t)ix->nchunks);
    WS(chunk_hits, B * (size_t)ix->nchunks);
    WS(cand_hit, B * (size_t)s->cand_cap);
    // dense stage 1: needs the sorted code copy (whole 16-byte pieces are read: its padding) and 32-bit token offsets
    s->s1d_img_rows = (ix->codes_sorted && s->ncol_max == 32 && !s->opt.is(FLMR_OPT_S1_IMPL, "scan")) ? flmr_s1_dense_image_rows(max_queries, s->idx_words, ix->mean_ulen) : 0;
    WS(s1d_any, 16); WS(s1d_mode, B); WS(s1d_band_count, B); WS(s1d_err, B); WS(s1d_in_count, B); WS(s1d_scan_skip, B); WS(row_ovf, B);
    if (s->s1d_img_rows > 0) WS(s1d_band, B * (size_t)s->cand_cap);
    WS(q3_hi, B * (size_t)flmr_round_up(max_nq, 32) * FLMR_DIM);
    WS(q3_lo, B * (size_t)flmr_round_up(max_nq, 32) * FLMR_DIM);
#undef WS
    {   // optional workspace of the planned-tile S3 kernel: one 8-byte descriptor per 32-token tile of a query's finalists
        s->s3_desc_stride = (int64_t)nd4 * ((ix->max_doclen + 31) / 32);
        s->s3_wcap = nd4 + 8;
        const size_t db = B * (size_t)s->s3_desc_stride * sizeof(uint2), wb = B * (size_t)s->s3_wcap * sizeof(int32_t);
        if (s->s3_desc_stride > 0 && db <= ((size_t)1 << 30) && hipMalloc(reinterpret_cast<void**>(&s->s3_desc), db) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&s->s3_wbeg), wb) == hipSuccess) {
            s->bytes += (int64_t)(db + wb);
        } else {
            (void)hipGetLastError();
            (void)hipFree(s->s3_desc); (void)hipFree(s->s3_wbeg);
            s->s3_desc = nullptr; s->s3_wbeg = nullptr;
        }
        if (s->s3_desc && max_nq > 32) {
            s->s3_colmax_cap = (int64_t)B * nd4 * flmr_round_up(max_nq, 32) + (int64_t)B;   // (+ the per-query output scales of the query-stationary kernel)
            if (s->s3_colmax_cap * (int64_t)sizeof(float) <= ((int64_t)2 << 30) &&
                hipMalloc(reinterpret_cast<void**>(&s->s3_colmax), (size_t)s->s3_colmax_cap * sizeof(float)) == hipSuccess) {
                s->bytes += s->s3_colmax_cap * (int64_t)sizeof(float);
            } else {
                (void)hipGetLastError();
                s->s3_colmax = nullptr; s->s3_colmax_cap = 0;
            }
        }
    }
    {
        int32_t* p = nullptr;
        rc = ws_alloc(s, &p, 2 * B + FLMR_FAST_HDR);
        if (rc) { flmr_searcher_destroy(s); return rc; }
        s->cand_fast = p;
        FLMR_HIP(hipMemset(s->cand_fast, 0, (2 * B + FLMR_FAST_HDR) * sizeof(int32_t)));
    }
    FLMR_HIP(hipMemset(s->overflow, 0, 4 * sizeof(int32_t)));
    FLMR_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->status_host), 4 * sizeof(int32_t), hipHostMallocDefault));
    s->status_host[0] = s->status_host[1] = s->status_host[2] = s->status_host[3] = 0;
    FLMR_HIP(hipEventCreateWithFlags(&s->status_ev, hipEventDisableTiming));
    for (int r = 0; r < FLMR_PROF_RING; r++)
        for (int i = 0; i <= FLMR_NUM_STAGES; i++) FLMR_HIP(hipEventCreate(&s->ev[r][i]));
    *out = s;
    return FLMR_OK;
}

extern "C" int flmr_searcher_destroy(flmr_searcher_t* s) {
    if (!s) return FLMR_OK;
    void* ptrs[] = {s->cs, s->rows, s->idx_prefix, s->idx_bits, s->part_val, s->part_idx, s->cells, s->ncell, s->bitmap, s->cand, s->cand_count,
                    s->keys1, s->s1_pids, s->s1_count, s->keys2, s->s2_pids, s->s2_count, s->keys3, s->doc_scores,
                    s->overflow, s->q_lens_ws, s->q_hi, s->q_lo, s->q_err, s->q_err_sum, s->s2_band, s->s2_band_count, s->s2_need, s->s2_def, s->keys2b, s->hit_bits, s->hit_valid, s->q3_hi, s->q3_lo, s->qual, s->nqual, s->chunk_cnt, s->chunk_hits, s->cand_hit, s->key_count, s->s1_slot, s->s2_slot, s->s2_part, s->s3_desc, s->s3_wbeg, s->s3_colmax, s->cand_fast, s->s1d_mode, s->s1d_band, s->s1d_band_count, s->s1d_err, s->s1d_in_count, s->s1d_scan_skip, s->row_ovf, s->s1d_any};
    for (void* p : ptrs) (void)hipFree(p);
    if (s->status_host) (void)hipHostFree(s->status_host);
    if (s->status_ev) (void)hipEventDestroy(s->status_ev);
    for (int r = 0; r < FLMR_PROF_RING; r++)
        for (int i = 0; i <= FLMR_NUM_STAGES; i++)
            if (s->ev[r][i]) (void)hipEventDestroy(s->ev[r][i]);
    delete s;
    return FLMR_OK;
}

extern "C" int flmr_searcher_workspace_bytes(const flmr_searcher_t* s, int64_t* bytes) {
    if (!s || !bytes) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    *bytes = s->bytes;
    return FLMR_OK;
}

// waits for the oldest unread event set and adds its stage times to acc_ms
static int fold_oldest(flmr_searcher* s) {
    hipEvent_t* e = s->ev[s->ev_first];
    FLMR_HIP(hipEventSynchronize(e[FLMR_NUM_STAGES]));
    for (int i = 0; i < FLMR_NUM_STAGES; i++) {
        float ms = 0.f;
        FLMR_HIP(hipEventElapsedTime(&ms, e[i], e[i + 1]));
        s->acc_ms[i] += ms;
    }
    s->ev_first = (s->ev_first + 1) % FLMR_PROF_RING;
    s->ev_pending--;
    return FLMR_OK;
}

extern "C" int flmr_searcher_set_profiling(flmr_searcher_t* s, int32_t enable) {
    if (!s) FLMR_FAIL(FLMR_ERR_INVALID, "NULL searcher");
    if (s->profiling != (enable != 0)) {   // a change of mode starts a fresh accumulation
        s->ev_pending = 0;
        s->have_ms = false;
        for (float& v : s->acc_ms) v = 0.f;
    }
    s->profiling = enable != 0;
    return FLMR_OK;
}

extern "C" int flmr_searcher_set_full_table(flmr_searcher_t* s, int32_t enable) {
    if (!s) FLMR_FAIL(FLMR_ERR_INVALID, "NULL searcher");
    s->full_table = enable != 0;
    return FLMR_OK;
}

// The smallest fp32 x with half(x) >= half(thr): the reference's CUDA path compares an fp16 tensor with the Python scalar
// (index_storage.py:116 on `centroid_scores.cuda()` half values: the scalar is taken in the tensor's dtype), so the
// predicate on the fp32-accumulated score s is  half(s) >= half(thr)  <=>  s >= that x (rounding is monotone).
static float f16_threshold(float thr) {
    const _Float16 h = (_Float16)thr;
    if (!(h == h)) return thr;
    uint16_t hb;
    memcpy(&hb, &h, 2);
    if ((hb & 0x7fffu) == 0) {   // +-0: anything that rounds to (+-)0 or above
        hb = 0x8001u;            // largest negative subnormal half = predecessor of zero
    } else if (hb & 0x8000u) {
        hb++;                    // negative: predecessor = larger magnitude
    } else {
        hb--;                    // positive: predecessor = smaller magnitude
    }
    _Float16 pred;
    memcpy(&pred, &hb, 2);
    const float mid = 0.5f * ((float)pred + (float)h);   // exact in fp32 (two adjacent halves)
    // the midpoint rounds to the neighbour with the even mantissa
    uint16_t hbits;
    memcpy(&hbits, &h, 2);
    const bool h_even = (hbits & 1u) == 0;
    return h_even ? mid : nextafterf(mid, INFINITY);
}

extern "C" int flmr_searcher_set_numerics(flmr_searcher_t* s, int32_t mode) {
    if (!s) FLMR_FAIL(FLMR_ERR_INVALID, "NULL searcher");
    if (mode != FLMR_NUMERICS_CPU && mode != FLMR_NUMERICS_GPU_FP16) FLMR_FAIL(FLMR_ERR_INVALID, "unknown numerics mode %d", mode);
    if (mode == FLMR_NUMERICS_GPU_FP16 && !(s->ix->centroids_f16_exact && s->ix->centroids_f16 && s->ix->K % 64 == 0))
        FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "the fp16 numerics mode needs fp16-representable centroids (every reference-format index) and K %% 64 == 0");
    s->numerics = mode;
    return FLMR_OK;
}

extern "C" int flmr_searcher_stage_ms(flmr_searcher_t* s, float* ms) {
    if (!s || !ms) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (!s->have_ms) FLMR_FAIL(FLMR_ERR_INVALID, "no profiled flmr_search_batch call since the last read");
    while (s->ev_pending) {
        const int rc = fold_oldest(s);
        if (rc) return rc;
    }
    for (int i = 0; i < FLMR_NUM_STAGES; i++) {
        ms[i] = s->acc_ms[i];
        s->acc_ms[i] = 0.f;
    }
    s->have_ms = false;
    return FLMR_OK;
}

// ---- one batch = a context (validated parameters + kernel argument blocks) run through stage helpers -----------------
struct run_ctx {
    flmr_opt_scope scope;   // every launch of this call sees the searcher's switch snapshot
    explicit run_ctx(flmr_searcher* s_) : scope(s_ ? &s_->opt : nullptr) {}
    flmr_searcher* s;
    const float* Q;
    const int32_t* q_lens;
    int32_t nqueries, nq, nqc, ncol;
    flmr_search_params_t p;
    bool sparse;
    hipStream_t st;
    flmr_s0_args a0{};
    flmr_filter_args f{};
    int stage;  // profiling event cursor
    int ev_set; // event set of this call
};

#define RUN(x)          \
    do {                \
        int rc__ = (x); \
        if (rc__) return rc__; \
    } while (0)

static int mark(run_ctx& c) {
    if (c.s->profiling && c.stage <= FLMR_NUM_STAGES) FLMR_HIP(hipEventRecord(c.s->ev[c.ev_set][c.stage], c.st));
    c.stage++;
    return FLMR_OK;
}

// Device-side error flags (candidate capacity, q_lens range) are copied to pinned host memory after every batch without a
// sync; the next call on the searcher (or flmr_searcher_check, which waits) reports a raised flag and clears it.
static int poll_status(flmr_searcher* s, bool wait) {
    if (!s->status_pending) return FLMR_OK;
    if (wait) {
        FLMR_HIP(hipEventSynchronize(s->status_ev));
    } else {
        const hipError_t e = hipEventQuery(s->status_ev);
        if (e == hipErrorNotReady) return FLMR_OK;
        FLMR_HIP(e);
    }
    s->status_pending = false;
    const int32_t ovf = s->status_host[0], bad = s->status_host[1];
    if (ovf || bad) {
        s->status_host[0] = s->status_host[1] = s->status_host[2] = 0;
        // cleared IN ORDER on the searcher's stream: a synchronous memset on the null stream is not ordered against a later
        // batch already running on a non-blocking stream and could wipe that batch's flag
        FLMR_HIP(hipMemsetAsync(s->overflow, 0, 4 * sizeof(int32_t), s->last_stream));
        if (ovf)
            FLMR_FAIL(FLMR_ERR_CAPACITY, "an earlier batch produced more candidates than the workspace bound (cand_cap=%lld): its "
                      "candidate lists were truncated and its results are not reliable (is the IVF consistent with the codes?)",
                      (long long)s->cand_cap);
        FLMR_FAIL(FLMR_ERR_INVALID, "an earlier batch passed q_lens outside [0, nq]; they were clamped");
    }
    return FLMR_OK;
}

static int push_status(run_ctx& c) {
    flmr_searcher* s = c.s;
    FLMR_HIP(hipMemcpyAsync(s->status_host, s->overflow, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c.st));
    FLMR_HIP(hipEventRecord(s->status_ev, c.st));
    s->status_pending = true;
    return FLMR_OK;
}

extern "C" int flmr_searcher_status_async(flmr_searcher_t* s, int32_t* host_flags, flmr_stream_t stream) {
    if (!s || !host_flags) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    FLMR_HIP(hipMemcpyAsync(host_flags, s->overflow, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
    return FLMR_OK;
}

extern "C" int flmr_searcher_check(flmr_searcher_t* s) {
    if (!s) FLMR_FAIL(FLMR_ERR_INVALID, "NULL searcher");
    return poll_status(s, true);
}

__global__ void sanitize_q_lens_kernel(const int32_t* in, int32_t n, int32_t nq, int32_t* out, int32_t* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = in[i];
    const int32_t c = v < 0 ? 0 : (v > nq ? nq : v);
    out[i] = c;
    if (c != v) atomicExch(flag, 1);
}

// the sparse score table + recomputing stage 2 (and with them the query-split stage 0) need the fp16-split S0 path and
// a single column tile
static bool sparse_path(const flmr_searcher* s, int ncol) {
    const flmr_index* ix = s->ix;
    const flmr_options& o = s->opt;
    const bool f16_path = ix->centroids_f16_exact && ix->centroids_f16 && (ix->K % 64 == 0) &&
                          !(o.has(FLMR_OPT_S0_IMPL) && !o.is(FLMR_OPT_S0_IMPL, "f16") && !o.is(FLMR_OPT_S0_IMPL, "f16rs") && !o.is(FLMR_OPT_S0_IMPL, "qs1"));
    return f16_path && ncol == 32 && !s->full_table && !o.has(FLMR_OPT_FULL_TABLE);
}

extern "C" int flmr_searcher_probe_supported(const flmr_searcher_t* s, int32_t nq, const flmr_search_params_t* p,
                                             int32_t* supported) {
    if (!s || !p || !supported) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    const int nqc = p->nq_cand < nq ? p->nq_cand : nq;
    *supported = sparse_path(s, (int)flmr_round_up(nqc, 32)) ? 1 : 0;
    return FLMR_OK;
}

static inline bool f16num_early(const flmr_searcher* s) { return s->numerics == FLMR_NUMERICS_GPU_FP16; }

static int prepare_ctx(run_ctx& c, flmr_searcher* s, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                       const flmr_search_params_t* p, flmr_stream_t stream) {
    if (!s || !Q) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    RUN(check_params(p));
    RUN(poll_status(s, false));
    if (nqueries < 1 || nqueries > s->max_queries) FLMR_FAIL(FLMR_ERR_CAPACITY, "nqueries=%d > max_queries=%d", nqueries, s->max_queries);
    if (nq < 1 || nq > s->max_nq) FLMR_FAIL(FLMR_ERR_CAPACITY, "nq=%d > max_nq=%d", nq, s->max_nq);
    const int nqc = p->nq_cand < nq ? p->nq_cand : nq;
    const int ncol = (int)flmr_round_up(nqc, 32);
    if (ncol > s->ncol_max || nqc * p->ncells > s->max_cells || nc_bucket_of(p->ncells) > s->nc_bucket || p->ndocs > s->maxp.ndocs)
        FLMR_FAIL(FLMR_ERR_CAPACITY, "params exceed the bounds given at flmr_searcher_create");
    const flmr_index* ix = s->ix;
    c.s = s; c.Q = Q; c.q_lens = q_lens; c.nqueries = nqueries; c.nq = nq; c.nqc = nqc; c.ncol = ncol; c.p = *p;
    c.st = reinterpret_cast<hipStream_t>(stream);
    c.stage = 0;
    if (s->profiling && s->ev_pending == FLMR_PROF_RING) RUN(fold_oldest(s));   // the ring is full: make room (waits)
    c.ev_set = (s->ev_first + s->ev_pending) % FLMR_PROF_RING;
    if (q_lens) {  // kernels only ever see lengths inside [0, nq]; a violation is reported by the next status poll
        hipLaunchKernelGGL(sanitize_q_lens_kernel, dim3((nqueries + 255) / 256), dim3(256), 0, c.st, q_lens, nqueries, nq,
                           s->q_lens_ws, s->overflow + 1);
        FLMR_LAUNCH_CHECK();
        c.q_lens = q_lens = s->q_lens_ws;
    }
    flmr_s0_args& a0 = c.a0;
    a0.centroids = ix->centroids; a0.Q = Q; a0.q_lens = q_lens;
    a0.K = ix->K; a0.nqueries = nqueries; a0.nq = nq; a0.nq_cand = nqc; a0.ncol = ncol; a0.ncells = p->ncells;
    const bool f16num = s->numerics == FLMR_NUMERICS_GPU_FP16;
    a0.thr = f16num ? f16_threshold(p->centroid_score_threshold) : p->centroid_score_threshold;
    a0.q_hi_only = f16num ? 1 : 0;
    c.sparse = sparse_path(s, ncol);
    if (!c.sparse && !s->cs) {   // first batch that needs the whole K x ncol table per query: allocate it now (synchronous, once)
        const size_t bytes = (size_t)s->max_queries * (size_t)ix->K * s->ncol_max * sizeof(float);
        if (hipMalloc(reinterpret_cast<void**>(&s->cs), bytes) != hipSuccess) {
            (void)hipGetLastError();
            s->cs = nullptr;
            FLMR_FAIL(FLMR_ERR_NOMEM, "the full centroid-score table needs %.1f GB for %d queries (full-table mode: taps, centroids that are "
                      "not fp16-representable, nq_cand > 32): create the searcher for fewer queries", bytes / 1e9, s->max_queries);
        }
        s->bytes += (int64_t)bytes;
    }
    if (s->opt.has(FLMR_OPT_POISON)) {   // test runs: nothing may be read from an earlier batch's workspace
        FLMR_HIP(hipMemsetAsync(s->rows, 0xFF, (size_t)nqueries * s->row_cap * 32 * sizeof(float), c.st));
        FLMR_HIP(hipMemsetAsync(s->idx_prefix, 0xFF, (size_t)nqueries * s->idx_words * sizeof(uint32_t), c.st));
        FLMR_HIP(hipMemsetAsync(s->qual, 0xFF, (size_t)nqueries * s->qmax * sizeof(int32_t), c.st));
        FLMR_HIP(hipMemsetAsync(s->keys1, 0xFF, (size_t)nqueries * s->cand_cap * sizeof(uint64_t), c.st));
        FLMR_HIP(hipMemsetAsync(s->keys2, 0xFF, (size_t)nqueries * s->maxp.ndocs * sizeof(uint64_t), c.st));
        FLMR_HIP(hipMemsetAsync(s->keys3, 0xFF, (size_t)nqueries * (s->maxp.ndocs / 4) * sizeof(uint64_t), c.st));
    }
    a0.cs = c.sparse ? nullptr : s->cs; a0.idx_bits = s->idx_bits; a0.idx_words = s->idx_words;
    a0.part_val = s->part_val; a0.part_idx = s->part_idx; a0.nblk = s->nblk;
    a0.cells = s->cells; a0.ncell = s->ncell; a0.max_cells = s->max_cells;
    a0.q_hi = s->q_hi; a0.q_lo = s->q_lo; a0.centroids_f16_exact = ix->centroids_f16_exact;
    // "hi first" stage 0: on unless the fp16 numerics mode makes it moot (q_lo = 0) or FLMR_S0_IMPL asks for another kernel
    a0.q_err_buf = f16num_early(s) ? nullptr : s->q_err;
    // (any batch size since the cell selection reads the coarse level of maxima this path leaves: a single-query call takes 0.309 ms
    // with it, 0.322 ms with both products everywhere; before the coarse level the selection's serial pass over a near-tie column
    // -- most queries have one -- made the shortcut a loss under 16 queries)
#ifndef S0_HIFIRST_MIN_QUERIES
#define S0_HIFIRST_MIN_QUERIES 1
#endif
    a0.q_err = (!f16num_early(s) && (!s->opt.has(FLMR_OPT_S0_IMPL) || s->opt.is(FLMR_OPT_S0_IMPL, "qs1")) && nqueries >= S0_HIFIRST_MIN_QUERIES) ? s->q_err : nullptr;
    a0.q_err_sum = s->q_err_sum;
    a0.cen_norm_max = ix->cen_norm_max;
    a0.centroids_f16 = ix->centroids_f16;
    a0.part_rows = 0;
    a0.full_table = c.sparse ? 0 : 1;
    flmr_filter_args& f = c.f;
    f.cs = c.sparse ? s->rows : s->cs; f.cs_query_stride = c.sparse ? (int64_t)s->row_cap * 32 : (int64_t)ix->K * ncol;
    f.cs_compact = c.sparse ? 1 : 0; f.row_cap = s->row_cap; f.idx_prefix = s->idx_prefix;
    f.K = ix->K; f.ncol = ncol; f.nq_cand = nqc;
    f.nqueries = nqueries; f.q_lens = q_lens; f.codes = ix->codes; f.doclens = nullptr; f.offsets = ix->doc_offsets;
    f.f16_round = f16num ? 1 : 0;
    f.ulen_sorted = ix->doc_ulen;
    s->last_nqueries = nqueries; s->last_ncol = ncol; s->last_ndocs = p->ndocs; s->last_stream = c.st;
    s->last_full_table = a0.full_table;
    s->last_hi_first = c.sparse && a0.q_err_buf != nullptr && (int64_t)ix->K * 256 < (1ll << 32) && !s->opt.is(FLMR_OPT_S0_IMPL, "f16rs");
    return FLMR_OK;
}

// S0 (scores, cells), candidates (+ hit flags), S1, top-ndocs selection -> s->s1_pids / s1_count (+ optional global keys)
static int stage_cand_s1(run_ctx& c, uint64_t* out_keys);

static int stage_s0_s1(run_ctx& c, uint64_t* out_keys) {
    hipStream_t st = c.st;
    RUN(mark(c));
    RUN(flmr_launch_centroid_scores(c.a0, st));
    RUN(mark(c));
    RUN(flmr_launch_select_cells(c.a0, st));
    RUN(mark(c));
    return stage_cand_s1(c, out_keys);
}

// candidates (+ hit flags), S1, top-ndocs selection, given idx_bits / cells / ncell (and the table rows of idx) in place
static int stage_cand_s1(run_ctx& c, uint64_t* out_keys) {
    flmr_searcher* s = c.s;
    const flmr_index* ix = s->ix;
    hipStream_t st = c.st;
    // FLMR_CAND_IMPL=atomic keeps the first implementation (global atomicOr bitmap + separate hit bitmap) for A/B runs
    const bool chunked = !s->opt.is(FLMR_OPT_CAND_IMPL, "atomic");
    const bool use_hits = !s->opt.has(FLMR_OPT_S1_NO_HITMAP);
    bool scatter = false;
    s->last_scatter = false;
    // the dense stage-1 forms (flmr_stage1_dense.hip) need the sparse path, one column tile and the CPU-path numerics
    const bool dense = s->s1d_img_rows > 0 && chunked && c.sparse && c.ncol == 32 && !c.f.f16_round && !s->opt.is(FLMR_OPT_S1_IMPL, "scan");
    s->last_dense = dense;
    const int32_t* sel_counts = s->cand_count;
    bool modes_fused = false;
    flmr_cand_args ca{};
    {
        ca.nqueries = c.nqueries; ca.idx_words = s->idx_words; ca.max_cells = s->max_cells; ca.qmax = s->qmax;
        ca.nchunks = ix->nchunks; ca.words = s->bitmap_words; ca.cand_cap = s->cand_cap;
        ca.idx_bits = s->idx_bits; ca.cells = s->cells; ca.ncell = s->ncell;
        ca.ivf_pids = ix->ivf_pids; ca.ivf_offsets = ix->ivf_offsets; ca.chunk_tab = ix->ivf_chunk_tab;
        ca.qual = s->qual; ca.nqual = s->nqual; ca.hit_valid = s->hit_valid;
        ca.cand_bits = s->bitmap; ca.hit_bits = s->hit_bits; ca.chunk_cnt = s->chunk_cnt;
        ca.cand = s->cand; ca.cand_hit = s->cand_hit; ca.cand_count = s->cand_count; ca.overflow = s->overflow;
        // stage 1 by scatter over the surviving centroids' IVF lists (single column tile, sparse or full table alike);
        // FLMR_S1_IMPL=scan keeps the code-scanning kernel for every query (A/B runs, cross-check tests)
        scatter = chunked && use_hits && c.ncol == 32 && !s->opt.is(FLMR_OPT_S1_IMPL, "scan");
        ca.scatter = scatter ? 1 : 0;
        modes_fused = scatter && !s->opt.is(FLMR_OPT_S1_IMPL, "slots");   // (cand_plan_kernel runs)
        ca.cs = c.f.cs; ca.cs_query_stride = c.f.cs_query_stride; ca.nq_cand = c.nqc; ca.q_lens = c.q_lens;
        ca.cs_compact = c.f.cs_compact; ca.idx_prefix = s->idx_prefix;
        ca.rows_out = c.sparse ? s->rows : nullptr; ca.cen16 = ix->centroids_f16; ca.q_hi = s->q_hi; ca.q_lo = s->q_lo;
        ca.keys = s->keys1; ca.key_count = s->key_count; ca.chunk_hits = s->chunk_hits; ca.n_select = c.p.ndocs;
        ca.f16_round = c.f.f16_round;
        ca.fast_state = s->opt.is(FLMR_OPT_S1_IMPL, "slots") ? nullptr : s->cand_fast;
        ca.row_ovf = s->row_ovf;
        // who takes which query afterwards: the list-scatter forms (hit_valid), the dense forms, the recompute form (row_ovf), the scan
        // (the rest); decided per query by cand_plan_kernel when it runs, else by flmr_launch_s1_dense_modes below
        if (modes_fused) {
            ca.s1d_mode = s->s1d_mode; ca.s1d_scan_skip = s->s1d_scan_skip; ca.s1d_any = s->s1d_any;
            ca.s1d_img_rows = dense ? s->s1d_img_rows : 0;
            ca.s1d_exact_too = (dense && !s->opt.is(FLMR_OPT_S1_IMPL, "image")) ? 1 : 0;
        }
    }
    if (chunked) {
        RUN(flmr_launch_candidates_chunked(ca, st));
        s->last_ca = ca; s->last_scatter = scatter;
        RUN(mark(c));
        RUN(mark(c));  // (the hit set is produced by the same pass: the s1_hitmap stage is empty in this mode)
    } else {
        if (c.sparse) RUN(flmr_launch_qualifying(ca, st));   // (the compact score rows stage 1 reads)
        RUN(flmr_launch_ivf_mark(s->cells, s->ncell, s->max_cells, c.nqueries, ix->ivf_pids, ix->ivf_offsets, s->bitmap,
                                 s->bitmap_words, st));
        RUN(flmr_launch_compact(s->bitmap, s->bitmap_words, ix->num_passages, c.nqueries, s->cand, s->cand_cap,
                                s->cand_count, s->overflow, st));
        RUN(mark(c));
        if (use_hits)
            RUN(flmr_launch_hit_bitmap(s->idx_bits, s->idx_words, c.nqueries, ix->ivf_pids, ix->ivf_offsets, s->cand_count,
                                       s->hit_bits, s->bitmap_words, s->hit_valid, st));
        RUN(mark(c));
    }
    // Queries the list-scatter forms did not take (hit_valid == 0: more surviving centroids or longer lists than they handle):
    // the dense forms of flmr_stage1_dense.hip -- fp16 images of the query's score rows in LDS, upper-bound keys for every
    // candidate, the band around the cut rescored exactly -- where the rows fit (<= ~2 k survivors), the same kernel's exact form
    // beyond; the round-5 scan keeps the cases those do not cover (several column tiles, the fp16 numerics mode, no sorted copy).
    const bool exact_too = dense && !s->opt.is(FLMR_OPT_S1_IMPL, "image");   // (development: "image" leaves the queries beyond the images to the scan)
    if (!modes_fused)
        RUN(flmr_launch_s1_dense_modes(scatter ? s->hit_valid : nullptr, s->nqual, s->row_ovf, c.nqueries, dense ? s->s1d_img_rows : 0,
                                       exact_too ? 1 : 0, s->s1d_mode, s->s1d_scan_skip, s->s1d_any, st));
    const int32_t* scan_skip = s->s1d_scan_skip;
    if (dense) {
        flmr_s1d_args d{};
        d.codes = ix->codes_sorted; d.offsets = ix->doc_offsets; d.ulen = ix->doc_ulen; d.codes_len = ix->N;
        d.idx_bits = s->idx_bits; d.idx_prefix = s->idx_prefix; d.idx_words = s->idx_words;
        d.rows = s->rows; d.row_cap = s->row_cap; d.nqual = s->nqual; d.q_lens = c.q_lens; d.nq_cand = c.nqc; d.nqueries = c.nqueries;
        d.cand = s->cand; d.cand_stride = s->cand_cap; d.cand_count = s->cand_count;
        d.band = s->s1d_band; d.band_count = s->s1d_band_count; d.mode = s->s1d_mode; d.keys = s->keys1; d.img_err = s->s1d_err; d.any = s->s1d_any;
        d.parts = 0; d.group = 64;
        RUN(flmr_launch_s1_image(d, ix->mean_ulen, st));
        RUN(flmr_launch_s1_band(s->keys1, s->cand_cap, s->cand_count, s->s1d_mode, s->s1d_err, c.nqueries, c.p.ndocs, s->s1d_band,
                                s->s1d_band_count, s->s1d_in_count, st));
        // (16 parts a query: 2 queries' rows per XCD at a time stay in its L2 -- profiles/r06/s1_dense_probe_v5.txt; groups of 16
        // candidates for a band, 32 for a whole list: the kernel's choice per query)
        d.parts = 16; d.group = 0;
        RUN(flmr_launch_s1_exact(d, ix->mean_ulen, st));
        sel_counts = s->s1d_in_count;
    }
    if (c.sparse && s->row_cap < ix->K)   // (a query over the score-row capacity; leaves at once when there is none)
        RUN(flmr_launch_filter_stage1_recompute(c.f, s->idx_bits, s->idx_words, s->cand, s->cand_cap, s->cand_count, s->row_ovf, s->keys1,
                                                ix->centroids_f16, s->q_hi, s->q_lo, st));
    if (!(dense && exact_too))   // (with both dense forms on, every query the list scatter leaves is theirs or the recompute form's: nothing to scan)
        RUN(flmr_launch_filter_stage1(c.f, s->idx_bits, s->idx_words, s->cand, s->cand_cap, s->cand_count, s->keys1,
                                      (use_hits && !chunked) ? s->hit_bits : nullptr, s->bitmap_words, use_hits ? s->hit_valid : nullptr,
                                      (use_hits && chunked) ? s->cand_hit : nullptr, st, scan_skip));
    RUN(mark(c));
    RUN(flmr_launch_select_topn(s->keys1, s->cand_cap, sel_counts, c.nqueries, c.p.ndocs, s->s1_pids, s->maxp.ndocs,
                                s->s1_count, st, out_keys, (uint64_t)ix->pid_base));
    RUN(mark(c));
    return push_status(c);
}

// S2 over s->s1_pids / s1_count -> s->keys2 (slot-aligned with s1_pids)
// `whole_batch`: the survivors are the full top-ndocs lists of a single-index search (flmr_search_batch).  Phase 2 of the
// sharded protocol scores only this shard's members (a fraction of ndocs per query): the table walk, whose cost does not
// shrink with the number of survivors, is then the wrong form unless forced; so is the XCD-sliced kernel, whose waves each take
// one eighth of 64 survivors' tokens (per-rank compute of a step at 8 shards, profiles/shard_step_model.py: 2.16 ms sliced,
// 1.92 ms gather).
static int stage_s2(run_ctx& c, bool whole_batch) {
    flmr_searcher* s = c.s;
    const flmr_index* ix = s->ix;
    const flmr_options& o = s->opt;
    const bool walk = c.sparse && !c.f.f16_round && ix->codes_sorted && ix->centroids_f16_tiled &&
                      (o.is(FLMR_OPT_S2_IMPL, "walk") ||
                       (!o.has(FLMR_OPT_S2_IMPL) && whole_batch && flmr_stage2_walk_pays(ix, c.nqueries, c.p.ndocs)));
    const bool xcd = !walk && c.sparse && s->s2_part && c.f.ncol == 32 &&
                     (o.is(FLMR_OPT_S2_IMPL, "xcd") || (!o.has(FLMR_OPT_S2_IMPL) && whole_batch && flmr_stage2_xcd_pays(ix)));
    if (xcd)
        RUN(flmr_launch_filter_stage2_xcd(c.f, s->s1_pids, s->maxp.ndocs, s->s1_count, c.p.ndocs, s->keys2, s->maxp.ndocs, ix,
                                          s->q_hi, s->q_lo, s->s2_part, s->maxp.ndocs, c.st));
    else if (walk)
        RUN(flmr_launch_filter_stage2_walk(c.f, s->s1_pids, s->maxp.ndocs, s->s1_count, c.p.ndocs, s->keys2, s->maxp.ndocs,
                                           ix->centroids_f16_tiled, s->q_hi, s->q_lo, ix->codes_sorted, c.st));
    else if (c.sparse)
        RUN(flmr_launch_filter_stage2_mfma(c.f, s->s1_pids, s->maxp.ndocs, s->s1_count, c.p.ndocs, s->keys2, s->maxp.ndocs,
                                           ix->centroids_f16, s->q_hi, s->q_lo, c.st));
    else
        RUN(flmr_launch_filter_stage2(c.f, s->s1_pids, s->maxp.ndocs, s->s1_count, c.p.ndocs, s->keys2, s->maxp.ndocs, c.st));
    return FLMR_OK;
}

// S2 of a whole batch + the selection of the ndocs/4 survivors -> s->s2_pids / s2_count.
// Default on the sliced kernel (sparse path, CPU-path numerics, the batch's error bounds at hand): hi-only pass, plan, full
// rescoring of the band by the gather kernel, finish (flmr_filter.hip) -- the same SET as the sorted form.  FLMR_S2_IMPL set to
// anything but "xcda" keeps the full-score forms and the sorted list (taps, cross-check tests).
static int stage_s2_select(run_ctx& c) {
    flmr_searcher* s = c.s;
    const flmr_index* ix = s->ix;
    const flmr_options& o = s->opt;
    const int nd4m = s->maxp.ndocs / 4;
    const bool bounds = c.sparse && !c.f.f16_round && c.a0.q_err_buf != nullptr && c.f.ncol == 32 &&
                        c.a0.centroids_f16 && (int64_t)ix->K * 256 < (1ll << 32) && !o.is(FLMR_OPT_S0_IMPL, "f16rs");
    const bool approx = bounds && s->s2_part && (o.is(FLMR_OPT_S2_IMPL, "xcda") || (!o.has(FLMR_OPT_S2_IMPL) && flmr_stage2_xcd_pays(ix) &&
                        !(ix->codes_sorted && ix->centroids_f16_tiled && flmr_stage2_walk_pays(ix, c.nqueries, c.p.ndocs))));
    if (!approx) {
        RUN(stage_s2(c, true));
        return flmr_launch_sort_topn(s->keys2, s->maxp.ndocs, s->s1_count, c.p.ndocs, c.nqueries, c.p.ndocs / 4, s->s2_pids,
                                     nullptr, nd4m, s->s2_count, 0, 0, c.st);
    }
    RUN(flmr_launch_filter_stage2_xcd_ex(c.f, s->s1_pids, s->maxp.ndocs, s->s1_count, c.p.ndocs, s->keys2, s->maxp.ndocs, ix,
                                         s->q_hi, s->q_lo, s->s2_part, s->maxp.ndocs, true, c.st));
    RUN(flmr_launch_s2_refine_plan(s->keys2, s->maxp.ndocs, s->s1_count, c.p.ndocs, c.nqueries, c.p.ndocs / 4, s->q_err_sum,
                                   s->s2_pids, nd4m, s->s2_band, s->maxp.ndocs, s->s2_band_count, s->s2_need, s->s2_def, c.st));
    RUN(flmr_launch_filter_stage2_mfma(c.f, s->s2_band, s->maxp.ndocs, s->s2_band_count, c.p.ndocs, s->keys2b, s->maxp.ndocs,
                                       ix->centroids_f16, s->q_hi, s->q_lo, c.st));
    return flmr_launch_s2_refine_finish(s->keys2b, s->maxp.ndocs, s->s2_band_count, s->s2_need, s->s2_def, c.p.ndocs, c.nqueries,
                                        s->s2_pids, nd4m, s->s2_count, c.st);
}

// S3 over s->s2_pids / s2_count -> s->keys3 / doc_scores (slot-aligned with s2_pids)
// `s0_images`: stage 0 of THIS call split the same queries (flmr_search_batch; not the phase entry points)
static int stage_s3(run_ctx& c, bool s0_images) {
    flmr_searcher* s = c.s;
    flmr_maxsim_args m{};
    m.ix = s->ix; m.Q = c.Q; m.q_lens = c.q_lens; m.nqueries = c.nqueries; m.nq = c.nq;
    m.pids = s->s2_pids; m.pid_stride = s->maxp.ndocs / 4; m.counts = s->s2_count; m.max_count = c.p.ndocs / 4;
    m.keys = s->keys3; m.key_stride = s->maxp.ndocs / 4; m.scores = s->doc_scores;
    m.q_hi = s->q3_hi; m.q_lo = s->q3_lo;
    m.gpu_fp16 = c.f.f16_round;
    m.plan_desc = s->s3_desc; m.plan_stride = s->s3_desc_stride; m.plan_wbeg = s->s3_wbeg; m.plan_wcap = s->s3_wcap;
    m.colmax_ws = s->s3_colmax; m.colmax_cap = s->s3_colmax_cap;
    // Stage 0's fp16 images ARE stage 3's when every query row is a candidate-generation column (nq <= 32 <= nq_cand): rows
    // below min(q_len, nq) real, the rest zero, one tile of 32 -- no second split launch
    if (s0_images && c.sparse && !c.f.f16_round && c.nq <= 32 && c.nqc == c.nq && c.ncol == 32) {
        m.q_hi = s->q_hi; m.q_lo = s->q_lo;
        m.q_split_done = 1;
    }
    return flmr_launch_maxsim(m, c.st);
}

extern "C" int flmr_search_batch(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries,
                                 int32_t nq, const flmr_search_params_t* p, int32_t* out_pids, float* out_scores,
                                 int32_t* out_counts, flmr_stream_t stream) {
    if (!out_pids || !out_scores || !out_counts) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    RUN(stage_s0_s1(c, nullptr));
    // ---- S2: full centroid MaxSim over the survivors, keep ndocs/4 (in (score,pid) order, or -- approximate-then-refine --
    // as the same set) ----------------
    RUN(stage_s2_select(c));
    RUN(mark(c));
    // ---- S3: decompress + normalise + MaxSim --------------------------------------------------------
    RUN(stage_s3(c, true));
    RUN(mark(c));
    // ---- S4: final ranking, global pids ---------------------------------------------------------------
    RUN(flmr_launch_sort_topn(s->keys3, s->maxp.ndocs / 4, s->s2_count, p->ndocs / 4, nqueries, p->k, out_pids,
                              out_scores, p->k, out_counts, s->ix->pid_base, 1, c.st));
    RUN(mark(c));
    if (s->profiling) {   // the set is complete: count it
        s->ev_pending++;
        s->have_ms = true;
    }
    return FLMR_OK;
}

// ---- exact sharded protocol ("exact-parity mode"): three phases with a key exchange after each --------------
// phase 1: S0..S1 on this shard -> its top-ndocs stage-1 keys (global pids), 0 padded, unordered.
extern "C" int flmr_search_phase1(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                                  const flmr_search_params_t* p, uint64_t* out_keys, flmr_stream_t stream) {
    if (!out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    if (s->maxp.ndocs != p->ndocs) FLMR_FAIL(FLMR_ERR_INVALID, "the phased protocol needs ndocs == the searcher's max ndocs (key rows are ndocs wide)");
    return stage_s0_s1(c, out_keys);
}

// ---- query-split stage 0 (optional phase 0 of the sharded protocol) ----------------------------------------------------
// Stage 0 does not depend on the passage shard, so W ranks would repeat the same K x 128 x 32 product per query.  Instead
// rank r runs flmr_search_probe for its slice of the batch, the ranks all-gather (idx bitset, cells, ncell) -- K/8 bytes
// + a few cells per query -- and flmr_search_phase1_probed rebuilds the sparse score-table rows from the bitset.
extern "C" int flmr_searcher_probe_dims(const flmr_searcher_t* s, int32_t* idx_words, int32_t* max_cells) {
    if (!s || !idx_words || !max_cells) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    *idx_words = s->idx_words;
    *max_cells = s->max_cells;
    return FLMR_OK;
}

extern "C" int flmr_search_probe(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                                 const flmr_search_params_t* p, int32_t q_begin, int32_t q_count, uint32_t* out_idx_bits,
                                 int32_t* out_cells, int32_t* out_ncell, flmr_stream_t stream) {
    if (q_count == 0 && s && q_begin >= 0 && q_begin <= nqueries) return FLMR_OK;  // an empty slice (more ranks than queries) has no outputs
    if (!out_idx_bits || !out_cells || !out_ncell) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    if (!c.sparse) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "query-split stage 0 needs the sparse-table path (fp16-exact centroids, K %% 64 == 0, nq_cand <= 32)");
    if (q_begin < 0 || q_count < 0 || q_begin + q_count > nqueries) FLMR_FAIL(FLMR_ERR_INVALID, "query slice [%d, %d) outside the batch of %d", q_begin, q_begin + q_count, nqueries);
    flmr_s0_args a0 = c.a0;  // the slice uses workspace slots 0..q_count; its results go straight to the caller's buffers
    a0.Q = Q + (size_t)q_begin * nq * FLMR_DIM;
    a0.q_lens = c.q_lens ? c.q_lens + q_begin : nullptr;
    a0.nqueries = q_count;
    a0.idx_bits = out_idx_bits; a0.cells = out_cells; a0.ncell = out_ncell;
    RUN(flmr_launch_centroid_scores(a0, c.st));
    return flmr_launch_select_cells(a0, c.st);
}

extern "C" int flmr_search_phase1_probed(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries,
                                         int32_t nq, const flmr_search_params_t* p, const uint32_t* idx_bits,
                                         const int32_t* cells, const int32_t* ncell, uint64_t* out_keys,
                                         flmr_stream_t stream) {
    if (!idx_bits || !cells || !ncell || !out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    if (!c.sparse) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "query-split stage 0 needs the sparse-table path (fp16-exact centroids, K %% 64 == 0, nq_cand <= 32)");
    if (s->maxp.ndocs != p->ndocs) FLMR_FAIL(FLMR_ERR_INVALID, "the phased protocol needs ndocs == the searcher's max ndocs (key rows are ndocs wide)");
    FLMR_HIP(hipMemcpyAsync(s->idx_bits, idx_bits, (size_t)nqueries * s->idx_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c.st));
    FLMR_HIP(hipMemcpyAsync(s->cells, cells, (size_t)nqueries * s->max_cells * sizeof(int32_t), hipMemcpyDeviceToDevice, c.st));
    FLMR_HIP(hipMemcpyAsync(s->ncell, ncell, (size_t)nqueries * sizeof(int32_t), hipMemcpyDeviceToDevice, c.st));
    RUN(mark(c));
    RUN(flmr_launch_split_q(c.a0, c.st));   // (the score rows of the surviving centroids are rebuilt from the bitset by qualifying_kernel)
    RUN(mark(c));
    RUN(mark(c));
    return stage_cand_s1(c, out_keys);
}

// phase 2: global_s1 [nqueries, n_in] = the GLOBAL top-ndocs stage-1 keys; this shard scores its own members in stage 2
// -> out_keys [nqueries, ndocs], slot j = the key of global_s1[j]'s passage when it lives on this shard, else 0 (so the
// shards' outputs combine by a SUM all-reduce or by a gather).  Q / q_lens / params must be those of phase 1.
extern "C" int flmr_search_phase2(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                                  const flmr_search_params_t* p, const uint64_t* global_s1, int32_t n_in, uint64_t* out_keys,
                                  flmr_stream_t stream) {
    if (!global_s1 || !out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    if (n_in > p->ndocs) FLMR_FAIL(FLMR_ERR_INVALID, "n_in=%d > ndocs=%d", n_in, p->ndocs);
    RUN(flmr_launch_filter_local_keys(global_s1, nqueries, n_in, s->ix->pid_base, s->ix->num_passages, s->s1_pids, s->maxp.ndocs,
                                      s->s1_count, c.st, s->s1_slot));
    RUN(stage_s2(c, false));
    return flmr_launch_export_keys_slotted(s->keys2, s->maxp.ndocs, s->s1_count, s->s1_slot, nqueries, (uint64_t)s->ix->pid_base,
                                           p->ndocs, out_keys, c.st);
}

// phase 3: global_s2 [nqueries, n_in] = the GLOBAL top-(ndocs/4) stage-2 keys; this shard computes the exact MaxSim of its
// own members -> out_keys [nqueries, ndocs/4], slot-aligned with global_s2 like phase 2.
extern "C" int flmr_search_phase3(flmr_searcher_t* s, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                                  const flmr_search_params_t* p, const uint64_t* global_s2, int32_t n_in, uint64_t* out_keys,
                                  flmr_stream_t stream) {
    if (!global_s2 || !out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    run_ctx c(s);
    RUN(prepare_ctx(c, s, Q, q_lens, nqueries, nq, p, stream));
    if (n_in > p->ndocs / 4) FLMR_FAIL(FLMR_ERR_INVALID, "n_in=%d > ndocs/4=%d", n_in, p->ndocs / 4);
    RUN(flmr_launch_filter_local_keys(global_s2, nqueries, n_in, s->ix->pid_base, s->ix->num_passages, s->s2_pids,
                                      s->maxp.ndocs / 4, s->s2_count, c.st, s->s2_slot));
    RUN(stage_s3(c, false));
    return flmr_launch_export_keys_slotted(s->keys3, s->maxp.ndocs / 4, s->s2_count, s->s2_slot, nqueries,
                                           (uint64_t)s->ix->pid_base, p->ndocs / 4, out_keys, c.st);
}

// keys [nqueries, m] -> the n largest, descending, 0 padded (+ optional counts); m <= 8192
extern "C" int flmr_topn_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out_keys,
                              int32_t* out_counts, flmr_stream_t stream) {
    if (!keys || !out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    return flmr_launch_sort_keys_topn(keys, nqueries, m, n, out_keys, out_counts, reinterpret_cast<hipStream_t>(stream));
}

// keys [nqueries, m] -> the n largest, UNORDERED, 0 padded (any m)
extern "C" int flmr_select_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out_keys,
                                flmr_stream_t stream) {
    if (!keys || !out_keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (m < 1 || n < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes m=%d n=%d", m, n);
    return flmr_launch_select_keys(keys, nqueries, m, n, out_keys, reinterpret_cast<hipStream_t>(stream));
}

// descending keys [nqueries, n] -> pids / scores / counts of the first k
extern "C" int flmr_unpack_keys(const uint64_t* keys, int32_t nqueries, int32_t n, int32_t k, int32_t* out_pids,
                                float* out_scores, int32_t* out_counts, flmr_stream_t stream) {
    if (!keys || !out_pids || !out_scores || !out_counts) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    return flmr_launch_unpack_keys(keys, nqueries, n, k, out_pids, out_scores, out_counts, reinterpret_cast<hipStream_t>(stream));
}
#undef RUN

extern "C" int flmr_searcher_tap(flmr_searcher_t* s, int32_t what, int32_t q, void* host_out, int64_t capacity,
                                 int64_t* count) {
    if (!s || !host_out || !count) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (q < 0 || q >= s->last_nqueries) FLMR_FAIL(FLMR_ERR_INVALID, "query %d outside the last batch (%d)", q, s->last_nqueries);
    FLMR_HIP(hipStreamSynchronize(s->last_stream));
    {
        const int rc = poll_status(s, true);
        if (rc) return rc;
    }
    flmr_opt_scope scope(&s->opt);
    const flmr_index* ix = s->ix;
    const void* src = nullptr;
    int64_t n = 0;
    size_t esz = 4;
    int32_t c = 0;
    const int nd4 = s->maxp.ndocs / 4;
    switch (what) {
        case FLMR_TAP_CENTROID_SCORES:
            if (!s->last_full_table) FLMR_FAIL(FLMR_ERR_INVALID, "the last batch ran with the sparse score table: call flmr_searcher_set_full_table(s, 1) first");
            n = (int64_t)ix->K * s->last_ncol; src = s->cs + (size_t)q * n; break;
        case FLMR_TAP_IDX_BITS:
            n = s->idx_words; src = s->idx_bits + (size_t)q * n; break;
        case FLMR_TAP_CELLS:
            FLMR_HIP(hipMemcpy(&c, s->ncell + q, 4, hipMemcpyDeviceToHost));
            n = c; src = s->cells + (size_t)q * s->max_cells; break;
        case FLMR_TAP_CANDIDATES:
            if (s->last_scatter) {  // the hot path skipped the ascending lists: build them now from the bitmaps
                int rc = flmr_launch_cand_emit_all(s->last_ca, s->last_stream);
                if (rc) return rc;
                FLMR_HIP(hipStreamSynchronize(s->last_stream));
                s->last_scatter = false;
            }
            FLMR_HIP(hipMemcpy(&c, s->cand_count + q, 4, hipMemcpyDeviceToHost));
            n = c; src = s->cand + (size_t)q * s->cand_cap; break;
        case FLMR_TAP_STAGE1:
            FLMR_HIP(hipMemcpy(&c, s->s1_count + q, 4, hipMemcpyDeviceToHost));
            n = c; src = s->s1_pids + (size_t)q * s->maxp.ndocs; break;
        case FLMR_TAP_STAGE2:
            FLMR_HIP(hipMemcpy(&c, s->s2_count + q, 4, hipMemcpyDeviceToHost));
            n = c; src = s->s2_pids + (size_t)q * nd4; break;
        case FLMR_TAP_DOC_SCORES:
            FLMR_HIP(hipMemcpy(&c, s->s2_count + q, 4, hipMemcpyDeviceToHost));
            n = c; src = s->doc_scores + (size_t)q * nd4; break;
        case FLMR_TAP_Q_ERR:
            n = s->last_hi_first ? 32 : 0; src = s->q_err + (size_t)q * s->ncol_max; break;
        case FLMR_TAP_Q_ERR_SUM:
            n = s->last_hi_first ? 1 : 0; src = s->q_err_sum + q; break;
        case FLMR_TAP_STAGE1_FORM:
            {   // 7: a query over the score-row capacity (stage 1 recomputed from the centroids)
                int32_t ro = 0;
                FLMR_HIP(hipMemcpy(&ro, s->row_ovf + q, 4, hipMemcpyDeviceToHost));
                if (ro) {
                    *count = 1;
                    if (capacity < 1) FLMR_FAIL(FLMR_ERR_CAPACITY, "tap needs 1 element");
                    *static_cast<int32_t*>(host_out) = 7;
                    return FLMR_OK;
                }
            }
            if (s->last_dense) {   // a query the dense forms took: 5 (images + band) / 6 (exact rows)
                int32_t m = 0;
                FLMR_HIP(hipMemcpy(&m, s->s1d_mode + q, 4, hipMemcpyDeviceToHost));
                if (m != FLMR_S1D_SKIP) {
                    *count = 1;
                    if (capacity < 1) FLMR_FAIL(FLMR_ERR_CAPACITY, "tap needs 1 element");
                    *static_cast<int32_t*>(host_out) = 4 + m;
                    return FLMR_OK;
                }
            }
            n = (s->last_ca.scatter && s->last_ca.fast_state) ? 1 : 0; src = s->cand_fast + FLMR_FAST_HDR + q; break;
        default: FLMR_FAIL(FLMR_ERR_INVALID, "unknown tap %d", what);
    }
    *count = n;
    if (n > capacity) FLMR_FAIL(FLMR_ERR_CAPACITY, "tap needs %lld elements, capacity %lld", (long long)n, (long long)capacity);
    if (n) FLMR_HIP(hipMemcpy(host_out, src, (size_t)n * esz, hipMemcpyDeviceToHost));
    return FLMR_OK;
}
