// Internal helpers shared by the HIP translation units of libflmr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "flmr_hip.h"

#define FLMR_DIM 128            // embedding width this build is specialised for
#define FLMR_WAVE 64            // CDNA wavefront
#define FLMR_MAX_NCELLS 8
#define FLMR_MAX_NQ_CAND 128    // candidate-generation width limit (4 column tiles of 32)
#define FLMR_MAX_NDOCS 8192
#define FLMR_CODE_PAD 256        // ints readable behind the sorted code copy (the dense stage 1 reads whole 16-byte pieces)

extern thread_local char flmr_err_buf[512];

#define FLMR_FAIL(code, ...)                                     \
    do {                                                         \
        snprintf(flmr_err_buf, sizeof(flmr_err_buf), __VA_ARGS__); \
        return (code);                                           \
    } while (0)

#define FLMR_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            snprintf(flmr_err_buf, sizeof(flmr_err_buf), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
                     hipGetErrorString(e__));                                                       \
            return FLMR_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

#define FLMR_LAUNCH_CHECK() FLMR_HIP(hipGetLastError())

// ---- variant switches (A/B runs, cross-check tests) ------------------------------------------------------------------
// The FLMR_* environment variables are read ONCE per process (first use of the library); afterwards the table changes only
// through flmr_set_option().  A searcher snapshots the table when it is created and every launch of its batches sees that
// snapshot (flmr_opt_scope): nothing on the per-batch launch path calls getenv, and the environment cannot flip a running
// searcher to another kernel path.
enum flmr_opt_id {
    FLMR_OPT_S0_IMPL = 0,    // (unset: fp16-split, query-stationary + "hi first" on the sparse path) | qs1 ("hi first" with the dense epilogue inside the loop) | f16 (both products everywhere) | f16rs (row-stationary fp16 kernel) | f32 | mfma | valu
    FLMR_OPT_FULL_TABLE,     // set: keep the whole centroid-score table
    FLMR_OPT_CAND_IMPL,      // atomic: first candidate-generation implementation
    FLMR_OPT_S1_NO_HITMAP,   // set: no hit prefilter
    FLMR_OPT_S1_IMPL,        // scan: code-scanning stage 1 for every query | slots: the slot form of the scatter kernel for every query (default: its queue form first)
    FLMR_OPT_S2_IMPL,        // xcda (approximate-then-refine on the sliced kernel: the default where the sliced kernel is) | xcd | walk | lds | ldsb | regs: force the XCD-sliced gather / the dense walk / the LDS-DMA gather (4-wave blocks; 16-wave blocks with the query operand in LDS) / the register gather (default: cost model)
    FLMR_OPT_S0_STAGED,      // set: staged epilogue for every tile
    FLMR_OPT_S3_NO_MULTIQ,   // set: single-tile MaxSim kernel for long queries too
    FLMR_OPT_S3_IMPL,        // lean (default for Nq <= 32: the cw arithmetic on planned tiles) | cw: (c.q + w.q) * 1/norm with table-decoded weights; regs / dma: decompress-normalise-split kernel with register / LDS-DMA row gathers; f32: fp32-MFMA kernel
    FLMR_OPT_SCORE_IMPL,     // valu: plain-FMA padded scorer
    FLMR_OPT_POISON,         // set (test runs): every batch starts with its per-query score rows, ranks and survivor lists filled with 0xFF -- a stage that lives off an earlier batch's workspace then produces NaNs / wild indices instead of passing
    FLMR_OPT_ROW_CAP,        // score rows a searcher keeps per query (64 .. 65535; default 16384, at most K): a query with more centroids above the threshold raises FLMR_ERR_CAPACITY
    FLMR_OPT_COUNT
};
struct flmr_options {
    char v[FLMR_OPT_COUNT][16];
    bool has(int id) const { return v[id][0] != 0; }
    bool is(int id, const char* x) const { return strcmp(v[id], x) == 0; }
};
const flmr_options& flmr_opts();                        // the active snapshot (searcher scope) or the process table
const flmr_options& flmr_process_options();             // process table (environment resolved on first use)
struct flmr_opt_scope {                                 // RAII: launches inside the scope see `o`
    const flmr_options* prev;
    explicit flmr_opt_scope(const flmr_options* o);
    ~flmr_opt_scope();
};

static inline int64_t flmr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t flmr_round_up(int64_t a, int64_t b) { return flmr_ceil_div(a, b) * b; }

// ---- (score, pid) keys -----------------------------------------------------------------------
// The reference selects with std::priority_queue<std::pair<float,int>> (filter_pids.cpp:24): descending
// lexicographic (score, pid).  A 64-bit key whose unsigned order equals that order lets one integer
// compare / radix pass do the same: high word = order-preserving map of the fp32 score, low word = pid.
__host__ __device__ static inline uint32_t flmr_f2ord(float f) {
    if (f == 0.0f) f = 0.0f;  // -0.0 == +0.0 for the reference's float compare
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float flmr_ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ static inline uint64_t flmr_make_key(float score, int32_t pid) {
    return ((uint64_t)flmr_f2ord(score) << 32) | (uint32_t)pid;
}
__host__ __device__ static inline int32_t flmr_key_pid(uint64_t key) { return (int32_t)(uint32_t)key; }
__host__ __device__ static inline float flmr_key_score(uint64_t key) { return flmr_ord2f((uint32_t)(key >> 32)); }

// ---- index object ----------------------------------------------------------------------------
struct flmr_index {
    int32_t dim, nbits, K, device;
    int64_t N, num_passages, pid_base;
    int32_t packed_dim;  // dim*nbits/8 bytes per token
    bool owns;           // arrays below were hipMalloc'ed by flmr_index_open
    int32_t* codes;
    uint8_t* residuals;
    int64_t* doc_offsets;
    int32_t* ivf_pids;
    int64_t* ivf_offsets;
    float* centroids;
    float* wlut;  // [256][8/nbits] fused decode table: bucket_weights[lut[rev[byte]][l]]
    _Float16* centroids_f16;  // [K,128] fp16 image of the centroids (only when centroids_f16_exact)
    uint32_t* wtab16;   // fp16 hi / lo images of wlut as MFMA A-operand fragments per residual byte (S3 "cw" form; flmr_maxsim.hip)
    float* inv_norm;    // [N + 64] 1 / max(||centroid + weights||, 1e-12) per token (S3 "cw" form); NULL when unavailable
    float bucket_weights[256];
    // host copy of the IVF list lengths sorted descending, prefix-summed: bound on #candidates for c cells
    int64_t* ivf_len_prefix;  // [K+1] host
    int64_t max_doclen;
    int32_t centroids_f16_exact;  // every centroid value is representable in fp16 (true for reference-format indexes)
    uint32_t* ivf_chunk_tab;      // [K][nchunks+1]: first entry of each IVF list with pid >= chunk*32768
    int32_t nchunks;
    int32_t* codes_sorted;        // [N + FLMR_CODE_PAD] per-passage ascending copy of `codes` with the DISTINCT values first (the rest of a
                                  // passage's run repeats its largest code: still ascending, the same set); NULL when a passage is too long
    uint16_t* doc_ulen;           // [num_passages] distinct codes per passage = the length of the distinct prefix in codes_sorted; NULL when
                                  // fewer than a tenth of the tokens repeat a code of their passage (every token then counts)
    double dup_share;             // share of the tokens that repeat a code of their passage (0 when codes_sorted was not built)
    double mean_ulen;             // mean number of distinct codes per passage
    _Float16* centroids_f16_tiled;  // centroids_f16 in MFMA A-operand order, one contiguous 1 KB run per (tile, k-step) (stage-2 walk)
    uint16_t* doc_splits;         // [num_passages][nslices]: codes_sorted position of the first code >= s * slice_rows (XCD-sliced stage 2)
    float cen_norm_max;           // >= the largest centroid 2-norm (bound of the "hi first" stage 0)
    int32_t nslices;              // 8, 16, 24 or 32: the fp16 centroid table cut so that a slice fits an XCD's L2
    int32_t slice_rows;           // ceil(K / nslices)
    int32_t xcd_round_robin;      // probed at open: workgroup L of a 1-D grid runs on XCD (L % 8) of 8
};
int flmr_build_sorted_codes(flmr_index* ix);
int flmr_build_tiled_centroids(flmr_index* ix);
int flmr_build_doc_splits(flmr_index* ix);
int flmr_build_s3_tables(flmr_index* ix);   // wtab16 + inv_norm (optional: left NULL when they cannot be built)

// build the fused byte -> (8/nbits) fp32 decode table from the codec tables (host)
void flmr_build_wlut(int nbits, const float* bucket_weights, const uint8_t* reversed_bit_map,
                     const uint8_t* combos, float* wlut /* [256 * 8/nbits] */);
void flmr_default_codec_tables(int nbits, uint8_t* reversed_bit_map /*256*/, uint8_t* combos /*256*8/nbits*/);

// ---- kernels' host launchers (defined in the stage .hip files) ---------------------------------
struct flmr_s0_args {
    const float* centroids;  // [K,128]
    const float* Q;          // [nqueries, nq, 128]
    const int32_t* q_lens;   // nullable
    int32_t K, nqueries, nq, nq_cand, ncol, ncells;
    float thr;
    float* cs;               // [nqueries, K, ncol]
    uint32_t* idx_bits;      // [nqueries, idx_words]
    int32_t idx_words;
    float* part_val;         // [nqueries, nblk, ncol, ncells]
    int32_t* part_idx;
    int32_t nblk;            // number of partial-list row blocks (set by flmr_launch_centroid_scores for the kernel it picks)
    int32_t* cells;          // [nqueries, max_cells]
    int32_t* ncell;          // [nqueries]
    int32_t max_cells;
    int32_t full_table;      // 1: store every row of the score table; 0: only rows of surviving centroids (fp16 S0, nq_cand<=32)
    const _Float16* centroids_f16;  // [K,128] (nullable)
    int32_t part_rows;       // 0: partials are per-block top-ncells lists; >0: partials are per-block column MAXIMA over
                             // `part_rows` centroid rows and s0_select_cells rescans the winning blocks in the table
    _Float16* q_hi;          // [nqueries, ncol, 128] fp16 split of Q (fp16 MFMA path)
    _Float16* q_lo;          //   Q ~= q_hi + q_lo * 2^-11
    int32_t centroids_f16_exact;
    int32_t q_hi_only;       // FLMR_NUMERICS_GPU_FP16: Q is ROUNDED to fp16 (q_lo = 0), as the reference's `Q.cuda().half()`
    // "hi first" stage 0 (sparse query-stationary path): the hi products alone give every score to within q_err[query][column];
    // the lo products are computed only for the tiles that can hold a surviving row, the block maxima stay hi-only and
    // s0_select_cells verifies its choice against that bound (flmr_stage0.hip).  q_err == NULL: both products everywhere.
    float* q_err;            // [nqueries, ncol] rigorous bound on |c . q_lo| / 2048 (+ the combine's rounding), written by s0_prepare_kernel
    float* q_err_buf;        // where s0_prepare_kernel writes (the sparse query-stationary path always fills it: stage 2's
                             // approximate-then-refine form reads it too); q_err == q_err_buf when stage 0 itself uses the bound
    float* q_err_sum;        // [nqueries] bound on |approximate - full| for a passage's stage-2 score (sum over the columns + summation rounding)
    float cen_norm_max;      // >= max_c ||c||_2
    int32_t grp_blocks;      // set by flmr_launch_centroid_scores: > 0 = the kernel it picked ALSO left one column maximum per `grp_blocks`
                             // consecutive 64-row blocks, as floats [nqueries, nblk / grp_blocks, ncol] in the part_idx buffer (unused on
                             // that path): s0_select_cells reads those first and only the best groups' block maxima after them
};
int flmr_launch_centroid_scores(flmr_s0_args& a, hipStream_t st);
int flmr_launch_select_cells(const flmr_s0_args& a, hipStream_t st);
int flmr_launch_centroid_argmax(flmr_s0_args& a, int32_t* out_codes, hipStream_t st);  // index build: nearest centroid per column
int flmr_launch_split_q(const flmr_s0_args& a, hipStream_t st);  // q_hi / q_lo images only (query-split stage 0)
int flmr_check_f16_exact(const float* dev, size_t n, int32_t* host_result);
int flmr_convert_f16(const float* dev, size_t n, _Float16* out);
int flmr_max_row_norm(const float* dev, int64_t rows, float* host_result);   // >= max_r ||row_r||_2 over [rows, 128] (rounded up)

int flmr_launch_ivf_mark(const int32_t* cells, const int32_t* ncell, int32_t max_cells, int32_t nqueries,
                         const int32_t* ivf_pids, const int64_t* ivf_offsets, uint32_t* bitmap, int64_t bitmap_words,
                         hipStream_t st);
int flmr_launch_compact(const uint32_t* bitmap, int64_t bitmap_words, int64_t num_passages, int32_t nqueries,
                        int32_t* cand, int64_t cand_cap, int32_t* cand_count, int32_t* overflow, hipStream_t st);

struct flmr_filter_args {
    const float* cs;           // score rows: the full table [nqueries, K, ncol] (row = centroid id, stride K*ncol), or -- cs_compact --
                               // only the rows of each query's surviving centroids, [nqueries, row_cap, ncol], row = RANK of the
                               // centroid among the set bits of the query's idx mask (written by qualifying_kernel)
    int64_t cs_query_stride;
    int32_t cs_compact;
    int32_t row_cap;            // compact form: rows per query (a rank beyond it -- a query over capacity, flagged by qualifying_kernel -- is clamped)
    const uint32_t* idx_prefix; // [nqueries, idx_words] exclusive popcount of the idx words before each word (compact rows' ranks)
    int32_t K, ncol, nq_cand, nqueries;
    const int32_t* q_lens;     // nullable (effective columns = min(q_len, nq_cand))
    const int32_t* codes;
    const int64_t* doclens;    // nullable -> offsets[p+1]-offsets[p]
    const int64_t* offsets;
    int32_t f16_round;         // FLMR_NUMERICS_GPU_FP16: column maxima rounded to fp16, fp32 sum rounded to fp16 (flmr_device.h)
    const uint16_t* ulen_sorted;  // nullable: distinct codes per passage -- the part of a passage's run in `codes_sorted` the sliced stage 2
                                  // reads (a maximum is idempotent: filter_pids.cpp:50-63)
};
// stage 1: candidates (cand[q*cand_stride + i], i < cand_count[q]) restricted to idx_bits -> keys
int flmr_launch_filter_stage1(const flmr_filter_args& f, const uint32_t* idx_bits, int32_t idx_words,
                              const int32_t* cand, int64_t cand_stride, const int32_t* cand_count, uint64_t* keys,
                              const uint32_t* hit_bits, int64_t hit_words, const int32_t* hit_valid,
                              const uint8_t* hit_flags, hipStream_t st, const int32_t* skip = nullptr);
// chunked candidate generation (flmr_candidates.hip): bitmaps in LDS per (query, 32768-passage chunk)
#define FLMR_FAST_HDR 0
struct flmr_cand_args {
    int32_t nqueries, idx_words, max_cells, qmax, nchunks;
    int64_t words, cand_cap;
    const uint32_t* idx_bits; const int32_t* cells; const int32_t* ncell;
    const int32_t* ivf_pids; const int64_t* ivf_offsets; const uint32_t* chunk_tab;
    int32_t* qual; int32_t* nqual; int32_t* hit_valid;     // [nqueries, qmax], [nqueries], [nqueries]
    uint32_t* cand_bits; uint32_t* hit_bits; int32_t* chunk_cnt;  // [nqueries, words] x2, [nqueries, nchunks]
    int32_t* cand; uint8_t* cand_hit; int32_t* cand_count; int32_t* overflow;
    // stage 1 by scatter over the surviving centroids' IVF lists (cand_mark_score_kernel), for queries with hit_valid
    int32_t scatter;                 // 0: bitmaps only (stage 1 = filter_stage1_kernel for every query)
    const float* cs; int64_t cs_query_stride; int32_t nq_cand; const int32_t* q_lens;
    // compact score rows (flmr_filter_args::cs_compact): qualifying_kernel computes them -- rows_out [nqueries, qmax, 32], row j =
    // the j-th surviving centroid's scores, the stage-0 MFMA sequence on the fp16 centroid rows (bitwise the values stage 0
    // computes) -- and `cs` points at the same buffer; rows_out == NULL: `cs` is the full table written by stage 0
    float* rows_out; const _Float16* cen16; const _Float16* q_hi; const _Float16* q_lo; uint32_t* idx_prefix;
    int32_t cs_compact;
    uint64_t* keys; int32_t* key_count;   // [nqueries, cand_cap] unordered stage-1 keys, [nqueries] running count
    int32_t* chunk_hits;                  // [nqueries, nchunks] candidates of the chunk that are in the hit set (scatter mode)
    int32_t n_select;                     // how many keys the selection after stage 1 keeps (ndocs)
    int32_t f16_round;                    // see flmr_filter_args
    // the queue and small-dense forms of the scatter kernel and their hand-over to the slot form: per query of the batch [b] the plan
    // word written by cand_plan_kernel from statistics measured on a sample of the query's chunks (0 queue form, FLMR_PLAN_SMALL
    // small-dense form, 1 slot form), overwritten by the forms with what happened (FLMR_TAP_STAGE1_FORM: 2 / 4 = handed over to the slot
    // form after the queue / small-dense form ran into a limit, 3 = done by the small-dense form), and [B + b] the fast forms' own key
    // counter.  Nothing in it outlives the batch.  NULL: slot kernel only (FLMR_S1_IMPL=slots)
    int32_t* fast_state;
    // which queries the dense stage-1 forms take (flmr_launch_s1_dense_modes' outputs), written by cand_plan_kernel when it runs -- one
    // launch less per batch; s1d_mode == NULL: not fused
    int32_t* s1d_mode; int32_t* s1d_scan_skip; int32_t* s1d_any; int32_t s1d_img_rows, s1d_exact_too;
    int32_t* row_ovf;   // [nqueries] out: 1 = the query has more surviving centroids than compact score rows (qmax): its stage 1 is
                        // recomputed from the fp16 centroids (flmr_launch_filter_stage1_recompute); NULL on the full-table path
};
int flmr_launch_candidates_chunked(const flmr_cand_args& a, hipStream_t st);
int flmr_launch_qualifying(const flmr_cand_args& a, hipStream_t st);   // list + ranks + compact score rows only
int flmr_launch_cand_emit_all(const flmr_cand_args& a, hipStream_t st);
int flmr_build_chunk_table(const int32_t* ivf_pids, const int64_t* ivf_offsets, int K, int64_t num_passages,
                           uint32_t** out_tab, int32_t* out_nchunks);
// passage bitmap of the union of the surviving centroids' IVF lists (+ per-query validity flag)
int flmr_launch_hit_bitmap(const uint32_t* idx_bits, int32_t idx_words, int32_t nqueries, const int32_t* ivf_pids,
                           const int64_t* ivf_offsets, const int32_t* cand_count, uint32_t* hit_bits, int64_t hit_words,
                           int32_t* hit_valid, hipStream_t st);
// stage 2: all centroids, one wave per document
int flmr_launch_filter_stage2(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride,
                              const int32_t* counts, int32_t max_count, uint64_t* keys, int64_t key_stride,
                              hipStream_t st);
// stage 2 without the score table: recompute the survivors' centroid scores from the fp16 centroids (bitwise = S0)
int flmr_launch_filter_stage2_mfma(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                   int32_t max_count, uint64_t* keys, int64_t key_stride, const _Float16* cen16,
                                   const _Float16* q_hi, const _Float16* q_lo, hipStream_t st);
// stage 2 as a query-stationary dense walk over the centroid table (flmr_stage2_walk.hip): same keys, bit for bit
int flmr_launch_filter_stage2_walk(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                   int32_t max_count, uint64_t* keys, int64_t key_stride, const _Float16* cen16,
                                   const _Float16* q_hi, const _Float16* q_lo, const int32_t* codes_sorted, hipStream_t st);
bool flmr_stage2_walk_pays(const flmr_index* ix, int nqueries, int max_count);
// stage 1 of the queries flagged in row_ovf (more surviving centroids than compact score rows) recomputed from the fp16 centroids
int flmr_launch_filter_stage1_recompute(const flmr_filter_args& f, const uint32_t* idx_bits, int32_t idx_words, const int32_t* cand,
                                        int64_t cand_stride, const int32_t* cand_count, const int32_t* row_ovf, uint64_t* keys,
                                        const _Float16* cen16, const _Float16* q_hi, const _Float16* q_lo, hipStream_t st);
// stage 2 with the centroid table cut into one L2-resident slice per XCD (flmr_stage2_xcd.hip): same keys, bit for bit;
// `part`: workspace of flmr_stage2_xcd_part_floats(max_queries, part_stride) floats
int flmr_launch_filter_stage2_xcd(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                  int32_t max_count, uint64_t* keys, int64_t key_stride, const flmr_index* ix,
                                  const _Float16* q_hi, const _Float16* q_lo, float* part, int64_t part_stride, hipStream_t st);
bool flmr_stage2_xcd_pays(const flmr_index* ix);
size_t flmr_stage2_xcd_part_floats(const flmr_index* ix, int64_t nqueries, int64_t ndocs);
// hi_only: the hi products alone (the fp16 numerics mode, where they ARE the scores; or the approximate pass of the
// approximate-then-refine form below)
int flmr_launch_filter_stage2_xcd_ex(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                     int32_t max_count, uint64_t* keys, int64_t key_stride, const flmr_index* ix,
                                     const _Float16* q_hi, const _Float16* q_lo, float* part, int64_t part_stride, bool hi_only,
                                     hipStream_t st);
// Stage-2 survivor selection from APPROXIMATE keys (hi-only scores, each within err_sum[q] of the full one): the passages
// certainly inside the top n go straight to out_pids, the passages within 2 err of the cut form the band whose full scores
// decide the rest (flmr_launch_s2_refine_finish after the band has been rescored).
int flmr_launch_s2_refine_plan(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t max_count, int32_t nqueries,
                               int32_t n, const float* err_sum, int32_t* out_pids, int64_t out_stride, int32_t* band_pids,
                               int64_t band_stride, int32_t* band_count, int32_t* need, int32_t* def_count, hipStream_t st);
int flmr_launch_s2_refine_finish(const uint64_t* band_keys, int64_t key_stride, const int32_t* band_count, const int32_t* need,
                                 const int32_t* def_count, int32_t max_count, int32_t nqueries, int32_t* out_pids,
                                 int64_t out_stride, int32_t* n_out, hipStream_t st);
// top-n of count[q] keys, unordered output (radix select); n_out[q] = min(n, count[q])
int flmr_launch_select_topn(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t nqueries,
                            int32_t n, int32_t* out_pids, int64_t out_stride, int32_t* n_out, hipStream_t st,
                            uint64_t* out_keys = nullptr, uint64_t key_add = 0);
// exact sharded protocol helpers (keys carry GLOBAL pids between ranks; 0 = empty)
int flmr_launch_select_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out, hipStream_t st);
int flmr_launch_sort_keys_topn(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out,
                               int32_t* out_counts, hipStream_t st);
int flmr_launch_filter_local_keys(const uint64_t* keys, int32_t nqueries, int32_t n_in, int64_t pid_base,
                                  int64_t num_passages, int32_t* out_pids, int64_t out_stride, int32_t* out_count,
                                  hipStream_t st, int32_t* out_slot = nullptr);
int flmr_launch_export_keys_slotted(const uint64_t* keys, int64_t key_stride, const int32_t* counts, const int32_t* slot,
                                    int32_t nqueries, uint64_t key_add, int32_t n, uint64_t* out, hipStream_t st);
int flmr_launch_export_keys(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t nqueries, uint64_t key_add,
                            int32_t n, uint64_t* out, hipStream_t st);
int flmr_launch_unpack_keys(const uint64_t* keys, int32_t nqueries, int32_t n, int32_t k, int32_t* out_pids, float* out_scores,
                            int32_t* out_counts, hipStream_t st);
// sorted (descending key) top-n of count[q] <= FLMR_MAX_NDOCS keys; writes pids (+ optional scores), n_out.
// pad_pid / pad_score fill positions [n_out, n) when fill != 0.
int flmr_launch_sort_topn(const uint64_t* keys, int64_t key_stride, const int32_t* counts, int32_t max_count,
                          int32_t nqueries, int32_t n, int32_t* out_pids, float* out_scores, int64_t out_stride,
                          int32_t* n_out, int64_t pid_base, int fill, hipStream_t st);

struct flmr_maxsim_args {
    const flmr_index* ix;
    const float* Q;           // [nqueries, nq, 128]
    const int32_t* q_lens;    // nullable
    int32_t nqueries, nq;
    const int32_t* pids;      // [nqueries, pid_stride]
    int64_t pid_stride;
    const int32_t* counts;    // [nqueries]
    int32_t max_count;
    uint64_t* keys;           // out [nqueries, key_stride] (score,pid) keys   (nullable)
    int64_t key_stride;
    float* scores;            // out [nqueries, key_stride] fp32 scores        (nullable)
    _Float16* q_hi;           // scratch [nqueries, round_up(nq,32), 128]: fp16 split of Q (nullable -> fp32 MFMA kernel)
    _Float16* q_lo;
    int32_t gpu_fp16;         // FLMR_NUMERICS_GPU_FP16: the reference's CUDA-path scoring (fp16 embeddings, -9999 padding, no clamp)
    int32_t q_split_done;     // 1: q_hi / q_lo already hold this batch's images (stage 0 made the same ones: nq <= 32 <= nq_cand)
    // workspace of the planned-tile kernels (nullable -> the kernels that walk the passages themselves):
    uint2* plan_desc;         // [nqueries, plan_stride] one descriptor per 32-token tile of a query's finalists
    int64_t plan_stride;      // >= max_count * ceil(max_doclen / 32)
    int32_t* plan_wbeg;       // [nqueries, plan_wcap] first tile of every wave's share (+ the end)
    int32_t plan_wcap;        // >= waves per query + 1
    float* colmax_ws;         // long queries: [nqueries, key_stride, round_up(nq, 32)] per-passage column maxima (nullable)
    int64_t colmax_cap;       // floats
};
int flmr_launch_maxsim(const flmr_maxsim_args& a, hipStream_t st);

// ---- stage 1 for dense survivor sets (flmr_stage1_dense.hip) ----
struct flmr_s1d_args {
    const int32_t* codes;        // `codes_sorted`: per-passage runs, distinct values first, FLMR_CODE_PAD ints readable behind the last
    const int64_t* offsets;      // [num_passages + 1]
    const uint16_t* ulen;        // [num_passages] distinct codes per passage (NULL: every token, offsets[p+1] - offsets[p])
    const uint32_t* idx_bits;    // [nqueries, idx_words] surviving-centroid mask (index_storage.py:116)
    const uint32_t* idx_prefix;  // [nqueries, idx_words] exclusive popcount per word (qualifying_kernel)
    int32_t idx_words;
    const float* rows;           // compact score rows [nqueries, row_cap, 32] (row = rank of the centroid among the survivors)
    int32_t row_cap;
    const int32_t* nqual;        // [nqueries] surviving centroids (clamped to row_cap by qualifying_kernel)
    const int32_t* q_lens;       // nullable
    int32_t nq_cand, nqueries;
    const int32_t* cand;         // [nqueries, cand_stride] ascending candidate pids
    int64_t cand_stride;
    const int32_t* cand_count;
    const int32_t* band;         // [nqueries, cand_stride] the band of an image query (EXACT pass)
    const int32_t* band_count;
    const int32_t* mode;         // [nqueries] FLMR_S1D_*: which queries this launch takes, from which list
    uint64_t* keys;              // [nqueries, cand_stride] out
    float* img_err;              // [nqueries] out (IMG pass): E + eps of the band rule (inf: the images cannot be used)
    int32_t parts;               // items per query (0: the launcher's choice)
    int32_t group;               // candidates per wave and group: 64, or 16 for short lists
    int64_t codes_len;           // tokens at `codes`; FLMR_CODE_PAD more ints must be readable behind them
    int32_t img_rows;            // rows of images the launch's LDS holds (IMG pass; set by the launcher)
    const int32_t* any;          // nullable: [0] some query takes the IMG pass, [1] some query takes the EXACT pass (flmr_launch_s1_dense_modes)
};
#define FLMR_S1D_SKIP 0    // stage 1 of the query is done elsewhere (list-scatter forms, the round-5 scan)
#define FLMR_S1D_IMAGE 1   // IMG pass over the candidates, band, EXACT pass over the band
#define FLMR_S1D_EXACT 2   // EXACT pass over the candidates
// `mean_codes` = the index's mean number of (distinct) codes per passage: 16 or 32 lanes per candidate x 4 or 8 codes per lane
int flmr_s1_dense_image_rows(int nqueries, int idx_words, double mean_codes);   // score-row images the LDS form holds per query (0: K too large)
int flmr_launch_s1_image(const flmr_s1d_args& a, double mean_codes, hipStream_t st);   // the approximate pass (U keys, img_err)
int flmr_launch_s1_exact(const flmr_s1d_args& a, double mean_codes, hipStream_t st);   // the exact pass (bands, whole lists)
// which queries the dense forms take: mode[q] = SKIP where skip[q] (done by a list-scatter form) or row_ovf[q] (no rows: the
// recompute form), IMAGE where the query's rows fit
// `img_rows` images, else EXACT (exact_too) or SKIP with scan[q] = 0 (the round-5 scan takes the query); scan[q] = 1 everywhere else
int flmr_launch_s1_dense_modes(const int32_t* skip, const int32_t* nqual, const int32_t* row_ovf, int32_t nqueries, int32_t img_rows,
                               int32_t exact_too, int32_t* mode, int32_t* scan_skip, int32_t* any, hipStream_t st);
// the band of every IMAGE query from its U keys (keys >= the n-th largest - err[q]) -> band pids / counts; in_count[q] = the number
// of keys the top-n selection after stage 1 reads for query q (band_count for IMAGE queries, counts[q] for the others)
int flmr_launch_s1_band(const uint64_t* keys, int64_t key_stride, const int32_t* counts, const int32_t* mode, const float* err,
                        int32_t nqueries, int32_t n, int32_t* band, int32_t* band_count, int32_t* in_count, hipStream_t st);

int flmr_launch_exclusive_scan_lengths(const int32_t* pids, const int64_t* doclens, const int64_t* offsets, int32_t n,
                                       int64_t* out_offsets /* [n+1] */, hipStream_t st);
