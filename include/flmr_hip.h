/*
 * flmr_hip.h -- C ABI of libflmr_hip.so: the MI355X (gfx950) late-interaction retrieval path.
 *
 * Drop-in boundary for the FLMR / ColBERTv2 search hot path of
 * LinWeizheDragon/Retrieval-Augmented-Visual-Question-Answering.  "TPC/" below abbreviates
 * third_party/ColBERT/colbert/ in the reference tree.  Each entry point names the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / C++ types; every function returns a flmr_status_t (0 = OK) and never throws;
 *     flmr_last_error() returns a thread-local message for the last non-OK status.
 *   - pointers are DEVICE pointers unless the parameter is documented as host; outputs are caller-
 *     allocated; all work is enqueued on `stream` (a hipStream_t cast to void*; NULL = default stream)
 *     and is asynchronous unless stated otherwise.
 *   - layouts are the reference's CPU layouts: codes i32[N], residuals
 *     u8[N, dim*nbits/8], doclens/offsets i64, row-major fp32 matrices.
 *   - numerics follow the reference's CPU path (fp32 everywhere, zero-clamped packed MaxSim,
 *     (score,pid)-lexicographic selection), not its fp16 CUDA path.
 *   - this build supports dim == 128 and nbits in {1,2,4,8}; other shapes return FLMR_ERR_UNSUPPORTED.
 */
#ifndef FLMR_HIP_H
#define FLMR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLMR_ABI_VERSION 5

typedef enum flmr_status {
    FLMR_OK = 0,
    FLMR_ERR_INVALID = 1,     /* bad argument */
    FLMR_ERR_UNSUPPORTED = 2, /* shape / option outside what this build implements */
    FLMR_ERR_HIP = 3,         /* a HIP runtime call failed (message in flmr_last_error) */
    FLMR_ERR_NOMEM = 4,
    FLMR_ERR_CAPACITY = 5     /* a workspace bound given at flmr_searcher_create was exceeded */
} flmr_status_t;

typedef struct flmr_index flmr_index_t;       /* opaque: index resident in HBM */
typedef struct flmr_searcher flmr_searcher_t; /* opaque: workspace + launch plan for batched search */
typedef void* flmr_stream_t;                  /* hipStream_t */

int flmr_abi_version(void);
const char* flmr_last_error(void);
/* number of visible HIP devices (0 and FLMR_ERR_HIP when the runtime reports none) */
int flmr_device_count(int* count);
/* Kernel-variant switches for A/B runs and cross-check tests (FLMR_S0_IMPL, FLMR_FULL_TABLE, FLMR_CAND_IMPL,
 * FLMR_S1_NO_HITMAP, FLMR_S1_IMPL, FLMR_S2_IMPL, FLMR_S0_STAGED, FLMR_S3_NO_MULTIQ, FLMR_S3_IMPL, FLMR_SCORE_IMPL).  The
 * environment variables of the same names are read ONCE per process, on first use of the library; afterwards the table
 * changes only through this call (value NULL or "" clears a switch).  Op-level entry points read the table when they are
 * called; a searcher snapshots it at flmr_searcher_create and keeps that snapshot.  Nothing on the per-batch launch path
 * reads the environment.  (The reference has no counterpart: its extensions have one implementation each.)
 * Values with a measured use: FLMR_S2_IMPL = xcd | lds | ldsb | walk | regs (stage 2: one L2-resident table slice per XCD --
 * the default for whole batches when the fp16 centroid table exceeds an L2 -- / row gather with 4-wave or 16-wave blocks /
 * dense table walk / register gather); FLMR_S3_IMPL = lean | cw | cwregs | regs | dma | f32 | qs (fused MaxSim.  Nq <= 32: lean, the
 * default = the centroid + weight arithmetic on planned tiles; cw / cwregs = the same arithmetic walking the passages, rows by
 * LDS-DMA / register gathers; regs / dma = decompress-normalise-split; f32 = fp32-MFMA kernel.  Longer queries: qs, the
 * query-stationary kernel, the default from 288 rows; any other value = the chunked kernel); FLMR_S1_IMPL = scan | slots
 * (stage 1: the code-scanning kernel for every query / the slot form of the list-scatter kernel for every query; default: its
 * queue form first, the slot form for the queries that one hands over, and beyond the list-scatter forms' limits -- more than 1024
 * surviving centroids, lists far longer than the probed cells' -- the dense forms of flmr_stage1_dense.hip: fp16 images of the
 * query's score rows in LDS, upper-bound keys for every candidate, the band around the cut rescored exactly; the exact form alone
 * when the rows do not fit).
 * Variants of one arithmetic are bit-identical to each other, the arithmetics agree to fp32 roundoff (tests/test_hip_parity.py).
 * One switch is a capacity, not a variant: FLMR_ROW_CAP = score rows a searcher keeps per query (64 .. 65535, default 16384,
 * never more than K).  The default path stores the centroid scores of a query only for the centroids that pass
 * centroid_score_threshold (the rows index_storage.py:116's `idx` selects; 128 bytes each) instead of the reference's
 * K x nq_cand table per query (16.8 MB at K = 131072).  A query with more surviving centroids than the capacity is NOT an error
 * (the reference has no such limit): its stage 1 is recomputed from the fp16 centroids inside the same batch (slower for that
 * query only; FLMR_TAP_STAGE1_FORM reads 7 for it). */
int flmr_set_option(const char* name, const char* value);

/* ------------------------------------------------------------------------------------------------
 * Index residency.  Replaces IndexLoader / ResidualCodec.load / ResidualEmbeddings.load_chunks /
 * ResidualEmbeddingsStrided (TPC/search/index_loader.py:14-86, TPC/indexing/codecs/residual.py:134-150,
 * residual_embeddings.py:27-52, residual_embeddings_strided.py:13-21): the host side parses the on-disk
 * files, this call moves the arrays into HBM once.
 * ---------------------------------------------------------------------------------------------- */
#define FLMR_MEM_HOST 0   /* desc pointers are host memory: the library allocates HBM and copies (synchronous) */
#define FLMR_MEM_DEVICE 1 /* desc pointers are device memory owned by the caller and must outlive the index */

typedef struct flmr_index_desc {
    int32_t dim;               /* 128 */
    int32_t nbits;             /* 1, 2, 4 or 8 */
    int32_t num_centroids;     /* K */
    int32_t memory;            /* FLMR_MEM_HOST | FLMR_MEM_DEVICE */
    int64_t num_embeddings;    /* N */
    int64_t num_passages;      /* passages in this shard */
    int64_t pid_base;          /* global pid of local passage 0 (sharded indexes); 0 otherwise */
    const int32_t* codes;      /* [N] centroid id per token                   ({i}.codes.pt) */
    const uint8_t* residuals;  /* [N, dim*nbits/8] packed bucket indices      ({i}.residuals.pt) */
    const int64_t* doc_offsets;/* [num_passages+1] token offset per passage   (cumsum of doclens.{i}.json) */
    const int32_t* ivf_pids;   /* [ivf_offsets[K]] sorted unique LOCAL pids per centroid (ivf.pid.pt) */
    const int64_t* ivf_offsets;/* [K+1] */
    const float* centroids;    /* [K, dim] fp32 (fp16 file values widened, residual.py:29) (centroids.pt) */
    const float* bucket_weights; /* [2^nbits] (buckets.pt) -- ALWAYS a host pointer */
} flmr_index_desc_t;

int flmr_index_open(const flmr_index_desc_t* desc, flmr_index_t** out_index);
int flmr_index_close(flmr_index_t* index);

/* What flmr_index_open derived for this index on this device (the reference has no counterpart: its loader keeps the files'
 * arrays only).  derived_bytes = HBM held by structures built at open on top of the caller's arrays. */
typedef struct flmr_index_info {
    int64_t derived_bytes;
    int64_t max_doclen;
    int32_t centroids_f16_exact;  /* 1: every centroid is fp16-representable -> fp16-split MFMA kernels (always true for reference-format indexes) */
    int32_t stage2_slices;        /* slices the fp16 centroid table is cut into for the XCD-sliced stage 2 (8/16/24/32) */
    int32_t xcd_round_robin;      /* 1: probe at open saw 8 XCDs and workgroup L of a 1-D grid on XCD (L % 8) */
    int32_t stage2_sliced;        /* 1: whole-batch stage 2 takes the XCD-sliced kernel on this index */
    int32_t passage_chunks;       /* 32768-passage chunks of the candidate stage */
    int32_t duplicate_permille;   /* tokens that repeat a code of their own passage, per thousand: what the distinct-code runs built at
                                   * open save stage 2 and the dense stage 1 (0 when the sorted code copy was not built; below 100 the
                                   * runs are not kept) */
} flmr_index_info_t;
int flmr_index_info(const flmr_index_t* index, flmr_index_info_t* out_info);

/* ------------------------------------------------------------------------------------------------
 * Batched search.  Replaces the per-query loop Searcher._search_all_Q -> dense_search ->
 * IndexScorer.rank (TPC/searcher.py:73-132, TPC/search/index_storage.py:67-182) for a whole batch
 * of queries: S0 centroid scores + cell probe + IVF union, S1/S2 centroid-only pruning, S3 fused
 * residual decompression + L2 normalisation + MaxSim, S4 top-k.
 * ---------------------------------------------------------------------------------------------- */
typedef struct flmr_search_params {
    int32_t k;                        /* results per query = width of the caller's output rows (>= 1; NOT a workspace bound: rows are
                                       * padded past the ndocs/4 finalists); policy defaults are the host's job (searcher.py:92-118) */
    int32_t ncells;                   /* 1..8 */
    float centroid_score_threshold;   /* idx = max_j score >= thr (index_storage.py:116) */
    int32_t ndocs;                    /* S1 keeps ndocs, S2 keeps ndocs/4 (filter_pids.cpp:126-164); 4..8192 */
    int32_t nq_cand;                  /* config.query_maxlen: candidate generation uses Q[:, :nq_cand] (index_storage.py:77); <=128 */
} flmr_search_params_t;

/* max_queries / max_nq / max_params bound the workspace (allocated here, synchronous). */
int flmr_searcher_create(const flmr_index_t* index, int32_t max_queries, int32_t max_nq,
                         const flmr_search_params_t* max_params, flmr_searcher_t** out_searcher);
int flmr_searcher_destroy(flmr_searcher_t* searcher);
/* bytes of HBM held by the searcher's workspace */
int flmr_searcher_workspace_bytes(const flmr_searcher_t* searcher, int64_t* bytes);
/* Deferred device-side errors.  Two conditions can only be detected on the device: a batch producing more candidates than
 * the workspace bound (the reference's counterpart is an `assert` inside filter_pids.cpp / an out-of-range write), and
 * q_lens entries outside [0, nq] (they are clamped before any kernel reads them).  Both raise a flag that is copied to the
 * host asynchronously after the batch; the NEXT call on the searcher returns FLMR_ERR_CAPACITY / FLMR_ERR_INVALID and
 * clears it.  flmr_searcher_check waits for the searcher's last batch and reports at once (synchronous). */
int flmr_searcher_check(flmr_searcher_t* searcher);
/* The same flags for a caller that pipelines sub-batches (device batch i+1 beside the host's reading of batch i): enqueues, on
 * `stream`, a copy of the searcher's four device status words -- [0] candidate bound exceeded, [1] q_lens outside [0, nq], [2], [3]
 * reserved (always 0) -- to `host_flags` (pinned host memory, 4 x int32).  Once an event recorded
 * after this call has completed the words are those of every batch issued on the searcher up to here (they are sticky until
 * flmr_searcher_check or a later call reports them and clears them).  Asynchronous; nothing is reported or cleared by this call.
 * Reference counterpart: none (Searcher._search_all_Q is synchronous per query, TPC/searcher.py:73-89). */
int flmr_searcher_status_async(flmr_searcher_t* searcher, int32_t* host_flags, flmr_stream_t stream);

/* Q [nqueries, nq, dim] fp32.  q_lens (nullable) i32[nqueries]: number of valid leading rows per query
 * (rows removed by remove_zero_tensors, searcher.py:120-126, are compacted away by the host).
 * out_pids i32[nqueries, k] (global pids, -1 padded), out_scores f32[nqueries, k] (0 padded),
 * out_counts i32[nqueries] = number of valid results (<= k; fewer when fewer candidates survive:
 * the build returns every surviving candidate once where the reference has UB). */
int flmr_search_batch(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries,
                      int32_t nq, const flmr_search_params_t* params, int32_t* out_pids, float* out_scores,
                      int32_t* out_counts, flmr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Exact passage-sharded search (SURVEY 8e "exact-parity mode"; the reference has no multi-GPU search at all,
 * src/executors/FLMR_executor.py:778-783).  Each rank owns a passage shard (pid_base) and runs three phases with
 * one small all-gather of keys after each; a key is u64 = order-preserving(score) << 32 | GLOBAL pid, 0 = empty.
 *   phase1: S0..S1 locally            -> out_keys [nqueries, ndocs]    (the shard's top-ndocs stage-1 keys)
 *   gather + flmr_topn_keys(n = ndocs)      = the global stage-1 survivors, identical to the single-index ones
 *   phase2: S2 of the shard's members -> out_keys [nqueries, ndocs], SLOT-ALIGNED with the global list given (0 where the
 *           passage lives on another shard): combine the shards by a SUM all-reduce (or a gather)
 *   gather + flmr_topn_keys(n = ndocs/4)    = the global stage-2 survivors
 *   phase3: S3 of the shard's members -> out_keys [nqueries, ndocs/4], slot-aligned likewise
 *   gather + flmr_topn_keys(n = k) + flmr_unpack_keys = the final ranking, bit-identical to flmr_search_batch on the
 *   unsharded index.  The same Q / q_lens / params must be passed to all three phases; params->ndocs must equal the
 *   ndocs the searcher was created with.
 * ---------------------------------------------------------------------------------------------- */
int flmr_search_phase1(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                       const flmr_search_params_t* params, uint64_t* out_keys, flmr_stream_t stream);
/* Query-split stage 0 (optional phase 0).  Stage 0 (centroids x Q', cells, idx -- colbert/search/candidate_generation.py:10-37,
 * index_storage.py:114-118) does not depend on the passage shard, so instead of every rank repeating it for the whole
 * batch, rank r runs flmr_search_probe on the query slice [q_begin, q_begin + q_count), the ranks all-gather the three
 * small outputs, and flmr_search_phase1_probed (instead of flmr_search_phase1) continues from them; it rebuilds the
 * score-table rows of the qualifying centroids with the identical MFMA sequence, so results stay bit-identical.
 *   out_idx_bits [q_count, idx_words] u32, out_cells [q_count, max_cells] i32, out_ncell [q_count] i32 (device);
 *   idx_words / max_cells from flmr_searcher_probe_dims.  FLMR_ERR_UNSUPPORTED when the searcher is not on the
 *   sparse-table path (centroids not fp16-exact, K % 64 != 0, or nq_cand > 32): use flmr_search_phase1 then. */
int flmr_searcher_probe_dims(const flmr_searcher_t* searcher, int32_t* idx_words, int32_t* max_cells);
/* *supported = 1 iff flmr_search_probe / flmr_search_phase1_probed run for batches of `nq` query tokens with `params` on
 * this searcher.  The answer depends only on replicated data (centroids, shapes, switches), so every rank of a sharded
 * job gets the same one: use it to choose between the query-split and the replicated stage 0. */
int flmr_searcher_probe_supported(const flmr_searcher_t* searcher, int32_t nq, const flmr_search_params_t* params,
                                  int32_t* supported);
int flmr_search_probe(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                      const flmr_search_params_t* params, int32_t q_begin, int32_t q_count, uint32_t* out_idx_bits,
                      int32_t* out_cells, int32_t* out_ncell, flmr_stream_t stream);
int flmr_search_phase1_probed(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                              const flmr_search_params_t* params, const uint32_t* idx_bits, const int32_t* cells,
                              const int32_t* ncell, uint64_t* out_keys, flmr_stream_t stream);
int flmr_search_phase2(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                       const flmr_search_params_t* params, const uint64_t* global_s1, int32_t n_in, uint64_t* out_keys,
                       flmr_stream_t stream);
int flmr_search_phase3(flmr_searcher_t* searcher, const float* Q, const int32_t* q_lens, int32_t nqueries, int32_t nq,
                       const flmr_search_params_t* params, const uint64_t* global_s2, int32_t n_in, uint64_t* out_keys,
                       flmr_stream_t stream);
/* keys [nqueries, m] (m <= 8192) -> the n largest in descending order, 0 padded; out_counts (nullable) = #non-empty */
int flmr_topn_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out_keys, int32_t* out_counts,
                   flmr_stream_t stream);
/* keys [nqueries, m] (any m) -> the n largest of each row, UNORDERED, 0 padded: for the exchange steps whose consumer
 * (flmr_search_phase2 / phase3) only filters the survivors by shard */
int flmr_select_keys(const uint64_t* keys, int32_t nqueries, int32_t m, int32_t n, uint64_t* out_keys, flmr_stream_t stream);
/* descending keys [nqueries, n] -> out_pids i32 / out_scores f32 [nqueries, k] (-1 / 0 padded), out_counts i32[nqueries] */
int flmr_unpack_keys(const uint64_t* keys, int32_t nqueries, int32_t n, int32_t k, int32_t* out_pids, float* out_scores,
                     int32_t* out_counts, flmr_stream_t stream);

/* Stage taps for parity tests: copy an internal per-query buffer of the LAST flmr_search_batch call to
 * HOST memory (synchronises the stream).  `host_out` capacity in elements; *count receives the number
 * of valid elements.  Element types: CENTROID_SCORES f32 [K, ncol] (ncol = nq_cand rounded up to 32),
 * IDX_BITS u32 [ceil(K/32)], CELLS i32, CANDIDATES i32 (ascending local pids), STAGE1 i32 (unordered),
 * STAGE2 i32 (descending (score,pid) order = filter_pids output), DOC_SCORES f32 (aligned with STAGE2),
 * Q_ERR f32 [32] / Q_ERR_SUM f32 [1]: the per-column and per-passage bounds the "hi first" stages 0 / 2 decide with
 * (0 elements when the last batch did not run that path); STAGE1_FORM i32 [1]: which form of the list-scatter stage 1 produced
 * the query's keys -- 0 the queue form, 1 the slot form (query not tried by the others: too many surviving lists), 2 the slot form
 * after the queue form gave the query up, 3 the small-dense form (a searcher whose queries mostly overflow the queue), 4 the slot
 * form after the small-dense form gave the query up, 5 the dense image form (fp16 images of the score rows in LDS, the band around the cut
 * rescored exactly), 6 the dense exact form, 7 the recompute form (a query with more surviving centroids than the searcher keeps score
 * rows for: FLMR_ROW_CAP) (0 elements: stage 1 ran in another mode). */
typedef enum flmr_tap {
    FLMR_TAP_CENTROID_SCORES = 0,
    FLMR_TAP_IDX_BITS = 1,
    FLMR_TAP_CELLS = 2,
    FLMR_TAP_CANDIDATES = 3,
    FLMR_TAP_STAGE1 = 4,
    FLMR_TAP_STAGE2 = 5,
    FLMR_TAP_DOC_SCORES = 6,
    FLMR_TAP_Q_ERR = 7,
    FLMR_TAP_Q_ERR_SUM = 8,
    FLMR_TAP_STAGE1_FORM = 9
} flmr_tap_t;
int flmr_searcher_tap(flmr_searcher_t* searcher, int32_t what, int32_t query, void* host_out, int64_t capacity,
                      int64_t* count);

/* By default the per-query centroid-score table is kept SPARSE (only the rows of centroids that pass the threshold are
 * stored; stage 2 and the cell probe recompute what they need).  Enable the full table before a batch whose
 * FLMR_TAP_CENTROID_SCORES tap will be read (what IndexScorer.retrieve() returns, index_storage.py:67-80). */
int flmr_searcher_set_full_table(flmr_searcher_t* searcher, int32_t enable);

/* Numerics mode.  FLMR_NUMERICS_CPU (default) = the reference's CPU path, the one the golden vectors pin.
 * FLMR_NUMERICS_GPU_FP16 = the reference's CUDA path (use_gpu=True: TPC/search/index_storage.py:113-149,157-158,
 * TPC/search/candidate_generation.py:50-52, TPC/indexing/codecs/residual.py:242-278 + decompress_residuals.cu,
 * TPC/modeling/colbert.py:235-263,289-311), what the reference runs when config.total_visible_gpus > 0
 * (src/executors/FLMR_executor.py:784):
 *   - Q rounded to fp16; centroid scores = fp16 x fp16 products accumulated in fp32, rounded to fp16;
 *   - idx = half(score) >= half(thr);
 *   - stage 1 / 2: per-column maxima of the fp16 scores, padding -9999 stored in fp16 (-10000), the column sum accumulated
 *     in fp32 and rounded to fp16 (a passage without a qualifying code scores -inf); top-ndocs, then top-ndocs/4;
 *   - stage 3: embeddings = half(centroid + half(bucket weight)), L2-normalised with an fp16 norm into fp16, scores =
 *     fp16 x fp16 products accumulated in fp32 and rounded to fp16, max over the passage's REAL tokens only (no zero clamp),
 *     summed in fp32 and rounded to fp16.
 * Where torch leaves an order open (topk / sort among equal fp16 scores, GEMM summation order) this build breaks ties by
 * (score, pid) descending and accumulates k-ascending inside the MFMA's blocks.  Parity of this mode is checked against the
 * reference's torch expressions evaluated on CPU half tensors (tests/golden/make_golden_gpu_numerics.py); the reference's
 * CUDA kernels themselves cannot run in the build environment, so where they decide (decompress_residuals.cu, the CUDA
 * GEMMs) parity is UNPINNED.  FLMR_ERR_UNSUPPORTED unless the centroids are fp16-representable and K % 64 == 0.  The mode
 * applies to flmr_search_batch and the phased protocol alike; the walk variant of stage 2 is not used in it. */
#define FLMR_NUMERICS_CPU 0
#define FLMR_NUMERICS_GPU_FP16 1
int flmr_searcher_set_numerics(flmr_searcher_t* searcher, int32_t mode);

/* Timing taps: per-stage HIP-event milliseconds, SUMMED over the flmr_search_batch calls made in profiling mode
 * (flmr_searcher_set_profiling(s, 1)) since the previous read; reading waits for those calls and clears the sums, so a
 * caller that splits a batch into sub-batches reads the whole batch's stage times once.  ms[FLMR_NUM_STAGES] is HOST
 * memory.  (Event sets are kept in a ring of 8 calls; a 9th unread call first waits for the oldest.) */
#define FLMR_NUM_STAGES 9
int flmr_searcher_set_profiling(flmr_searcher_t* searcher, int32_t enable);
int flmr_searcher_stage_ms(flmr_searcher_t* searcher, float* ms_host);
const char* flmr_stage_name(int32_t stage);

/* ------------------------------------------------------------------------------------------------
 * Op-level entry points: 1:1 with the reference's four pybind functions + the scoring head.
 * ---------------------------------------------------------------------------------------------- */

/* filter_pids_cpp (TPC/search/filter_pids.cpp:126-164).  centroid_scores f32[K, nq] row-major, idx u8[K]
 * (bool), doclens/offsets i64 indexed by pid.  out_pids i32[ndocs/4] in descending (score,pid) order,
 * *out_count (device i32) = number written.  npids < ndocs: keeps everything once (reference: UB). */
int flmr_filter_pids(const int32_t* pids, int64_t npids, const float* centroid_scores, int32_t K, int32_t nq,
                     const int32_t* codes, const int64_t* doclens, const int64_t* offsets, const uint8_t* idx,
                     int32_t ndocs, int32_t* out_pids, int32_t* out_count, flmr_stream_t stream);

/* decompress_residuals_cpp (TPC/search/decompress_residuals.cpp:80-155; CUDA twin
 * TPC/indexing/codecs/decompress_residuals.cu:8-40).  out f32[sum doclens[pids], dim] packed in pid
 * order; out_row_offsets i64[npids+1] (device, nullable) receives the exclusive prefix of the doc lengths.
 * out_capacity_rows bounds `out`. */
int flmr_decompress_residuals(const int32_t* pids, int32_t npids, const int64_t* doclens, const int64_t* offsets,
                              const float* bucket_weights, const uint8_t* reversed_bit_map,
                              const uint8_t* bucket_weight_combinations, const uint8_t* binary_residuals,
                              const int32_t* codes, const float* centroids, int32_t dim, int32_t nbits,
                              float* out, int64_t out_capacity_rows, int64_t* out_row_offsets, flmr_stream_t stream);

/* segmented_lookup_cpp (TPC/search/segmented_lookup.cpp:127-144): ragged gather of rows of `row_bytes`
 * bytes.  lengths/offsets i64[nseg] are per segment (already indexed by pid, as StridedTensor._prepare_lookup
 * does, TPC/search/strided_tensor.py:59-75).  out_row_offsets i64[nseg+1] device scratch/output. */
int flmr_segmented_lookup(const void* input, int64_t row_bytes, const int64_t* lengths, const int64_t* offsets,
                          int32_t nseg, void* out, int64_t out_capacity_rows, int64_t* out_row_offsets,
                          flmr_stream_t stream);

/* segmented_maxsim_cpp (TPC/modeling/segmented_maxsim.cpp:49-93): scores f32[ntok, nq], lengths i64[ndocs]
 * -> out f32[ndocs]; running max starts at 0 (zero clamp), then sum over nq. */
int flmr_segmented_maxsim(const float* scores, const int64_t* lengths, int32_t ndocs, int32_t nq, float* out,
                          flmr_stream_t stream);

/* colbert_score_packed fused with decompression + F.normalize (TPC/search/index_storage.py:160-177,
 * TPC/modeling/colbert.py:289-311): exact late-interaction score of Q [nq, dim] against `npids` local pids
 * of `index`, without materialising D in HBM.  out f32[npids]. */
int flmr_score_pids(const flmr_index_t* index, const float* Q, int32_t nq, const int32_t* pids, int32_t npids,
                    float* out, flmr_stream_t stream);

/* colbert_score / colbert_score_reduce, padded variant (TPC/modeling/colbert.py:235-286; callers
 * src/models/retriever/FLMR.py score(), src/executors/FLMR_executor.py:799-847 exhaustive search).
 * Q f32[q_batch, nq, dim] with q_batch in {1, B}; D f32[B, Ld, dim]; mask u8[B, Ld]; out f32[B].
 * Padded tokens score -9999 (no zero clamp); forward only. */
int flmr_colbert_score_padded(const float* Q, int32_t q_batch, int32_t nq, const float* D, const uint8_t* mask,
                              int32_t B, int32_t Ld, int32_t dim, float* out, flmr_stream_t stream);
/* The cross form: Q f32[nqueries, nq, dim] holds independent queries and EVERY one is scored against every document:
 * out f32[nqueries, B] = the rate matrix of the executor's exhaustive search (src/executors/FLMR_executor.py:799-847, which fills
 * it four documents at a time through model.score) in one launch per document chunk.  nqueries <= 65535. */
int flmr_colbert_score_cross(const float* Q, int32_t nqueries, int32_t nq, const float* D, const uint8_t* mask,
                             int32_t B, int32_t Ld, int32_t dim, float* out, flmr_stream_t stream);
/* The same kernel, stopped one step earlier: out_colmax f32 [B, nq] = the per-column maxima (-9999 padding) whose sum
 * flmr_colbert_score_padded returns -- what colbert_score_reduce's 'flipr' interaction reduces differently (the 32 largest of
 * the first 64 columns + the 8 largest of the rest, TPC/modeling/colbert.py:246-261). */
int flmr_colbert_colmax_padded(const float* Q, int32_t q_batch, int32_t nq, const float* D, const uint8_t* mask, int32_t B,
                               int32_t Ld, int32_t dim, float* out_colmax, flmr_stream_t stream);

/* Merge per-shard top-k lists after the RCCL all-gather: scores f32[nshards, nqueries, k],
 * pids i32[nshards, nqueries, k] (-1 = empty) -> global top-k per query in descending (score,pid) order. */
int flmr_merge_topk(const float* scores, const int32_t* pids, int32_t nshards, int32_t nqueries, int32_t k,
                    float* out_scores, int32_t* out_pids, int32_t* out_counts, flmr_stream_t stream);

/* Exchange steps over RCCL for a caller without torch.  `comm` is an ncclComm_t the CALLER created
 * (one rank per GPU); the collectives are enqueued on `stream`.  libflmr_hip.so does not link librccl: the symbols are
 * resolved on first use from the RCCL instance already loaded in the process (FLMR_ERR_UNSUPPORTED when there is none).
 *   flmr_topk_allgather     fast mode: all-gather of every rank's flmr_search_batch output (scores f32 / GLOBAL pids i32
 *                           [nqueries, k]) into the caller's [nranks, nqueries, k] workspaces, then flmr_merge_topk.
 *   flmr_keys_allgather     exact mode after phase 1: keys [count] per rank -> out [nranks, count] (then flmr_select_keys on
 *                           the [nqueries, nranks * ndocs] view; the permutation of ranks inside a row does not matter).
 *   flmr_keys_allreduce_sum exact mode after phases 2 / 3: slot-aligned keys, one non-zero contributor per slot, summed in place. */
int flmr_topk_allgather(void* comm, int32_t nranks, const float* scores, const int32_t* pids, int32_t nqueries, int32_t k,
                        float* gathered_scores, int32_t* gathered_pids, float* out_scores, int32_t* out_pids,
                        int32_t* out_counts, flmr_stream_t stream);
int flmr_keys_allgather(void* comm, int32_t nranks, const uint64_t* keys, int64_t count, uint64_t* out, flmr_stream_t stream);
int flmr_keys_allreduce_sum(void* comm, uint64_t* keys, int64_t count, flmr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Index build ops: ResidualCodec.compress (colbert/indexing/codecs/residual.py:169-222).  All pointers are
 * DEVICE pointers, dim == 128.
 *   flmr_nearest_centroids  out_codes[t] = argmax_c centroids[c] . emb[t] (lowest c on ties), residual.py:206-216.
 *                           centroids f32 [K,128] must be fp16-representable (the index stores them as half,
 *                           residual.py:161) -> FLMR_ERR_UNSUPPORTED otherwise.  Synchronises `stream` before returning.
 *   flmr_compress_residuals out_residuals u8 [n, 16*nbits]: bucketize(emb - centroids[codes], bucket_cutoffs[2^nbits - 1])
 *                           then the reference's bit order (each nbits group LSB-first, bytes packed MSB-first),
 *                           residual.py:186-204 (binarize) -- the layout flmr_index_open / decompress_residuals.cpp read.
 * ---------------------------------------------------------------------------------------------- */
int flmr_nearest_centroids(const float* centroids, int64_t K, const float* emb, int64_t n, int32_t* out_codes,
                           flmr_stream_t stream);
int flmr_compress_residuals(const float* centroids, int64_t K, const float* emb, const int32_t* codes, int64_t n,
                            const float* bucket_cutoffs, int32_t nbits, uint8_t* out_residuals, flmr_stream_t stream);
/*   flmr_build_ivf          the inverted file of an index: for every centroid the ascending unique pids of the passages that
 *                           hold a token assigned to it -- optimize_ivf, TPC/indexing/utils.py:8-53 (codes.sort() of
 *                           collection_indexer.py:388-426, embedding id -> pid, unique per centroid).  codes i32 [n_tokens] in
 *                           passage order, doc_offsets i64 [num_passages + 1]; ivf_pids must hold n_tokens entries (the
 *                           bound), ivf_lengths i64 [K]; *total (HOST) = entries written.  Synchronises `stream`. */
int flmr_build_ivf(const int32_t* codes, int64_t n_tokens, const int64_t* doc_offsets, int64_t num_passages, int32_t K,
                   int32_t* ivf_pids, int64_t* ivf_lengths, int64_t* total, flmr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLMR_HIP_H */
