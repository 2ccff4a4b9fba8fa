/*
 * flmr_oracle.c -- CPU restatement of the reference's late-interaction search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product path (retrieval-augmented-visual-question-answering_amd/)
 * never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced by running the reference's own CPU path
 * (tests/golden/make_golden.py, reference = third_party/ColBERT/colbert, "TPC/" below),
 * and oracle/_ref (the reference's four C++ files compiled in place) cross-checks it.
 *
 * Plain C99, scalar, sequential accumulation orders exactly as the reference's C++ where the
 * reference defines one; where the reference delegates to BLAS / torch reductions (order not
 * defined) the restatement uses a k-ascending fp32 loop and the tests carry a tolerance.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so no FMA contraction changes bits).
 */
#include <omp.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * S0a  centroid scores  (TPC/search/candidate_generation.py:13  `centroids @ Q.T`)
 * out[K, nq] row-major, fp32.  Reference = BLAS sgemm (order undefined); here k-ascending.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_centroid_scores(const float* centroids, const float* Q, int K, int nq, int dim, float* out) {
    for (int c = 0; c < K; c++) {
        const float* cr = centroids + (size_t)c * dim;
        for (int j = 0; j < nq; j++) {
            const float* qr = Q + (size_t)j * dim;
            float acc = 0.0f;
            for (int k = 0; k < dim; k++) acc += cr[k] * qr[k];
            out[(size_t)c * nq + j] = acc;
        }
    }
}

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------------------------------
 * S0b  probed cells  (candidate_generation.py:12-20): per query token the `ncells` best
 * centroids (topk over dim 0, or argmax when ncells==1), flattened and unique'd.
 * Ties at the cut are not defined by torch.topk; this restatement prefers the lower index.
 * Returns the number of unique cells; out_cells (capacity nq*ncells) ascending.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_select_cells(const float* scores, int K, int nq, int ncells, int32_t* out_cells) {
    int n = 0;
    float* bv = (float*)malloc(sizeof(float) * (size_t)ncells);
    int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)ncells);
    for (int j = 0; j < nq; j++) {
        int have = 0;
        for (int c = 0; c < K; c++) {
            float v = scores[(size_t)c * nq + j];
            /* insertion into a descending list; strict '>' keeps the earlier (lower) index on ties */
            int pos = have;
            while (pos > 0 && v > bv[pos - 1]) pos--;
            if (pos >= ncells) continue;
            int last = have < ncells ? have : ncells - 1;
            for (int t = last; t > pos; t--) { bv[t] = bv[t - 1]; bi[t] = bi[t - 1]; }
            bv[pos] = v; bi[pos] = c;
            if (have < ncells) have++;
        }
        for (int t = 0; t < have; t++) out_cells[n++] = bi[t];
    }
    free(bv); free(bi);
    qsort(out_cells, (size_t)n, sizeof(int32_t), cmp_i32);
    int u = 0;
    for (int i = 0; i < n; i++)
        if (u == 0 || out_cells[u - 1] != out_cells[i]) out_cells[u++] = out_cells[i];
    return u;
}

/* ------------------------------------------------------------------------------------------
 * S0c  candidate pids  (candidate_generation.py:31-37,57-60; segmented_lookup.cpp:46-47):
 * concatenate the IVF pid lists of the probed cells, sort, unique_consecutive.
 * Returns P; out_pids ascending (capacity num_passages).
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_candidates(const int32_t* cells, int ncell, const int32_t* ivf, const int64_t* ivf_offsets,
                               int64_t num_passages, int32_t* out_pids) {
    uint8_t* seen = (uint8_t*)calloc((size_t)num_passages, 1);
    for (int i = 0; i < ncell; i++)
        for (int64_t e = ivf_offsets[cells[i]]; e < ivf_offsets[cells[i] + 1]; e++) seen[ivf[e]] = 1;
    int64_t P = 0;
    for (int64_t p = 0; p < num_passages; p++)
        if (seen[p]) out_pids[P++] = (int32_t)p;
    free(seen);
    return P;
}

/* idx[c] = max_j scores[c,j] >= thr   (TPC/search/index_storage.py:116; '>=' in fp32) */
ORC_API void orc_idx_mask(const float* scores, int K, int nq, float thr, uint8_t* idx) {
    for (int c = 0; c < K; c++) {
        float m = scores[(size_t)c * nq];
        for (int j = 1; j < nq; j++) m = fmaxf(m, scores[(size_t)c * nq + j]);
        idx[c] = (m >= thr) ? 1 : 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * S1 / S2  centroid-only MaxSim pruning   (TPC/search/filter_pids.cpp)
 *   per doc (:27-69): per_tok[k] = max over tokens whose code has idx[code] of cs[code,k],
 *   initial -9999; score = sequential sum over k (:59-63).
 *   selection (:24,:64,:108-123): std::priority_queue<pair<float,int>> => descending by
 *   (score, pid) lexicographic; output in that order.
 * orc_filter_pass = one pass (filter_pids_helper); keeps min(n_keep, npids) docs.  The
 * reference pops an empty heap when npids < n_keep (undefined behaviour); the
 * build DEFINES that case as "keep all, no duplicates".
 * ---------------------------------------------------------------------------------------- */
typedef struct { float s; int32_t p; } orc_sp_t;

static int cmp_sp_desc(const void* a, const void* b) {
    const orc_sp_t* x = (const orc_sp_t*)a; const orc_sp_t* y = (const orc_sp_t*)b;
    if (x->s != y->s) return (x->s < y->s) ? 1 : -1;
    if (x->p != y->p) return (x->p < y->p) ? 1 : -1;
    return 0;
}

ORC_API int orc_filter_pass(const int32_t* pids, int64_t npids, const float* cs, int nq, const int32_t* codes,
                            const int64_t* doclens, const int64_t* offsets, const uint8_t* idx, int n_keep,
                            int32_t* out_pids, float* out_scores) {
    orc_sp_t* all = (orc_sp_t*)malloc(sizeof(orc_sp_t) * (size_t)(npids > 0 ? npids : 1));
    float* per = (float*)malloc(sizeof(float) * (size_t)nq);
    for (int64_t i = 0; i < npids; i++) {
        int32_t pid = pids[i];
        for (int k = 0; k < nq; k++) per[k] = -9999.0f;
        for (int64_t j = 0; j < doclens[pid]; j++) {
            int32_t code = codes[offsets[pid] + j];
            if (idx == NULL || idx[code]) {
                const float* row = cs + (size_t)code * nq;
                for (int k = 0; k < nq; k++) per[k] = per[k] > row[k] ? per[k] : row[k];
            }
        }
        float score = 0.0f;
        for (int k = 0; k < nq; k++) score += per[k];
        all[i].s = score; all[i].p = pid;
    }
    qsort(all, (size_t)npids, sizeof(orc_sp_t), cmp_sp_desc);
    int n = (int)(npids < n_keep ? npids : n_keep);
    for (int i = 0; i < n; i++) { out_pids[i] = all[i].p; if (out_scores) out_scores[i] = all[i].s; }
    free(all); free(per);
    return n;
}

/* filter_pids (:126-164): pass 1 with idx keeps ndocs, pass 2 with all centroids keeps ndocs/4. */
ORC_API int orc_filter_pids(const int32_t* pids, int64_t npids, const float* cs, int nq, const int32_t* codes,
                            const int64_t* doclens, const int64_t* offsets, const uint8_t* idx, int ndocs,
                            int32_t* out_pids /* cap ndocs/4 */) {
    int32_t* s1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ndocs > 0 ? ndocs : 1));
    int n1 = orc_filter_pass(pids, npids, cs, nq, codes, doclens, offsets, idx, ndocs, s1, NULL);
    int n2 = orc_filter_pass(s1, n1, cs, nq, codes, doclens, offsets, NULL, ndocs / 4, out_pids, NULL);
    free(s1);
    return n2;
}

/* ------------------------------------------------------------------------------------------
 * S3a  residual decompression   (TPC/search/decompress_residuals.cpp:27-78)
 *   out[t, k*vpb + l] = bucket_weights[lut[rev[byte_k]*vpb + l]] + centroids[code_t, k*vpb + l]
 * (that operand order), tokens packed in pid order.  Returns the number of rows written.
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_decompress_residuals(const int32_t* pids, int npids, const int64_t* doclens,
                                         const int64_t* offsets, const float* bucket_weights,
                                         const uint8_t* reversed_bit_map, const uint8_t* lut,
                                         const uint8_t* residuals, const int32_t* codes, const float* centroids,
                                         int dim, int nbits, float* out) {
    const int vpb = 8 / nbits;
    const int packed_dim = dim / vpb;
    int64_t row = 0;
    for (int i = 0; i < npids; i++) {
        int32_t pid = pids[i];
        int64_t off = offsets[pid];
        for (int64_t j = 0; j < doclens[pid]; j++, row++) {
            int32_t code = codes[off + j];
            const uint8_t* rb = residuals + (size_t)(off + j) * packed_dim;
            const float* cr = centroids + (size_t)code * dim;
            float* o = out + (size_t)row * dim;
            for (int k = 0; k < packed_dim; k++) {
                uint8_t x = reversed_bit_map[rb[k]];
                for (int l = 0; l < vpb; l++) {
                    int d = k * vpb + l;
                    o[d] = bucket_weights[lut[(int)x * vpb + l]] + cr[d];
                }
            }
        }
    }
    return row;
}

/* S3b  F.normalize(p=2, dim=-1, eps=1e-12)  (TPC/search/index_storage.py:173): x / max(||x||, eps) */
ORC_API void orc_normalize_rows(float* D, int64_t n, int dim) {
    for (int64_t r = 0; r < n; r++) {
        float* x = D + (size_t)r * dim;
        float ss = 0.0f;
        for (int k = 0; k < dim; k++) ss += x[k] * x[k];
        float nrm = sqrtf(ss);
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (int k = 0; k < dim; k++) x[k] = x[k] / nrm;
    }
}

/* ------------------------------------------------------------------------------------------
 * S3d  segmented MaxSim   (TPC/modeling/segmented_maxsim.cpp:22-47,49-93)
 *   per doc: column-wise running max over its token rows, INITIAL VALUE 0 (:58-59 torch::zeros),
 *   then sum over query tokens (:92 torch sum; order undefined -> k-ascending here).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_segmented_maxsim(const float* scores, const int64_t* lengths, int ndocs, int nq, float* out) {
    float* mx = (float*)malloc(sizeof(float) * (size_t)nq);
    const float* row = scores;
    for (int i = 0; i < ndocs; i++) {
        for (int k = 0; k < nq; k++) mx[k] = 0.0f;
        for (int64_t j = 0; j < lengths[i]; j++, row += nq)
            for (int k = 0; k < nq; k++) mx[k] = mx[k] > row[k] ? mx[k] : row[k];
        float s = 0.0f;
        for (int k = 0; k < nq; k++) s += mx[k];
        out[i] = s;
    }
    free(mx);
}

/* S3c+S3d  colbert_score_packed (TPC/modeling/colbert.py:289-311): D_packed @ Q.T then segmented maxsim */
ORC_API void orc_maxsim_packed(const float* D, const float* Q, const int64_t* lengths, int ndocs, int nq, int dim,
                               float* out) {
    int64_t n = 0;
    for (int i = 0; i < ndocs; i++) n += lengths[i];
    float* sc = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * (size_t)nq);
    for (int64_t t = 0; t < n; t++)
        for (int j = 0; j < nq; j++) {
            float acc = 0.0f;
            for (int k = 0; k < dim; k++) acc += D[(size_t)t * dim + k] * Q[(size_t)j * dim + k];
            sc[(size_t)t * nq + j] = acc;
        }
    orc_segmented_maxsim(sc, lengths, ndocs, nq, out);
    free(sc);
}

/* ------------------------------------------------------------------------------------------
 * a9  colbert_score (padded) + colbert_score_reduce  (TPC/modeling/colbert.py:235-286)
 *   scores[b,t,j] = D[b,t,:] . Q[qb,j,:]; padded tokens -> -9999; max over t; sum over j.
 *   q_batch is 1 (Q shared) or B (aligned).  No zero clamp here (that is the packed path only).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_colbert_score_padded(const float* Q, int q_batch, int nq, const float* D, const uint8_t* mask,
                                      int B, int Ld, int dim, float* out) {
    for (int b = 0; b < B; b++) {
        const float* Qb = Q + (q_batch == 1 ? 0 : (size_t)b * nq * dim);
        float total = 0.0f;
        for (int j = 0; j < nq; j++) {
            float m = -INFINITY;
            for (int t = 0; t < Ld; t++) {
                float v;
                if (mask[(size_t)b * Ld + t]) {
                    v = 0.0f;
                    for (int k = 0; k < dim; k++) v += D[((size_t)b * Ld + t) * dim + k] * Qb[(size_t)j * dim + k];
                } else {
                    v = -9999.0f;
                }
                m = m > v ? m : v;
            }
            total += m;
        }
        out[b] = total;
    }
}

/* a4  segmented_lookup (TPC/search/segmented_lookup.cpp:46-47): memcpy each [offset, +length) segment */
ORC_API int64_t orc_segmented_lookup(const uint8_t* input, int64_t row_bytes, const int64_t* lengths,
                                     const int64_t* offsets, int nseg, uint8_t* out) {
    int64_t w = 0;
    for (int i = 0; i < nseg; i++) {
        memcpy(out + w * row_bytes, input + offsets[i] * row_bytes, (size_t)(lengths[i] * row_bytes));
        w += lengths[i];
    }
    return w;
}

/* ------------------------------------------------------------------------------------------
 * a2  IndexScorer.rank end to end (TPC/search/index_storage.py:86-98) for ONE query.
 *   Q [nq, dim]; candidate generation uses the first nq_cand rows (index_storage.py:77).
 *   Final order: descending score (torch sort is unstable; ties -> larger pid first here).
 *   Returns the number of ranked docs (<= ndocs/4); out_* capacity ndocs/4.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dim, nbits, K;
    int64_t num_passages;
    const int32_t* codes; const uint8_t* residuals;
    const int64_t* doclens; const int64_t* offsets;
    const int32_t* ivf; const int64_t* ivf_offsets;
    const float* centroids; const float* bucket_weights;
    const uint8_t* reversed_bit_map; const uint8_t* lut;
} orc_index_t;

ORC_API int orc_rank(const orc_index_t* ix, const float* Q, int nq, int nq_cand, int ncells, float thr, int ndocs,
                     int32_t* out_pids, float* out_scores, int64_t* out_ncand) {
    if (nq_cand > nq) nq_cand = nq;
    const int K = ix->K, dim = ix->dim;
    float* cs = (float*)malloc(sizeof(float) * (size_t)K * (size_t)nq_cand);
    orc_centroid_scores(ix->centroids, Q, K, nq_cand, dim, cs);
    int32_t* cells = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq_cand * (size_t)ncells);
    int ncell = orc_select_cells(cs, K, nq_cand, ncells, cells);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ix->num_passages > 0 ? ix->num_passages : 1));
    int64_t P = orc_candidates(cells, ncell, ix->ivf, ix->ivf_offsets, ix->num_passages, cand);
    if (out_ncand) *out_ncand = P;
    uint8_t* idx = (uint8_t*)malloc((size_t)K);
    orc_idx_mask(cs, K, nq_cand, thr, idx);
    int nfin_cap = ndocs / 4 > 0 ? ndocs / 4 : 1;
    int32_t* fin = (int32_t*)malloc(sizeof(int32_t) * (size_t)nfin_cap);
    int nfin = orc_filter_pids(cand, P, cs, nq_cand, ix->codes, ix->doclens, ix->offsets, idx, ndocs, fin);
    int64_t ntok = 0;
    int64_t* lens = (int64_t*)malloc(sizeof(int64_t) * (size_t)nfin_cap);
    for (int i = 0; i < nfin; i++) { lens[i] = ix->doclens[fin[i]]; ntok += lens[i]; }
    float* D = (float*)malloc(sizeof(float) * (size_t)(ntok > 0 ? ntok : 1) * (size_t)dim);
    orc_decompress_residuals(fin, nfin, ix->doclens, ix->offsets, ix->bucket_weights, ix->reversed_bit_map, ix->lut,
                             ix->residuals, ix->codes, ix->centroids, dim, ix->nbits, D);
    orc_normalize_rows(D, ntok, dim);
    float* sc = (float*)malloc(sizeof(float) * (size_t)nfin_cap);
    orc_maxsim_packed(D, Q, lens, nfin, nq, dim, sc);
    orc_sp_t* sp = (orc_sp_t*)malloc(sizeof(orc_sp_t) * (size_t)nfin_cap);
    for (int i = 0; i < nfin; i++) { sp[i].s = sc[i]; sp[i].p = fin[i]; }
    qsort(sp, (size_t)nfin, sizeof(orc_sp_t), cmp_sp_desc);
    for (int i = 0; i < nfin; i++) { out_pids[i] = sp[i].p; out_scores[i] = sp[i].s; }
    free(cs); free(cells); free(cand); free(idx); free(fin); free(lens); free(D); free(sc); free(sp);
    return nfin;
}

/* Batched driver used by bench.py's cpu_baseline ("port") leg: one query per call semantics, queries
 * distributed over OpenMP threads.  out_* are [nqueries, k] (short lists padded with pid -1 / score 0). */
static int orc_threads = 0;   /* 0: the OpenMP default; the caller sets the CPUs it may really use (cgroup quota, not nproc) */
ORC_API void orc_set_threads(int n) { orc_threads = n > 0 ? n : 0; }

ORC_API void orc_search_batch(const orc_index_t* ix, const float* Q, int nqueries, int nq, int nq_cand, int ncells,
                              float thr, int ndocs, int k, int32_t* out_pids, float* out_scores, int32_t* out_counts) {
    const int nthreads = orc_threads > 0 ? orc_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int b = 0; b < nqueries; b++) {
        int cap = ndocs / 4 > 0 ? ndocs / 4 : 1;
        int32_t* p = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
        float* s = (float*)malloc(sizeof(float) * (size_t)cap);
        int n = orc_rank(ix, Q + (size_t)b * nq * ix->dim, nq, nq_cand, ncells, thr, ndocs, p, s, NULL);
        int m = n < k ? n : k;
        for (int i = 0; i < k; i++) {
            out_pids[(size_t)b * k + i] = i < m ? p[i] : -1;
            out_scores[(size_t)b * k + i] = i < m ? s[i] : 0.0f;
        }
        out_counts[b] = m;
        free(p); free(s);
    }
}
