// Index residency: host arrays in the reference's CPU layout -> HBM (flmr_index_open / flmr_index_close).
// Replaces IndexLoader + ResidualCodec.load + ResidualEmbeddings.load_chunks (TPC/search/index_loader.py:14-86,
// TPC/indexing/codecs/residual.py:134-150, TPC/indexing/codecs/residual_embeddings.py:27-52).
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "flmr_common.h"

thread_local char flmr_err_buf[512] = {0};

static const char* kOptNames[FLMR_OPT_COUNT] = {"FLMR_S0_IMPL", "FLMR_FULL_TABLE", "FLMR_CAND_IMPL", "FLMR_S1_NO_HITMAP",
                                                 "FLMR_S1_IMPL", "FLMR_S2_IMPL", "FLMR_S0_STAGED", "FLMR_S3_NO_MULTIQ",
                                                 "FLMR_S3_IMPL", "FLMR_SCORE_IMPL", "FLMR_POISON", "FLMR_ROW_CAP"};
static flmr_options g_opts;
static std::once_flag g_opts_once;
static thread_local const flmr_options* t_active_opts = nullptr;

static void opts_from_env() {
    memset(&g_opts, 0, sizeof(g_opts));
    for (int i = 0; i < FLMR_OPT_COUNT; i++) {
        const char* e = getenv(kOptNames[i]);
        if (e) snprintf(g_opts.v[i], sizeof(g_opts.v[i]), "%s", e[0] ? e : "1");
    }
}
const flmr_options& flmr_process_options() {
    std::call_once(g_opts_once, opts_from_env);
    return g_opts;
}
const flmr_options& flmr_opts() { return t_active_opts ? *t_active_opts : flmr_process_options(); }
flmr_opt_scope::flmr_opt_scope(const flmr_options* o) : prev(t_active_opts) { t_active_opts = o; }
flmr_opt_scope::~flmr_opt_scope() { t_active_opts = prev; }

// name = one of the FLMR_* switch names; value NULL or "" clears it.  Affects op-level calls at once and searchers
// created afterwards (existing searchers keep the snapshot taken at flmr_searcher_create).
extern "C" int flmr_set_option(const char* name, const char* value) {
    if (!name) FLMR_FAIL(FLMR_ERR_INVALID, "name is NULL");
    (void)flmr_process_options();
    for (int i = 0; i < FLMR_OPT_COUNT; i++)
        if (strcmp(name, kOptNames[i]) == 0) {
            if (value && strlen(value) >= sizeof(g_opts.v[i])) FLMR_FAIL(FLMR_ERR_INVALID, "value too long for %s", name);
            snprintf(g_opts.v[i], sizeof(g_opts.v[i]), "%s", value ? value : "");
            return FLMR_OK;
        }
    FLMR_FAIL(FLMR_ERR_INVALID, "unknown option %s", name);
}

extern "C" int flmr_abi_version(void) { return FLMR_ABI_VERSION; }
extern "C" const char* flmr_last_error(void) { return flmr_err_buf; }

extern "C" int flmr_device_count(int* count) {
    if (!count) FLMR_FAIL(FLMR_ERR_INVALID, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        FLMR_FAIL(FLMR_ERR_HIP, "no HIP device visible (%s)", hipGetErrorString(e));
    }
    *count = n;
    return FLMR_OK;
}

// reversed_bit_map / decompression_lookup_table as ResidualCodec.__init__ builds them
// (TPC/indexing/codecs/residual.py:51-95): bit-reverse inside each nbits group; base-2^nbits digits MSB first.
void flmr_default_codec_tables(int nbits, uint8_t* rev, uint8_t* combos) {
    const int vpb = 8 / nbits, mask = (1 << nbits) - 1;
    for (int i = 0; i < 256; i++) {
        int z = 0;
        for (int g = 0; g < vpb; g++) {
            const int sh = 8 - nbits * (g + 1);
            const int x = (i >> sh) & mask;
            int y = 0;
            for (int b = 0; b < nbits; b++) y |= ((x >> b) & 1) << (nbits - 1 - b);
            z |= y << sh;
            combos[i * vpb + g] = (uint8_t)x;
        }
        rev[i] = (uint8_t)z;
    }
}

void flmr_build_wlut(int nbits, const float* bucket_weights, const uint8_t* rev, const uint8_t* combos, float* wlut) {
    const int vpb = 8 / nbits;
    for (int byte = 0; byte < 256; byte++)
        for (int l = 0; l < vpb; l++) wlut[byte * vpb + l] = bucket_weights[combos[(int)rev[byte] * vpb + l]];
}

template <typename T>
static int to_device(const T* src, size_t count, int memory, T** dst) {
    if (memory == FLMR_MEM_DEVICE) {
        *dst = const_cast<T*>(src);
        return FLMR_OK;
    }
    const size_t bytes = (count ? count : 1) * sizeof(T);
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (count) FLMR_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return FLMR_OK;
}

extern "C" int flmr_index_open(const flmr_index_desc_t* d, flmr_index_t** out) {
    if (!d || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (d->dim != FLMR_DIM) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dim=%d (this build is specialised for dim=128)", d->dim);
    if (d->nbits != 1 && d->nbits != 2 && d->nbits != 4 && d->nbits != 8)
        FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d", d->nbits);
    if (d->num_centroids <= 0 || d->num_embeddings < 0 || d->num_passages < 0) FLMR_FAIL(FLMR_ERR_INVALID, "negative sizes");
    if (d->num_passages + d->pid_base > 0x7fffffffLL) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "pids do not fit int32");
    if (!d->codes || !d->residuals || !d->doc_offsets || !d->ivf_pids || !d->ivf_offsets || !d->centroids ||
        !d->bucket_weights)
        FLMR_FAIL(FLMR_ERR_INVALID, "NULL array in descriptor");
    int ndev = 0;
    int rc = flmr_device_count(&ndev);
    if (rc) return rc;

    flmr_index* ix = new (std::nothrow) flmr_index();
    if (!ix) FLMR_FAIL(FLMR_ERR_NOMEM, "host allocation failed");
    memset(ix, 0, sizeof(*ix));
    ix->dim = d->dim; ix->nbits = d->nbits; ix->K = d->num_centroids;
    ix->N = d->num_embeddings; ix->num_passages = d->num_passages; ix->pid_base = d->pid_base;
    ix->packed_dim = d->dim * d->nbits / 8;
    ix->owns = (d->memory == FLMR_MEM_HOST);
    if (hipGetDevice(&ix->device) != hipSuccess) {
        delete ix;
        FLMR_FAIL(FLMR_ERR_HIP, "hipGetDevice failed");
    }
    const int K = ix->K;

    // host-side metadata: IVF offsets (for the candidate-capacity bound) and the longest document
    std::vector<int64_t> ivf_off((size_t)K + 1), doc_off((size_t)ix->num_passages + 1);
    if (d->memory == FLMR_MEM_HOST) {
        memcpy(ivf_off.data(), d->ivf_offsets, sizeof(int64_t) * ((size_t)K + 1));
        memcpy(doc_off.data(), d->doc_offsets, sizeof(int64_t) * ((size_t)ix->num_passages + 1));
    } else {
        FLMR_HIP(hipMemcpy(ivf_off.data(), d->ivf_offsets, sizeof(int64_t) * ((size_t)K + 1), hipMemcpyDeviceToHost));
        FLMR_HIP(hipMemcpy(doc_off.data(), d->doc_offsets, sizeof(int64_t) * ((size_t)ix->num_passages + 1),
                           hipMemcpyDeviceToHost));
    }
    if (doc_off[ix->num_passages] != ix->N) {
        delete ix;
        FLMR_FAIL(FLMR_ERR_INVALID, "doc_offsets[-1]=%lld != num_embeddings=%lld", (long long)doc_off[ix->num_passages],
                  (long long)ix->N);
    }
    ix->max_doclen = 0;
    for (int64_t p = 0; p < ix->num_passages; p++) ix->max_doclen = std::max(ix->max_doclen, doc_off[p + 1] - doc_off[p]);
    std::vector<int64_t> lens((size_t)K);
    for (int c = 0; c < K; c++) lens[c] = ivf_off[c + 1] - ivf_off[c];
    std::sort(lens.begin(), lens.end(), [](int64_t x, int64_t y) { return x > y; });
    ix->ivf_len_prefix = new int64_t[(size_t)K + 1];
    ix->ivf_len_prefix[0] = 0;
    for (int c = 0; c < K; c++) ix->ivf_len_prefix[c + 1] = ix->ivf_len_prefix[c] + lens[c];
    const int64_t ivf_total = ivf_off[K];

#define FLMR_TRY(x)                 \
    do {                            \
        rc = (x);                   \
        if (rc) { flmr_index_close(ix); return rc; } \
    } while (0)
    FLMR_TRY(to_device(d->codes, (size_t)ix->N, d->memory, &ix->codes));
    FLMR_TRY(to_device(d->residuals, (size_t)ix->N * ix->packed_dim, d->memory, &ix->residuals));
    FLMR_TRY(to_device(d->doc_offsets, (size_t)ix->num_passages + 1, d->memory, &ix->doc_offsets));
    FLMR_TRY(to_device(d->ivf_pids, (size_t)ivf_total, d->memory, &ix->ivf_pids));
    FLMR_TRY(to_device(d->ivf_offsets, (size_t)K + 1, d->memory, &ix->ivf_offsets));
    FLMR_TRY(to_device(d->centroids, (size_t)K * FLMR_DIM, d->memory, &ix->centroids));
    FLMR_TRY(flmr_check_f16_exact(ix->centroids, (size_t)K * FLMR_DIM, &ix->centroids_f16_exact));
    FLMR_TRY(flmr_max_row_norm(ix->centroids, K, &ix->cen_norm_max));
    if (ix->centroids_f16_exact) {
        rc = hipMalloc(reinterpret_cast<void**>(&ix->centroids_f16), (size_t)K * FLMR_DIM * sizeof(_Float16)) == hipSuccess ? FLMR_OK : FLMR_ERR_NOMEM;
        if (rc) { snprintf(flmr_err_buf, sizeof(flmr_err_buf), "hipMalloc centroids_f16"); flmr_index_close(ix); return rc; }
        FLMR_TRY(flmr_convert_f16(ix->centroids, (size_t)K * FLMR_DIM, ix->centroids_f16));
    }
    FLMR_TRY(flmr_build_sorted_codes(ix));
    FLMR_TRY(flmr_build_tiled_centroids(ix));
    FLMR_TRY(flmr_build_doc_splits(ix));
    FLMR_TRY(flmr_build_chunk_table(ix->ivf_pids, ix->ivf_offsets, K, ix->num_passages, &ix->ivf_chunk_tab, &ix->nchunks));
    // fused decode table (always built on the host from the host bucket_weights)
    {
        const int vpb = 8 / ix->nbits;
        uint8_t rev[256], combos[256 * 8];
        float wl[256 * 8];
        for (int i = 0; i < (1 << ix->nbits); i++) ix->bucket_weights[i] = d->bucket_weights[i];
        flmr_default_codec_tables(ix->nbits, rev, combos);
        flmr_build_wlut(ix->nbits, ix->bucket_weights, rev, combos, wl);
        float* dw = nullptr;
        rc = to_device(wl, (size_t)256 * vpb, FLMR_MEM_HOST, &dw);
        if (rc) { flmr_index_close(ix); return rc; }
        ix->wlut = dw;
    }
    FLMR_TRY(flmr_build_s3_tables(ix));
#undef FLMR_TRY
    *out = ix;
    return FLMR_OK;
}

extern "C" int flmr_index_info(const flmr_index_t* ix, flmr_index_info_t* out) {
    if (!ix || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    const size_t K = (size_t)ix->K;
    size_t b = 256 * (8 / ix->nbits) * sizeof(float);
    if (ix->centroids_f16) b += K * FLMR_DIM * sizeof(_Float16);
    if (ix->centroids_f16_tiled) b += K * FLMR_DIM * sizeof(_Float16);
    if (ix->codes_sorted) b += ((size_t)ix->N + FLMR_CODE_PAD) * sizeof(int32_t);
    if (ix->doc_ulen) b += (size_t)ix->num_passages * sizeof(uint16_t);
    if (ix->doc_splits) b += (size_t)ix->num_passages * ix->nslices * sizeof(uint16_t);
    if (ix->ivf_chunk_tab) b += K * ((size_t)ix->nchunks + 1) * sizeof(uint32_t);
    if (ix->inv_norm) b += ((size_t)ix->N + 64) * sizeof(float);
    out->derived_bytes = (int64_t)b;
    out->max_doclen = ix->max_doclen;
    out->centroids_f16_exact = ix->centroids_f16_exact;
    out->stage2_slices = ix->nslices;
    out->xcd_round_robin = ix->xcd_round_robin;
    out->stage2_sliced = flmr_stage2_xcd_pays(ix) ? 1 : 0;
    out->passage_chunks = ix->nchunks;
    out->duplicate_permille = (int32_t)(ix->dup_share * 1000.0 + 0.5);
    return FLMR_OK;
}

extern "C" int flmr_index_close(flmr_index_t* ix) {
    if (!ix) return FLMR_OK;
    if (ix->owns) {
        (void)hipFree(ix->codes); (void)hipFree(ix->residuals); (void)hipFree(ix->doc_offsets);
        (void)hipFree(ix->ivf_pids); (void)hipFree(ix->ivf_offsets); (void)hipFree(ix->centroids);
    }
    (void)hipFree(ix->wlut);
    (void)hipFree(ix->wtab16);
    (void)hipFree(ix->inv_norm);
    (void)hipFree(ix->centroids_f16);
    (void)hipFree(ix->ivf_chunk_tab);
    (void)hipFree(ix->codes_sorted);
    (void)hipFree(ix->doc_ulen);
    (void)hipFree(ix->centroids_f16_tiled);
    (void)hipFree(ix->doc_splits);
    delete[] ix->ivf_len_prefix;
    delete ix;
    return FLMR_OK;
}
