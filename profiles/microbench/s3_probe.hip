// Stand-alone timing harness for the fused decompress + MaxSim kernels (csrc/flmr_maxsim.hip) on synthetic inputs of BASELINE's
// shape: 1024 queries x 256 finalists x 128 tokens, K = 131072, nbits = 2.  -DS3P_DMA=1 forces the LDS-DMA form; the
// -DS3P_NO_* switches remove parts of that kernel (results become garbage, timings do not): what does a step consist of?
// Build (repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iretrieval-augmented-visual-question-answering_amd/csrc \
//                          [-DS3P_NO_MFMA -DS3P_NO_DECODE ...] -o profiles/microbench/s3_probe profiles/microbench/s3_probe.hip
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_maxsim.hip"

#include <random>
#include <vector>

thread_local char flmr_err_buf[512] = {0};
static flmr_options g_opts;
const flmr_options& flmr_opts() { return g_opts; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int K = getenv("S3P_K") ? atoi(getenv("S3P_K")) : 131072, P = 200000, L = 128, ND = 256, NBITS = 2;
    const int NQR = getenv("S3P_NQ") ? atoi(getenv("S3P_NQ")) : 32;          // S3P_NQ: query rows (> 32: the long-query kernels)
    const int NQ = getenv("S3P_B") ? atoi(getenv("S3P_B")) : (NQR > 32 ? 256 : 1024);   // queries per launch
    const int NQP = (NQR + 31) / 32 * 32;   // S3P_K: a smaller (cache-resident) centroid table
    memset(&g_opts, 0, sizeof g_opts);
    std::mt19937 rng(1);
    std::vector<int32_t> codes((size_t)P * L);
    for (auto& c : codes) c = (int32_t)(rng() % K);
    std::vector<uint8_t> res((size_t)P * L * 32);
    for (auto& x : res) x = (uint8_t)rng();
    std::vector<int64_t> off(P + 1);
    const bool ragged = getenv("S3P_RAGGED") != nullptr;   // S3P_RAGGED: passage lengths ~ U{32..224} (mean 128) inside the same token array
    int maxlen = L;
    off[0] = 0;
    for (int p = 0; p < P; p++) {
        int len = ragged ? 32 + (int)(rng() % 193) : L;
        if (off[p] + len > (int64_t)P * L) len = (int)((int64_t)P * L - off[p]);
        off[p + 1] = off[p] + len;
        if (len > maxlen) maxlen = len;
    }
    std::vector<int32_t> pids((size_t)NQ * ND), counts(NQ, ND);
    for (auto& x : pids) x = (int32_t)(rng() % P);
    std::vector<_Float16> cen((size_t)K * 128);
    for (auto& x : cen) x = (_Float16)((float)((int)(rng() % 2001) - 1000) / 8000.0f);
    std::vector<float> Q((size_t)NQ * NQR * 128), wl(256 * 4);
    for (auto& x : Q) x = (float)((int)(rng() % 2001) - 1000) / 8000.0f;
    for (auto& x : wl) x = (float)((int)(rng() % 2001) - 1000) / 80000.0f;
    flmr_index ix{};
    ix.K = K; ix.nbits = NBITS; ix.N = (int64_t)P * L; ix.num_passages = P; ix.centroids_f16_exact = 1; ix.packed_dim = 32;
    float* dQ; int32_t *d_pids, *d_counts; uint64_t* d_keys; _Float16 *qh, *ql;
    CK(hipMalloc(&ix.codes, codes.size() * 4)); CK(hipMemcpy(ix.codes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.residuals, res.size())); CK(hipMemcpy(ix.residuals, res.data(), res.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.doc_offsets, off.size() * 8)); CK(hipMemcpy(ix.doc_offsets, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.centroids_f16, cen.size() * 2)); CK(hipMemcpy(ix.centroids_f16, cen.data(), cen.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.wlut, wl.size() * 4)); CK(hipMemcpy(ix.wlut, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pids, pids.size() * 4)); CK(hipMemcpy(d_pids, pids.data(), pids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counts, counts.size() * 4)); CK(hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_keys, (size_t)NQ * ND * 8)); CK(hipMalloc(&qh, (size_t)NQ * NQP * 128 * 2)); CK(hipMalloc(&ql, (size_t)NQ * NQP * 128 * 2));
    flmr_maxsim_args a{};
    a.ix = &ix; a.Q = dQ; a.q_lens = nullptr; a.nqueries = NQ; a.nq = NQR; a.pids = d_pids; a.pid_stride = ND; a.counts = d_counts;
    a.max_count = ND; a.keys = d_keys; a.key_stride = ND; a.scores = nullptr; a.q_hi = qh; a.q_lo = ql;
    // "cw" / "lean": the index-side tables of the centroid + weight form, and the planned-tile kernel's workspace
    {
        std::vector<float> cen32(cen.size());
        for (size_t t = 0; t < cen.size(); t++) cen32[t] = (float)cen[t];
        CK(hipMalloc(&ix.centroids, cen32.size() * 4)); CK(hipMemcpy(ix.centroids, cen32.data(), cen32.size() * 4, hipMemcpyHostToDevice));
        ix.max_doclen = maxlen;
        if (flmr_build_s3_tables(&ix) != 0 || !ix.inv_norm) { printf("no S3 tables\n"); return 1; }
        a.plan_stride = (int64_t)ND * ((maxlen + 31) / 32); a.plan_wcap = ND + 8;
        CK(hipMalloc(&a.plan_desc, (size_t)NQ * a.plan_stride * sizeof(uint2)));
        CK(hipMalloc(&a.plan_wbeg, (size_t)NQ * a.plan_wcap * 4));
        if (NQR > 32) { a.colmax_cap = (int64_t)NQ * ND * NQP + NQ; CK(hipMalloc(&a.colmax_ws, (size_t)a.colmax_cap * 4)); }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint64_t> ref;
    const char* impls[8]; int nimpl = 0;
    for (int k = 1; k < argc && nimpl < 8; k++) impls[nimpl++] = argv[k];
    if (!nimpl) impls[nimpl++] = "dma";
    for (int v = 0; v < nimpl; v++) {
        snprintf(g_opts.v[FLMR_OPT_S3_IMPL], sizeof g_opts.v[0], "%s", impls[v]);
        CK(hipMemset(d_keys, 0xAB, (size_t)NQ * ND * 8));
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, 0));
            if (flmr_launch_maxsim(a, 0) != 0) { printf("launch failed: %s\n", flmr_err_buf); return 1; }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("S3 (%s): %.3f ms per %d queries (%d rows) x %d finalists\n", impls[v], ms, NQ, NQR, ND);
        }
#ifdef S3Q_PROFILE
        if (NQR > 32 && !strcmp(impls[v], "qs")) {
            unsigned long long pr[64];
            CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(s3q_prof), sizeof pr));
            printf("   s_memtime ticks per round and wave (all launches): top-of-round wait | G + C issue | decode | consume | barrier\n");
            for (int w = 0; w < 8; w++) {
                const double n = (double)pr[w * 8 + 7];
                printf("   wave %d: %8.1f %8.1f %8.1f %8.1f %8.1f   (%.0f rounds)\n", w, pr[w * 8] / n, pr[w * 8 + 1] / n, pr[w * 8 + 2] / n,
                       pr[w * 8 + 3] / n, pr[w * 8 + 4] / n, n);
            }
        }
#endif
        std::vector<uint64_t> got((size_t)NQ * ND);
        CK(hipMemcpy(got.data(), d_keys, got.size() * 8, hipMemcpyDeviceToHost));
        printf("   scores[0..3] %.7g %.7g %.7g %.7g\n", flmr_key_score(got[0]), flmr_key_score(got[1]), flmr_key_score(got[2]), flmr_key_score(got[3]));
        if (v == 0) ref = got;
        else {
            size_t bad = 0; double maxd = 0;
            for (size_t t = 0; t < got.size(); t++) {
                if (got[t] != ref[t]) bad++;
                const double dd = fabs((double)flmr_key_score(got[t]) - (double)flmr_key_score(ref[t]));
                if (dd > maxd) maxd = dd;
            }
            printf("   vs %s: %zu of %zu keys differ, max |dscore| %.3g\n", impls[0], bad, got.size(), maxd);
        }
    }
    return 0;
}
