// Drives the C ABI of libflmr_hip.so from plain C++ (no python, no torch): the program the AddressSanitizer / UBSan build of
// the library's HOST side runs under (profiles/sanitize_host.sh) -- torch's HIP start-up does not survive a preloaded
// sanitizer runtime, a plain HIP program does.  It builds a small synthetic index in host memory (the layout of
// include/flmr_hip.h: codes, packed residuals, passage offsets, IVF, centroids, bucket weights), opens it with
// FLMR_MEM_HOST, and walks the entry points a caller uses: info, searcher create, batched search at two k-policies with
// ragged q_lens, taps, deferred check, the op-level calls, the argument errors, destroy / close -- twice, for nbits 2 and 8.
// Results are sanity-checked (counts, pid range, descending scores); parity is the job of tests/test_hip_parity.py.
// Build (GPU box or here): hipcc --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
//        -Iinclude tests/native/abi_harness.cpp -o tests/native/abi_harness -L<lib dir> -lflmr_hip_asan -Wl,-rpath,<lib dir>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <cstring>
#include <vector>
#include "flmr_hip.h"

#define REQUIRE(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s   [%s]\n", __LINE__, #cond, flmr_last_error()); exit(1); } } while (0)
#define HIPOK(x) REQUIRE((x) == hipSuccess)

struct HostIndex {
    int K, nbits, dim = 128;
    int64_t npass, ntok;
    std::vector<int32_t> codes, ivf_pids;
    std::vector<uint8_t> residuals;
    std::vector<int64_t> doc_offsets, ivf_offsets;
    std::vector<float> centroids, buckets;
};

static float f16_round(float x) { return (float)(_Float16)x; }

static HostIndex make_index(int K, int nbits, int64_t npass, unsigned seed) {
    std::mt19937 g(seed);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    HostIndex ix;
    ix.K = K; ix.nbits = nbits; ix.npass = npass;
    ix.centroids.resize((size_t)K * 128);
    for (int c = 0; c < K; c++) {
        float n2 = 0.0f;
        for (int d = 0; d < 128; d++) { const float v = nd(g); ix.centroids[(size_t)c * 128 + d] = v; n2 += v * v; }
        const float inv = 1.0f / std::sqrt(n2);
        for (int d = 0; d < 128; d++) ix.centroids[(size_t)c * 128 + d] = f16_round(ix.centroids[(size_t)c * 128 + d] * inv);
    }
    ix.doc_offsets.assign(1, 0);
    for (int64_t p = 0; p < npass; p++) {
        const int len = (p % 97 == 0) ? 0 : 4 + (int)(g() % 60);   // some empty passages
        ix.doc_offsets.push_back(ix.doc_offsets.back() + len);
    }
    ix.ntok = ix.doc_offsets.back();
    ix.codes.resize((size_t)ix.ntok);
    for (auto& c : ix.codes) c = (int32_t)(g() % K);
    ix.residuals.resize((size_t)ix.ntok * (128 * nbits / 8));
    for (auto& r : ix.residuals) r = (uint8_t)(g() & 0xff);
    ix.buckets.resize((size_t)1 << nbits);
    for (size_t b = 0; b < ix.buckets.size(); b++) ix.buckets[b] = -0.08f + 0.16f * (float)b / (float)(ix.buckets.size() - 1);
    // IVF: sorted unique passages per centroid (indexing/utils.py:8-53)
    std::vector<std::vector<int32_t>> lists((size_t)K);
    for (int64_t p = 0; p < npass; p++)
        for (int64_t t = ix.doc_offsets[p]; t < ix.doc_offsets[p + 1]; t++) {
            auto& l = lists[(size_t)ix.codes[(size_t)t]];
            if (l.empty() || l.back() != (int32_t)p) l.push_back((int32_t)p);
        }
    ix.ivf_offsets.assign(1, 0);
    for (auto& l : lists) { ix.ivf_pids.insert(ix.ivf_pids.end(), l.begin(), l.end()); ix.ivf_offsets.push_back((int64_t)ix.ivf_pids.size()); }
    return ix;
}

template <typename T>
static T* to_device(const std::vector<T>& v) {
    T* d = nullptr;
    HIPOK(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) HIPOK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static void run_case(int K, int nbits, int64_t npass, unsigned seed) {
    HostIndex h = make_index(K, nbits, npass, seed);
    flmr_index_desc_t desc = {};
    desc.dim = 128; desc.nbits = nbits; desc.num_centroids = K; desc.memory = FLMR_MEM_HOST;
    desc.num_embeddings = h.ntok; desc.num_passages = h.npass; desc.pid_base = 0;
    desc.codes = h.codes.data(); desc.residuals = h.residuals.data(); desc.doc_offsets = h.doc_offsets.data();
    desc.ivf_pids = h.ivf_pids.data(); desc.ivf_offsets = h.ivf_offsets.data(); desc.centroids = h.centroids.data();
    desc.bucket_weights = h.buckets.data();
    flmr_index_t* ix = nullptr;
    REQUIRE(flmr_index_open(&desc, &ix) == FLMR_OK && ix);
    flmr_index_info_t info = {};
    REQUIRE(flmr_index_info(ix, &info) == FLMR_OK);
    REQUIRE(info.centroids_f16_exact == 1 && info.max_doclen <= 64 && info.derived_bytes > 0);
    // argument errors come back as codes, never as crashes
    REQUIRE(flmr_index_open(nullptr, &ix) != FLMR_OK);
    REQUIRE(flmr_index_info(nullptr, &info) != FLMR_OK);
    { flmr_index_desc_t bad = desc; bad.nbits = 3; flmr_index_t* t = nullptr; REQUIRE(flmr_index_open(&bad, &t) != FLMR_OK && !t); }
    { flmr_index_desc_t bad = desc; bad.dim = 64; flmr_index_t* t = nullptr; REQUIRE(flmr_index_open(&bad, &t) != FLMR_OK && !t); }

    const int nqueries = 19, nq = 32;
    flmr_search_params_t maxp = {64, 4, 0.3f, 1024, 32};
    flmr_searcher_t* s = nullptr;
    REQUIRE(flmr_searcher_create(ix, 32, nq, &maxp, &s) == FLMR_OK && s);
    int64_t ws = 0;
    REQUIRE(flmr_searcher_workspace_bytes(s, &ws) == FLMR_OK && ws > 0);
    std::mt19937 g(seed + 1);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> Q((size_t)nqueries * nq * 128);
    for (int q = 0; q < nqueries * nq; q++) {   // a centroid plus noise, normalised: scores above the threshold exist
        const int c = (int)(g() % K);
        float n2 = 0.0f;
        for (int d = 0; d < 128; d++) { const float v = h.centroids[(size_t)c * 128 + d] + 0.05f * nd(g); Q[(size_t)q * 128 + d] = v; n2 += v * v; }
        for (int d = 0; d < 128; d++) Q[(size_t)q * 128 + d] /= std::sqrt(n2);
    }
    std::vector<int32_t> q_lens(nqueries, nq);
    q_lens[3] = 7; q_lens[5] = 1; q_lens[11] = 20;
    float* dQ = to_device(Q);
    int32_t* dL = to_device(q_lens);
    for (const flmr_search_params_t p : {flmr_search_params_t{10, 2, 0.45f, 256, 32}, flmr_search_params_t{64, 4, 0.3f, 1024, 32}}) {
        int32_t *dP = nullptr, *dC = nullptr; float* dS = nullptr;
        HIPOK(hipMalloc(reinterpret_cast<void**>(&dP), (size_t)nqueries * p.k * 4));
        HIPOK(hipMalloc(reinterpret_cast<void**>(&dS), (size_t)nqueries * p.k * 4));
        HIPOK(hipMalloc(reinterpret_cast<void**>(&dC), (size_t)nqueries * 4));
        for (int rep = 0; rep < 2; rep++) {
            REQUIRE(flmr_search_batch(s, dQ, rep ? dL : nullptr, nqueries, nq, &p, dP, dS, dC, nullptr) == FLMR_OK);
            REQUIRE(flmr_searcher_check(s) == FLMR_OK);
            std::vector<int32_t> P((size_t)nqueries * p.k), C(nqueries);
            std::vector<float> S((size_t)nqueries * p.k);
            HIPOK(hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            for (int q = 0; q < nqueries; q++) {
                REQUIRE(C[q] >= 0 && C[q] <= p.k);
                for (int r = 0; r < p.k; r++) {
                    const int32_t pid = P[(size_t)q * p.k + r];
                    if (r < C[q]) { REQUIRE(pid >= 0 && pid < h.npass); if (r) REQUIRE(S[(size_t)q * p.k + r - 1] >= S[(size_t)q * p.k + r]); }
                    else REQUIRE(pid == -1);
                }
            }
            // taps of a query (host buffers with exact and with short capacity)
            std::vector<int32_t> cells(4 * 32);   // capacity is in ELEMENTS
            int64_t cnt = 0;
            REQUIRE(flmr_searcher_tap(s, FLMR_TAP_CELLS, 2, cells.data(), (int64_t)cells.size(), &cnt) == FLMR_OK && cnt > 0 && cnt <= 4 * 32);
            REQUIRE(flmr_searcher_tap(s, FLMR_TAP_CELLS, 2, cells.data(), 1, &cnt) == FLMR_ERR_CAPACITY && cnt > 1);   // short buffer: the needed size, nothing written
            { int64_t need = 0; REQUIRE(flmr_searcher_tap(s, FLMR_TAP_CANDIDATES, 2, cells.data(), 0, &need) != FLMR_OK || need == 0);
              std::vector<int32_t> cand((size_t)std::max<int64_t>(need, 1));
              REQUIRE(flmr_searcher_tap(s, FLMR_TAP_CANDIDATES, 2, cand.data(), (int64_t)cand.size(), &cnt) == FLMR_OK && cnt == need);
              for (int64_t e = 1; e < cnt; e++) REQUIRE(cand[(size_t)e - 1] < cand[(size_t)e]); }
            REQUIRE(flmr_searcher_tap(s, 99, 2, cells.data(), 16, &cnt) != FLMR_OK);
            REQUIRE(flmr_searcher_tap(s, FLMR_TAP_CELLS, nqueries + 5, cells.data(), 16, &cnt) != FLMR_OK);
        }
        // limits: too many queries, too long queries, ndocs / ncells beyond the workspace (k is only the width of the caller's
        // output rows: not a workspace bound)
        REQUIRE(flmr_search_batch(s, dQ, nullptr, 33, nq, &p, dP, dS, dC, nullptr) != FLMR_OK);
        REQUIRE(flmr_search_batch(s, dQ, nullptr, nqueries, nq + 1, &p, dP, dS, dC, nullptr) != FLMR_OK);
        { flmr_search_params_t big = p; big.ndocs = 4096; REQUIRE(flmr_search_batch(s, dQ, nullptr, nqueries, nq, &big, dP, dS, dC, nullptr) == FLMR_ERR_CAPACITY); }
        { flmr_search_params_t big = p; big.ncells = 8; REQUIRE(flmr_search_batch(s, dQ, nullptr, nqueries, nq, &big, dP, dS, dC, nullptr) == FLMR_ERR_CAPACITY); }
        { flmr_search_params_t bad = p; bad.k = 0; REQUIRE(flmr_search_batch(s, dQ, nullptr, nqueries, nq, &bad, dP, dS, dC, nullptr) != FLMR_OK); }
        REQUIRE(flmr_search_batch(s, nullptr, nullptr, nqueries, nq, &p, dP, dS, dC, nullptr) != FLMR_OK);
        // q_lens outside [0, nq]: clamped on the device, reported by the deferred check
        { std::vector<int32_t> bad(nqueries, nq); bad[4] = nq + 9; int32_t* dB = to_device(bad);
          REQUIRE(flmr_search_batch(s, dQ, dB, nqueries, nq, &p, dP, dS, dC, nullptr) == FLMR_OK);
          REQUIRE(flmr_searcher_check(s) != FLMR_OK);
          REQUIRE(flmr_searcher_check(s) == FLMR_OK);   // (the flag is cleared by the report)
          HIPOK(hipFree(dB)); }
        // exact score of a few passages (fused decompress + MaxSim op), same index
        { std::vector<int32_t> pids = {0, 1, 2, (int32_t)h.npass - 1}; int32_t* dp = to_device(pids); float* out = nullptr;
          HIPOK(hipMalloc(reinterpret_cast<void**>(&out), 16));
          REQUIRE(flmr_score_pids(ix, dQ, nq, dp, 4, out, nullptr) == FLMR_OK);
          HIPOK(hipDeviceSynchronize());
          REQUIRE(flmr_score_pids(ix, dQ, nq, nullptr, 4, out, nullptr) != FLMR_OK);
          HIPOK(hipFree(dp)); HIPOK(hipFree(out)); }
        HIPOK(hipFree(dP)); HIPOK(hipFree(dS)); HIPOK(hipFree(dC));
    }
    // both numerics modes and the profiling switch on the same searcher
    REQUIRE(flmr_searcher_set_numerics(s, FLMR_NUMERICS_GPU_FP16) == FLMR_OK);
    REQUIRE(flmr_searcher_set_numerics(s, 7) != FLMR_OK);
    REQUIRE(flmr_searcher_set_profiling(s, 1) == FLMR_OK);
    { flmr_search_params_t p = {10, 2, 0.45f, 256, 32}; int32_t *dP, *dC; float* dS;
      HIPOK(hipMalloc(reinterpret_cast<void**>(&dP), nqueries * 10 * 4)); HIPOK(hipMalloc(reinterpret_cast<void**>(&dS), nqueries * 10 * 4));
      HIPOK(hipMalloc(reinterpret_cast<void**>(&dC), nqueries * 4));
      REQUIRE(flmr_search_batch(s, dQ, dL, nqueries, nq, &p, dP, dS, dC, nullptr) == FLMR_OK);
      REQUIRE(flmr_searcher_check(s) == FLMR_OK);
      float ms[16] = {};
      REQUIRE(flmr_searcher_stage_ms(s, ms) == FLMR_OK);
      HIPOK(hipFree(dP)); HIPOK(hipFree(dS)); HIPOK(hipFree(dC)); }
    // A query with MORE surviving centroids than the searcher keeps score rows for is a legal query (the reference has no such
    // limit, TPC/search/index_storage.py:116): a searcher created under FLMR_ROW_CAP=64 must return, through this same C ABI and
    // without any error, exactly what the uncapped searcher returns (the library recomputes that query's stage 1 itself).
    { flmr_search_params_t p = {32, 2, 0.2f, 256, 32};   // a low threshold: hundreds of centroids above it
      int32_t *dP[2], *dC[2]; float* dS[2];
      std::vector<int32_t> P[2], Cn[2]; std::vector<float> S[2];
      int over = 0;
      for (int capped = 0; capped < 2; capped++) {
          REQUIRE(flmr_set_option("FLMR_ROW_CAP", capped ? "64" : nullptr) == FLMR_OK);
          flmr_searcher_t* s2 = nullptr;
          REQUIRE(flmr_searcher_create(ix, 32, nq, &maxp, &s2) == FLMR_OK && s2);
          HIPOK(hipMalloc(reinterpret_cast<void**>(&dP[capped]), (size_t)nqueries * p.k * 4));
          HIPOK(hipMalloc(reinterpret_cast<void**>(&dS[capped]), (size_t)nqueries * p.k * 4));
          HIPOK(hipMalloc(reinterpret_cast<void**>(&dC[capped]), (size_t)nqueries * 4));
          REQUIRE(flmr_search_batch(s2, dQ, dL, nqueries, nq, &p, dP[capped], dS[capped], dC[capped], nullptr) == FLMR_OK);
          REQUIRE(flmr_searcher_check(s2) == FLMR_OK);   // no deferred capacity error either
          P[capped].resize((size_t)nqueries * p.k); S[capped].resize((size_t)nqueries * p.k); Cn[capped].resize(nqueries);
          HIPOK(hipMemcpy(P[capped].data(), dP[capped], P[capped].size() * 4, hipMemcpyDeviceToHost));
          HIPOK(hipMemcpy(S[capped].data(), dS[capped], S[capped].size() * 4, hipMemcpyDeviceToHost));
          HIPOK(hipMemcpy(Cn[capped].data(), dC[capped], Cn[capped].size() * 4, hipMemcpyDeviceToHost));
          if (capped)
              for (int q = 0; q < nqueries; q++) {
                  int32_t form = -1; int64_t cnt = 0;
                  REQUIRE(flmr_searcher_tap(s2, FLMR_TAP_STAGE1_FORM, q, &form, 1, &cnt) == FLMR_OK);
                  over += (cnt == 1 && form == 7) ? 1 : 0;
              }
          REQUIRE(flmr_searcher_destroy(s2) == FLMR_OK);
          HIPOK(hipFree(dP[capped])); HIPOK(hipFree(dS[capped])); HIPOK(hipFree(dC[capped]));
      }
      REQUIRE(flmr_set_option("FLMR_ROW_CAP", nullptr) == FLMR_OK);
      REQUIRE(over >= nqueries / 2);   // the capped searcher really was over its capacity
      REQUIRE(Cn[0] == Cn[1] && P[0] == P[1]);
      REQUIRE(memcmp(S[0].data(), S[1].data(), S[0].size() * 4) == 0); }
    REQUIRE(flmr_set_option("FLMR_NO_SUCH_SWITCH", "x") != FLMR_OK);
    REQUIRE(flmr_set_option("FLMR_S2_IMPL", "lds") == FLMR_OK);
    REQUIRE(flmr_set_option("FLMR_S2_IMPL", nullptr) == FLMR_OK);
    HIPOK(hipFree(dQ)); HIPOK(hipFree(dL));
    REQUIRE(flmr_searcher_destroy(s) == FLMR_OK);
    REQUIRE(flmr_index_close(ix) == FLMR_OK);
    printf("case K=%d nbits=%d passages=%lld tokens=%lld: ok (workspace %.1f MB, derived %.1f MB)\n", K, nbits, (long long)npass, (long long)h.ntok,
           ws / 1e6, info.derived_bytes / 1e6);
}

int main() {
    int n = 0;
    REQUIRE(flmr_device_count(&n) == FLMR_OK && n >= 1);
    REQUIRE(flmr_abi_version() == FLMR_ABI_VERSION);
    run_case(1024, 2, 4000, 11);
    run_case(2048, 8, 2500, 12);
    run_case(512, 4, 40000, 13);   // more than one 32768-passage chunk
    printf("abi harness: all cases passed\n");
    return 0;
}
