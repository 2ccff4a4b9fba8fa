// Stand-alone timing harness for the XCD-sliced stage 2 (csrc/flmr_stage2_xcd.hip) on synthetic inputs of BASELINE's shape:
// K = 131072 fp16 centroids, 1024 queries x 1024 survivors x 128 uniformly random codes.  Built with -DX2_PROFILE it also
// prints how a wave's cycles split over the phases of a step.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX2_PROFILE -Iinclude -Iretrieval-augmented-visual-question-answering_amd/csrc \
//         -o profiles/microbench/s2_xcd_probe profiles/microbench/s2_xcd_probe.hip
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_stage2_xcd.hip"

#include <algorithm>
#include <random>
#include <vector>

thread_local char flmr_err_buf[512] = {0};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int K = argc > 2 ? atoi(argv[2]) : 131072, P = 200000, L = 128, NQ = argc > 1 ? atoi(argv[1]) : 1024, ND = 1024;
    std::mt19937 rng(1);
    std::vector<int32_t> codes((size_t)P * L + 8, 0x7f7f7f7f);
    for (int p = 0; p < P; p++) {
        for (int t = 0; t < L; t++) codes[(size_t)p * L + t] = (int32_t)(rng() % K);
        std::sort(codes.begin() + (size_t)p * L, codes.begin() + (size_t)(p + 1) * L);
    }
    std::vector<int64_t> off(P + 1);
    for (int p = 0; p <= P; p++) off[p] = (int64_t)p * L;
    std::vector<int32_t> pids((size_t)NQ * ND), counts(NQ, ND);
    for (auto& x : pids) x = (int32_t)(rng() % P);
    std::vector<_Float16> cen((size_t)K * 128), qh((size_t)NQ * 32 * 128), ql(qh.size());
    for (auto& x : cen) x = (_Float16)((float)(rng() % 2001 - 1000) / 8000.0f);
    for (auto& x : qh) x = (_Float16)((float)(rng() % 2001 - 1000) / 8000.0f);
    for (auto& x : ql) x = (_Float16)((float)(rng() % 2001 - 1000) / 8000.0f);
    flmr_index ix{};
    ix.K = K; ix.N = (int64_t)P * L; ix.num_passages = P; ix.max_doclen = L;
    int32_t *d_pids, *d_counts; _Float16 *d_qh, *d_ql; uint64_t* d_keys; float* d_part;
    CK(hipMalloc(&ix.codes_sorted, codes.size() * 4)); CK(hipMemcpy(ix.codes_sorted, codes.data(), codes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.doc_offsets, off.size() * 8)); CK(hipMemcpy(ix.doc_offsets, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pids, pids.size() * 4)); CK(hipMemcpy(d_pids, pids.data(), pids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counts, counts.size() * 4)); CK(hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ix.centroids_f16, cen.size() * 2)); CK(hipMemcpy(ix.centroids_f16, cen.data(), cen.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_qh, qh.size() * 2)); CK(hipMemcpy(d_qh, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ql, ql.size() * 2)); CK(hipMemcpy(d_ql, ql.data(), ql.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_keys, (size_t)NQ * ND * 8));
    if (flmr_build_doc_splits(&ix) != 0 || !ix.doc_splits) { printf("split table failed: %s\n", flmr_err_buf); return 1; }
    printf("slices %d, XCD round-robin dispatch confirmed: %d\n", ix.nslices, ix.xcd_round_robin);
    CK(hipMalloc(&d_part, flmr_stage2_xcd_part_floats(&ix, NQ, ND) * sizeof(float)));
#ifdef X2_PROFILE
    CK(hipMalloc(&x2_prof_buffer, 128));
#endif
    flmr_filter_args f{};
    f.K = K; f.ncol = 32; f.nq_cand = 32; f.nqueries = NQ; f.q_lens = nullptr; f.codes = nullptr; f.doclens = nullptr; f.offsets = ix.doc_offsets;
    {
        int nb = 0;
#ifdef X2_ROWS_IN_REGS
        const size_t lds = (size_t)X2_WAVES * X2R_WAVE_LDS;
        printf("rows in registers: %d tiles in flight per wave, launch bound %d waves per SIMD\n", X2_ROWS_IN_REGS, X2_REG_MINW);
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, filter_stage2_xreg_kernel<true>, 64 * X2_WAVES, lds));
#else
        const size_t lds = (size_t)X2_WAVES * X2_WAVE_LDS;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_stage2_xcd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, filter_stage2_xcd_kernel<false>, 64 * X2_WAVES, lds));
#endif
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        printf("dynamic LDS per block %zu B, resident blocks per CU %d (LDS per CU %zu B, per block max %zu B)\n", lds, nb,
               (size_t)prop.maxSharedMemoryPerMultiProcessor, (size_t)prop.sharedMemPerBlockOptin);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool hi_only = argc > 3 && atoi(argv[3]) != 0;   // the approximate pass of "hi first" (one product)
    printf("%s\n", hi_only ? "hi-only instantiation" : "both products");
    for (int rep = 0; rep < 3; rep++) {
#ifdef X2_PROFILE
        CK(hipMemset(x2_prof_buffer, 0, 128));
#endif
        CK(hipEventRecord(e0, 0));
        if (flmr_launch_filter_stage2_xcd_ex(f, d_pids, ND, d_counts, ND, d_keys, ND, &ix, d_qh, d_ql, d_part, ND, hi_only, 0) != 0) { printf("launch failed: %s\n", flmr_err_buf); return 1; }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("K=%d nq=%d: sliced stage 2 + combine %.3f ms\n", K, NQ, ms);
#ifdef X2_PROFILE
        long long pr[10];
        CK(hipMemcpy(pr, x2_prof_buffer, 80, hipMemcpyDeviceToHost));
        const char* names[8] = {"set-up (per wave)", "wait rows", "LDS read rows", "codes: position + DMA issue", "wait codes", "rows: ring read + DMA issue", "MFMA + octet maxima", "fold into passages"};
        const double tiles = (double)pr[8], waves = (double)pr[9];
        printf("  waves %.0f, tiles/wave %.1f\n", waves, tiles / waves);
        double tot = 0;
        for (int k = 0; k < 8; k++) tot += (double)pr[k];
        for (int k = 0; k < 8; k++)
            printf("  %-30s %9.0f cycles per %s  (%4.1f %%)\n", names[k], k == 0 ? pr[k] / waves : pr[k] / tiles, k == 0 ? "wave" : "tile", 100.0 * pr[k] / tot);
        printf("  total per wave %.0f cycles\n", tot / waves);
#endif
    }
    return 0;
}
