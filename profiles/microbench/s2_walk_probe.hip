// Stand-alone timing harness for the stage-2 table walk (csrc/flmr_stage2_walk.hip) on synthetic inputs of BASELINE's shape:
// K = 131072 fp16 centroids, 1024 queries x 1024 survivors x 128 uniformly random codes.  Built with -DW2_PROFILE it also
// prints how a wave's cycles split over the three phases of a slice (MFMA + stores, consume, barrier wait).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW2_PROFILE -Iinclude -Iretrieval-augmented-visual-question-answering_amd/csrc \
//         -o profiles/microbench/s2_walk_probe profiles/microbench/s2_walk_probe.hip
#include "../../retrieval-augmented-visual-question-answering_amd/csrc/flmr_stage2_walk.hip"

#include <algorithm>
#include <random>
#include <vector>

thread_local char flmr_err_buf[512] = {0};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int K = 131072, P = 200000, L = 128, NQ = argc > 1 ? atoi(argv[1]) : 1024, ND = 1024;
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
    int32_t *d_codes, *d_pids, *d_counts; int64_t* d_off; _Float16 *d_cen, *d_qh, *d_ql; uint64_t* d_keys; long long* d_prof;
    CK(hipMalloc(&d_codes, codes.size() * 4)); CK(hipMemcpy(d_codes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_off, off.size() * 8)); CK(hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pids, pids.size() * 4)); CK(hipMemcpy(d_pids, pids.data(), pids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counts, counts.size() * 4)); CK(hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cen, cen.size() * 2)); CK(hipMemcpy(d_cen, cen.data(), cen.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_qh, qh.size() * 2)); CK(hipMemcpy(d_qh, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ql, ql.size() * 2)); CK(hipMemcpy(d_ql, ql.data(), ql.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_keys, (size_t)NQ * ND * 8)); CK(hipMalloc(&d_prof, 64)); CK(hipMemset(d_prof, 0, 64));
    flmr_filter_args f{};
    f.K = K; f.ncol = 32; f.nq_cand = 32; f.nqueries = NQ; f.q_lens = nullptr; f.codes = d_codes; f.doclens = nullptr; f.offsets = d_off;
    const size_t lds = (size_t)W2_BUF * sizeof(int) + 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_stage2_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = NQ < cu_count() ? NQ : cu_count();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(d_prof, 0, 64));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(filter_stage2_walk_kernel, dim3(grid), dim3(64 * W2_WAVES), lds, 0, f, d_pids, (int64_t)ND, d_counts, d_keys,
                           (int64_t)ND, d_cen, d_qh, d_ql, d_codes, 1, NQ
#ifdef W2_PROFILE
                           , d_prof
#endif
        );
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long prof[3] = {0, 0, 0};
        CK(hipMemcpy(prof, d_prof, 24, hipMemcpyDeviceToHost));
        const double waves = (double)grid * W2_WAVES, slices = (double)(K / W2_SLICE) * ((double)NQ / grid);
        printf("walk: %d queries  %.3f ms", NQ, ms);
#ifdef W2_PROFILE
        printf("   cycles per slice per wave: produce %.0f  consume %.0f  barrier %.0f", prof[0] / waves / slices, prof[1] / waves / slices,
               prof[2] / waves / slices);
#endif
        printf("\n");
    }
    uint64_t k0; CK(hipMemcpy(&k0, d_keys, 8, hipMemcpyDeviceToHost));
    printf("key[0] = %llx\n", (unsigned long long)k0);
    return 0;
}
