// IVF build: for every centroid the ascending unique pids of the passages that hold a token assigned to it.
// Reference: optimize_ivf, TPC/indexing/utils.py:8-53 -- the embedding ids sorted by code (collection_indexer.py:388-426:
// codes.sort()), mapped to pids through the doclens, `torch.unique` per centroid's slice.
//
// Tokens are stored in passage order, so a STABLE sort of (code, pid) pairs by code alone leaves every centroid's slice
// ascending in pid with the repeats of a passage adjacent: one radix sort over ceil(log2 K) key bits (rocPRIM's device radix
// sort through hipCUB -- the platform's primitive, the one piece of this library that is not a hand-written kernel; a
// counting sort by 17-bit keys is exactly what it runs), then "first of its (code, pid) run" flags, a per-centroid count of the
// flagged entries and a stream compaction.  Everything stays on the device; temporaries are 13 bytes per token.
#include <hip/hip_runtime.h>

#include <hipcub/device/device_radix_sort.hpp>
#include <hipcub/device/device_select.hpp>

#include "flmr_common.h"

namespace {

// pid of every token: one wave per passage writes its run
__global__ __launch_bounds__(256) void ivf_token_pids_kernel(const int64_t* __restrict__ doc_offsets, int64_t num_passages,
                                                             int32_t* __restrict__ pid_of) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < num_passages; p += (int64_t)gridDim.x * 4) {
        const int64_t beg = doc_offsets[p], end = doc_offsets[p + 1];
        for (int64_t t = beg + lane; t < end; t += 64) pid_of[t] = (int32_t)p;
    }
}

// after the stable sort by code: entry i opens a new (code, pid) run?  + the centroid's list length
__global__ __launch_bounds__(256) void ivf_flag_kernel(const uint32_t* __restrict__ codes_sorted, const int32_t* __restrict__ pids_sorted,
                                                       int64_t n, int32_t K, uint8_t* __restrict__ flags,
                                                       unsigned long long* __restrict__ lengths, int32_t* __restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t c = codes_sorted[i];
        const bool first = i == 0 || codes_sorted[i - 1] != c || pids_sorted[i - 1] != pids_sorted[i];
        flags[i] = first ? 1 : 0;
        if (c >= (uint32_t)K) { atomicExch(bad, 1); continue; }
        if (first) atomicAdd(&lengths[c], 1ull);
    }
}

struct scratch {
    void* p = nullptr;
    ~scratch() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { FLMR_HIP(hipMalloc(&p, bytes ? bytes : 1)); return FLMR_OK; }
};

}  // namespace

#define RUN(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
#define CUB(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) FLMR_FAIL(FLMR_ERR_HIP, "%s -> %s", #x, hipGetErrorString(e__)); } while (0)

extern "C" int flmr_build_ivf(const int32_t* codes, int64_t n_tokens, const int64_t* doc_offsets, int64_t num_passages, int32_t K,
                              int32_t* ivf_pids, int64_t* ivf_lengths, int64_t* total, flmr_stream_t stream) {
    if ((!codes && n_tokens > 0) || !doc_offsets || !ivf_pids || !ivf_lengths || !total) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (n_tokens < 0 || num_passages < 0 || K < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    if (n_tokens > 0x7fffffffll || num_passages > 0x7fffffffll) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "flmr_build_ivf: more than 2^31 - 1 tokens / passages");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FLMR_HIP(hipMemsetAsync(ivf_lengths, 0, (size_t)K * sizeof(int64_t), st));
    *total = 0;
    if (n_tokens == 0) { FLMR_HIP(hipStreamSynchronize(st)); return FLMR_OK; }
    const int n = (int)n_tokens;
    int bits = 1;
    while ((1ll << bits) < (int64_t)K) bits++;
    scratch pid_in, pid_out, key_out, flg, nsel, bad, tmp;
    RUN(pid_in.alloc((size_t)n * 4)); RUN(pid_out.alloc((size_t)n * 4)); RUN(key_out.alloc((size_t)n * 4));
    RUN(flg.alloc((size_t)n)); RUN(nsel.alloc(8)); RUN(bad.alloc(4));
    FLMR_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    hipLaunchKernelGGL(ivf_token_pids_kernel, dim3(4096), dim3(256), 0, st, doc_offsets, num_passages, static_cast<int32_t*>(pid_in.p));
    FLMR_LAUNCH_CHECK();
    size_t b1 = 0, b2 = 0;
    const uint32_t* kin = reinterpret_cast<const uint32_t*>(codes);   // (codes are >= 0: the unsigned order is theirs)
    CUB(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, kin, static_cast<uint32_t*>(key_out.p), static_cast<const int32_t*>(pid_in.p),
                                           static_cast<int32_t*>(pid_out.p), n, 0, bits, st));
    CUB(hipcub::DeviceSelect::Flagged(nullptr, b2, static_cast<const int32_t*>(pid_out.p), static_cast<const uint8_t*>(flg.p), ivf_pids,
                                      static_cast<int*>(nsel.p), n, st));
    RUN(tmp.alloc(b1 > b2 ? b1 : b2));
    CUB(hipcub::DeviceRadixSort::SortPairs(tmp.p, b1, kin, static_cast<uint32_t*>(key_out.p), static_cast<const int32_t*>(pid_in.p),
                                           static_cast<int32_t*>(pid_out.p), n, 0, bits, st));
    hipLaunchKernelGGL(ivf_flag_kernel, dim3(4096), dim3(256), 0, st, static_cast<const uint32_t*>(key_out.p),
                       static_cast<const int32_t*>(pid_out.p), (int64_t)n, K, static_cast<uint8_t*>(flg.p),
                       reinterpret_cast<unsigned long long*>(ivf_lengths), static_cast<int32_t*>(bad.p));
    FLMR_LAUNCH_CHECK();
    CUB(hipcub::DeviceSelect::Flagged(tmp.p, b2, static_cast<const int32_t*>(pid_out.p), static_cast<const uint8_t*>(flg.p), ivf_pids,
                                      static_cast<int*>(nsel.p), n, st));
    int host_n = 0, host_bad = 0;
    FLMR_HIP(hipMemcpyAsync(&host_n, nsel.p, 4, hipMemcpyDeviceToHost, st));
    FLMR_HIP(hipMemcpyAsync(&host_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    FLMR_HIP(hipStreamSynchronize(st));   // (the scratch buffers are freed on return)
    if (host_bad) FLMR_FAIL(FLMR_ERR_INVALID, "flmr_build_ivf: a code outside [0, K=%d)", K);
    *total = host_n;
    return FLMR_OK;
}
#undef RUN
#undef CUB
