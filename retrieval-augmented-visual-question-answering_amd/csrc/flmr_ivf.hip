// IVF build: for every centroid the ascending unique pids of the passages that hold a token assigned to it.
// Reference: optimize_ivf, TPC/indexing/utils.py:8-53 -- the embedding ids sorted by code (collection_indexer.py:388-426:
// codes.sort()), mapped to pids through the doclens, `torch.unique` per centroid's slice.
//
// Tokens are stored in passage order, so a STABLE sort of (code, pid) pairs by code alone leaves every centroid's slice
// ascending in pid with the repeats of a passage adjacent.  The sort is hand-written for exactly this job (round 5; rounds 3-4
// called rocPRIM's radix sort through hipCUB): an LSD counting sort over ceil(log2 K) <= 18 key bits in one or two 9-bit passes.
// One pass:
//   ivf_hist_kernel     a workgroup counts the digits of its tile of 16384 keys in LDS -> hist[digit][tile]
//   ivf_scan_*          exclusive prefix over that array, digit-major: where each (digit, tile) run starts in the output
//   ivf_scatter_kernel  the workgroup counts again per WAVE (each wave owns a contiguous quarter of the tile), turns the counts
//                       into every wave's start per digit, then each wave walks its quarter 64 keys at a time: the lanes that
//                       hold the same digit find each other with nine ballots, a lane's place is the wave's running start for
//                       the digit + the number of such lanes below it, the lowest of them advances the start -- stable by
//                       construction (tile order = wave order = step order = lane order), no atomics on the output side.
// Then "first of its (code, pid) run" flags + the per-centroid counts, and a compaction of the flagged pids by the same
// count / prefix / place scheme.  Positions are 64-bit throughout: no 2^31 limit on tokens (pids are the index's int32).
// Temporaries: 20 bytes per token + 8 bytes per (digit, tile).
#include <hip/hip_runtime.h>

#include "flmr_common.h"

namespace {

constexpr int IVF_BITS = 9;                 // key bits per pass
constexpr int IVF_R = 1 << IVF_BITS;        // digit values
constexpr int IVF_TILE = 16384;             // keys per workgroup (256 threads: four waves x 4096)
constexpr int IVF_WCHUNK = IVF_TILE / 4;

// pid of every token: one wave per passage writes its run
__global__ __launch_bounds__(256) void ivf_token_pids_kernel(const int64_t* __restrict__ doc_offsets, int64_t num_passages,
                                                             int32_t* __restrict__ pid_of) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < num_passages; p += (int64_t)gridDim.x * 4) {
        const int64_t beg = doc_offsets[p], end = doc_offsets[p + 1];
        for (int64_t t = beg + lane; t < end; t += 64) pid_of[t] = (int32_t)p;
    }
}

__global__ __launch_bounds__(256) void ivf_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, int64_t ntiles,
                                                       unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[IVF_R];
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < IVF_R; d += 256) h[d] = 0;
        __syncthreads();
        const int64_t base = tile * IVF_TILE;
        for (int e = threadIdx.x; e < IVF_TILE; e += 256) {
            const int64_t t = base + e;
            if (t < n) atomicAdd(&h[(keys[t] >> shift) & (IVF_R - 1)], 1u);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < IVF_R; d += 256) hist[(int64_t)d * ntiles + tile] = h[d];
        __syncthreads();
    }
}

// ---- exclusive prefix over 64-bit counts, 2048 per workgroup, recursive over the workgroup totals ---------------------------
constexpr int IVF_SCAN_ITEMS = 8;           // per thread (256 threads: 2048 per workgroup)

__device__ __forceinline__ unsigned long long ivf_wave_incl(unsigned long long v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// data[i] <- sum of data[0 .. i) within the workgroup's 2048-entry segment; totals[block] <- the segment's sum
__global__ __launch_bounds__(256) void ivf_scan_local_kernel(unsigned long long* __restrict__ data, int64_t n,
                                                             unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long wsum[4];
    const int64_t base = (int64_t)blockIdx.x * (256 * IVF_SCAN_ITEMS) + (int64_t)threadIdx.x * IVF_SCAN_ITEMS;
    unsigned long long v[IVF_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < IVF_SCAN_ITEMS; k++) { v[k] = base + k < n ? data[base + k] : 0ull; s += v[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long inc = ivf_wave_incl(s, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < IVF_SCAN_ITEMS; k++) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) totals[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void ivf_scan_add_kernel(unsigned long long* __restrict__ data, int64_t n,
                                                           const unsigned long long* __restrict__ offsets) {
    const unsigned long long add = offsets[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * (256 * IVF_SCAN_ITEMS) + (int64_t)threadIdx.x * IVF_SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < IVF_SCAN_ITEMS; k++)
        if (base + k < n) data[base + k] += add;
}

// the lanes of the wave that hold the same 9-bit digit as this one (valid lanes only)
__device__ __forceinline__ unsigned long long ivf_peers(unsigned int digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < IVF_BITS; b++) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__global__ __launch_bounds__(256) void ivf_scatter_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                          int shift, int64_t ntiles, const unsigned long long* __restrict__ offs,
                                                          uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
    __shared__ unsigned long long start[4][IVF_R];    // per wave and digit: count, then the next output position
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < 4 * IVF_R; d += 256) (&start[0][0])[d] = 0ull;
        __syncthreads();
        const int64_t wbase = tile * IVF_TILE + (int64_t)wave * IVF_WCHUNK;
        for (int e = lane; e < IVF_WCHUNK; e += 64) {
            const int64_t t = wbase + e;
            if (t < n) atomicAdd(&start[wave][(keys[t] >> shift) & (IVF_R - 1)], 1ull);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < IVF_R; d += 256) {   // counts -> starts: the (digit, tile) run, cut by wave
            unsigned long long run = offs[(int64_t)d * ntiles + tile];
#pragma unroll
            for (int w = 0; w < 4; w++) { const unsigned long long c = start[w][d]; start[w][d] = run; run += c; }
        }
        __syncthreads();
        for (int e0 = 0; e0 < IVF_WCHUNK; e0 += 64) {
            const int64_t t = wbase + e0 + lane;
            const bool valid = t < n;
            const uint32_t key = valid ? keys[t] : 0u;
            const unsigned int digit = (key >> shift) & (IVF_R - 1);
            const unsigned long long peers = ivf_peers(digit, valid);
            if (valid) {
                const unsigned long long pos = start[wave][digit] + (unsigned long long)__popcll(peers & lt);
                keys_out[pos] = key;
                vals_out[pos] = vals[t];
            }
            __builtin_amdgcn_wave_barrier();   // (every lane has read its start before a group's lowest lane advances it)
            if (valid && (peers & lt) == 0ull) start[wave][digit] += (unsigned long long)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
}

// after the stable sort by code: entry i opens a new (code, pid) run?  + the centroid's list length and the tile's flag count
__global__ __launch_bounds__(256) void ivf_flag_kernel(const uint32_t* __restrict__ codes_sorted, const int32_t* __restrict__ pids_sorted,
                                                       int64_t n, int32_t K, int64_t ntiles, uint8_t* __restrict__ flags,
                                                       unsigned long long* __restrict__ lengths, unsigned long long* __restrict__ tile_count,
                                                       int32_t* __restrict__ bad) {
    __shared__ unsigned int cnt;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
        unsigned int mine = 0;
        for (int e = threadIdx.x; e < IVF_TILE; e += 256) {
            const int64_t i = tile * IVF_TILE + e;
            if (i >= n) break;
            const uint32_t c = codes_sorted[i];
            const bool first = i == 0 || codes_sorted[i - 1] != c || pids_sorted[i - 1] != pids_sorted[i];
            flags[i] = first ? 1 : 0;
            if (c >= (uint32_t)K) { atomicExch(bad, 1); continue; }
            if (first) { atomicAdd(&lengths[c], 1ull); mine++; }
        }
        atomicAdd(&cnt, mine);
        __syncthreads();
        if (threadIdx.x == 0) tile_count[tile] = cnt;
        __syncthreads();
    }
}

// the flagged pids, in order, to ivf[tile_start[tile] ..]
__global__ __launch_bounds__(256) void ivf_compact_kernel(const int32_t* __restrict__ pids_sorted, const uint8_t* __restrict__ flags, int64_t n,
                                                          int64_t ntiles, const unsigned long long* __restrict__ tile_start,
                                                          int32_t* __restrict__ ivf) {
    __shared__ unsigned int wcount[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        unsigned long long run = tile_start[tile];
        for (int e0 = 0; e0 < IVF_TILE; e0 += 256) {   // 256 entries per step: wave order, then lane order
            const int64_t i = tile * IVF_TILE + e0 + threadIdx.x;
            const bool f = i < n && flags[i] != 0;
            const unsigned long long m = __ballot(f);
            if (lane == 0) wcount[wave] = (unsigned int)__popcll(m);
            __syncthreads();
            unsigned long long pos = run + (unsigned long long)__popcll(m & lt);
            for (int w = 0; w < wave; w++) pos += wcount[w];
            if (f) ivf[pos] = pids_sorted[i];
            run += (unsigned long long)wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        }
    }
}

struct scratch {
    void* p = nullptr;
    ~scratch() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { FLMR_HIP(hipMalloc(&p, bytes ? bytes : 1)); return FLMR_OK; }
};

#define RUN(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// in-place exclusive prefix of `n` 64-bit counts on the device (recursion over the per-workgroup totals)
static int ivf_exclusive_scan(unsigned long long* data, int64_t n, hipStream_t st) {
    if (n <= 0) return FLMR_OK;
    const int64_t per = 256 * IVF_SCAN_ITEMS, nb = (n + per - 1) / per;
    scratch totals;
    RUN(totals.alloc((size_t)nb * 8));
    hipLaunchKernelGGL(ivf_scan_local_kernel, dim3((unsigned)nb), dim3(256), 0, st, data, n, static_cast<unsigned long long*>(totals.p));
    FLMR_LAUNCH_CHECK();
    if (nb > 1) {
        RUN(ivf_exclusive_scan(static_cast<unsigned long long*>(totals.p), nb, st));
        hipLaunchKernelGGL(ivf_scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, st, data, n, static_cast<const unsigned long long*>(totals.p));
        FLMR_LAUNCH_CHECK();
    }
    FLMR_HIP(hipStreamSynchronize(st));   // (`totals` is freed on return)
    return FLMR_OK;
}

}  // namespace

extern "C" int flmr_build_ivf(const int32_t* codes, int64_t n_tokens, const int64_t* doc_offsets, int64_t num_passages, int32_t K,
                              int32_t* ivf_pids, int64_t* ivf_lengths, int64_t* total, flmr_stream_t stream) {
    if ((!codes && n_tokens > 0) || !doc_offsets || !ivf_pids || !ivf_lengths || !total) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (n_tokens < 0 || num_passages < 0 || K < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    if (num_passages > 0x7fffffffll) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "flmr_build_ivf: more than 2^31 - 1 passages (pids are int32)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FLMR_HIP(hipMemsetAsync(ivf_lengths, 0, (size_t)K * sizeof(int64_t), st));
    *total = 0;
    if (n_tokens == 0) { FLMR_HIP(hipStreamSynchronize(st)); return FLMR_OK; }
    const int64_t n = n_tokens, ntiles = (n + IVF_TILE - 1) / IVF_TILE;
    int bits = 1;
    while ((1ll << bits) < (int64_t)K) bits++;
    if (bits > 2 * IVF_BITS) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "flmr_build_ivf: K = %d needs more than %d key bits", K, 2 * IVF_BITS);
    const int npass = bits > IVF_BITS ? 2 : 1;
    scratch pid_in, key_a, pid_a, key_b, pid_b, hist, flg, tcount, bad;
    RUN(pid_in.alloc((size_t)n * 4)); RUN(key_a.alloc((size_t)n * 4)); RUN(pid_a.alloc((size_t)n * 4));
    if (npass == 2) { RUN(key_b.alloc((size_t)n * 4)); RUN(pid_b.alloc((size_t)n * 4)); }
    RUN(hist.alloc((size_t)IVF_R * ntiles * 8)); RUN(flg.alloc((size_t)n)); RUN(tcount.alloc((size_t)(ntiles + 1) * 8)); RUN(bad.alloc(4));
    FLMR_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    const unsigned grid = (unsigned)(ntiles < 8192 ? ntiles : 8192);
    hipLaunchKernelGGL(ivf_token_pids_kernel, dim3(4096), dim3(256), 0, st, doc_offsets, num_passages, static_cast<int32_t*>(pid_in.p));
    FLMR_LAUNCH_CHECK();
    const uint32_t* kin = reinterpret_cast<const uint32_t*>(codes);   // (codes are >= 0: the unsigned order is theirs; a code >= K is reported below)
    const int32_t* vin = static_cast<const int32_t*>(pid_in.p);
    uint32_t* kout = static_cast<uint32_t*>(key_a.p);
    int32_t* vout = static_cast<int32_t*>(pid_a.p);
    for (int pass = 0; pass < npass; pass++) {
        const int shift = pass * IVF_BITS;
        hipLaunchKernelGGL(ivf_hist_kernel, dim3(grid), dim3(256), 0, st, kin, n, shift, ntiles, static_cast<unsigned long long*>(hist.p));
        FLMR_LAUNCH_CHECK();
        RUN(ivf_exclusive_scan(static_cast<unsigned long long*>(hist.p), (int64_t)IVF_R * ntiles, st));
        hipLaunchKernelGGL(ivf_scatter_kernel, dim3(grid), dim3(256), 0, st, kin, vin, n, shift, ntiles,
                           static_cast<const unsigned long long*>(hist.p), kout, vout);
        FLMR_LAUNCH_CHECK();
        kin = kout; vin = vout;
        kout = static_cast<uint32_t*>(key_b.p); vout = static_cast<int32_t*>(pid_b.p);
    }
    // (keys above 2^18 would not be ordered by two 9-bit passes: they are invalid codes anyway, flagged here)
    hipLaunchKernelGGL(ivf_flag_kernel, dim3(grid), dim3(256), 0, st, kin, vin, n, K, ntiles, static_cast<uint8_t*>(flg.p),
                       reinterpret_cast<unsigned long long*>(ivf_lengths), static_cast<unsigned long long*>(tcount.p), static_cast<int32_t*>(bad.p));
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipMemsetAsync(static_cast<unsigned long long*>(tcount.p) + ntiles, 0, 8, st));
    RUN(ivf_exclusive_scan(static_cast<unsigned long long*>(tcount.p), ntiles + 1, st));   // [ntiles] = the total
    int host_bad = 0;
    unsigned long long host_n = 0;
    FLMR_HIP(hipMemcpyAsync(&host_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    FLMR_HIP(hipMemcpyAsync(&host_n, static_cast<unsigned long long*>(tcount.p) + ntiles, 8, hipMemcpyDeviceToHost, st));
    FLMR_HIP(hipStreamSynchronize(st));
    if (host_bad) FLMR_FAIL(FLMR_ERR_INVALID, "flmr_build_ivf: a code outside [0, K=%d)", K);
    hipLaunchKernelGGL(ivf_compact_kernel, dim3(grid), dim3(256), 0, st, vin, static_cast<const uint8_t*>(flg.p), n, ntiles,
                       static_cast<const unsigned long long*>(tcount.p), ivf_pids);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipStreamSynchronize(st));   // (the scratch buffers are freed on return)
    *total = (int64_t)host_n;
    return FLMR_OK;
}
#undef RUN
