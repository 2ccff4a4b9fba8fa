// Device-side helpers (wave64 reductions, block scans, small sorted lists) for the gfx950 kernels.
#pragma once
#include "flmr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FLMR_NEG_INF (-__builtin_huge_valf())

// ---- wave reductions by DPP (row maxima / minima: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror; the four rows
// through the scalar unit): ~10 instructions and no LDS round trip, against 6 x ds_bpermute for a shuffle butterfly ----------
__device__ __forceinline__ float flmr_wave_max_f32(float x) {
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, false)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int flmr_wave_min_i32(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

// ---- top-NC list ordered by (value desc, index asc) --------------------------------------------
// Used for the per-query-token probe of the `ncells` best centroids (candidate_generation.py:12-20).
template <int NC>
struct flmr_toplist {
    float v[NC];
    int id[NC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < NC; t++) { v[t] = FLMR_NEG_INF; id[t] = 0x7fffffff; }
    }
    // general insert: (value desc, index asc); branch-free (selects only) so it schedules next to MFMAs
    __device__ __forceinline__ void insert(float x, int i) {
#pragma unroll
        for (int t = NC - 1; t >= 0; --t) {
            const bool better = (x > v[t]) | ((x == v[t]) & (i < id[t]));
            if (t + 1 < NC) { v[t + 1] = better ? v[t] : v[t + 1]; id[t + 1] = better ? id[t] : id[t + 1]; }
            v[t] = better ? x : v[t];
            id[t] = better ? i : id[t];
        }
    }
    // insert for indices that arrive in ascending order: a strict '>' keeps the earlier (lower) index on ties
    __device__ __forceinline__ void insert_ascending(float x, int i) {
#pragma unroll
        for (int t = NC - 1; t >= 0; --t) {
            const bool better = x > v[t];
            if (t + 1 < NC) { v[t + 1] = better ? v[t] : v[t + 1]; id[t + 1] = better ? id[t] : id[t + 1]; }
            v[t] = better ? x : v[t];
            id[t] = better ? i : id[t];
        }
    }
    // top-NC of the union of the wave's 64 lists (entries distinct across lanes; finite values or -inf), left in EVERY lane: NC
    // rounds of (wave maximum of the heads, lowest index among the lanes holding it, pop) -- ~25 instructions a round against
    // six butterfly merges of 2 NC shuffles + NC inserts each.  Same result as merge_xor(32) ... merge_xor(1).
    __device__ __forceinline__ void merge_wave() {
        float ov[NC]; int oi[NC];
#pragma unroll
        for (int t = 0; t < NC; t++) {
            const float m = flmr_wave_max_f32(v[0]);
            const unsigned long long tie = __ballot(v[0] == m);   // (m = -inf: every exhausted lane holds (-inf, INT_MAX); harmless)
            int wid;
            if (__popcll(tie) == 1) wid = __builtin_amdgcn_readlane(id[0], (int)__builtin_ctzll(tie));   // wave-uniform branch
            else wid = flmr_wave_min_i32(v[0] == m ? id[0] : 0x7fffffff);
            ov[t] = m; oi[t] = wid;
            const bool win = v[0] == m && id[0] == wid;
#pragma unroll
            for (int u = 0; u + 1 < NC; u++) { v[u] = win ? v[u + 1] : v[u]; id[u] = win ? id[u + 1] : id[u]; }
            v[NC - 1] = win ? FLMR_NEG_INF : v[NC - 1];
            id[NC - 1] = win ? 0x7fffffff : id[NC - 1];
        }
#pragma unroll
        for (int t = 0; t < NC; t++) { v[t] = ov[t]; id[t] = oi[t]; }
    }
    // merge with the list held by lane (lane ^ mask)
    __device__ __forceinline__ void merge_xor(int mask) {
        float ov[NC]; int oi[NC];
#pragma unroll
        for (int t = 0; t < NC; t++) { ov[t] = __shfl_xor(v[t], mask, 64); oi[t] = __shfl_xor(id[t], mask, 64); }
#pragma unroll
        for (int t = 0; t < NC; t++) insert(ov[t], oi[t]);
    }
};

// ---- wave / block reductions ------------------------------------------------------------------
__device__ __forceinline__ float flmr_half_wave_max(float x) {  // max over the 32 lanes sharing lane>>5
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}
__device__ __forceinline__ int flmr_wave_inclusive_scan(int x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
// exclusive scan over a block of up to 1024 threads; `lds` needs 17 ints; returns exclusive prefix, sets total
__device__ __forceinline__ int flmr_block_exclusive_scan(int x, int* lds, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = flmr_wave_inclusive_scan(x, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = (lane < nw) ? lds[lane] : 0;
        int winc = flmr_wave_inclusive_scan(w, lane);
        if (lane < nw) lds[lane] = winc - w;
        if (lane == nw - 1) lds[16] = winc;
    }
    __syncthreads();
    int res = inc - x + lds[wave];
    *total = lds[16];
    __syncthreads();
    return res;
}

// ---- in-LDS bitonic sort, descending, n a power of two ------------------------------------------
template <typename T>
__device__ __forceinline__ void flmr_bitonic_sort_desc(T* s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n; t += blockDim.x) {
                int p = t ^ j;
                if (p > t) {
                    T a = s[t], b = s[p];
                    bool desc = ((t & k) == 0);
                    if (desc ? (a < b) : (a > b)) { s[t] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// The same sort for 64 <= n <= blockDim.x keys with ONE key per thread in a register: the steps whose partner sits in the same
// wave (j < 64: 21 + 6 per later level) are shuffles without a barrier; only the steps across waves go through LDS, one
// barrier each (two buffers in turn: `s` must hold 2 n entries).  n = 1024: 10 barriers instead of 55.  Leaves s[0 .. n) sorted
// and ends with a barrier, like the function above.
__device__ __forceinline__ void flmr_bitonic_sort_desc_reg(unsigned long long* s, int n) {
    const int t = threadIdx.x;
    const bool live = t < n;
    unsigned long long x = live ? s[t] : 0ull;
    int buf = 1;   // (s[0 .. n) holds the input until every thread has read its key: the first exchange writes the upper half)
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long y;
            if (j >= 64) {   // (block-uniform)
                unsigned long long* e = s + buf * n;
                if (live) e[t] = x;
                __syncthreads();
                y = live ? e[t ^ j] : 0ull;
                buf ^= 1;
            } else {
                y = __shfl_xor(x, j, 64);
            }
            const bool keep_max = ((t & j) == 0) == ((t & k) == 0);
            x = keep_max ? (x > y ? x : y) : (x < y ? x : y);
        }
    }
    __syncthreads();   // (the last reads of either buffer are done)
    if (live) s[t] = x;
    __syncthreads();
}
// bytes of dynamic LDS a kernel sorting npow2 keys with the dispatch below needs
static inline size_t flmr_sort_lds_bytes(int npow2) { return (size_t)(npow2 <= 1024 ? 2 * npow2 : npow2) * 8; }
__device__ __forceinline__ void flmr_sort_keys_desc(unsigned long long* s, int n) {
    if (n >= 64 && n <= (int)blockDim.x) flmr_bitonic_sort_desc_reg(s, n);
    else flmr_bitonic_sort_desc<unsigned long long>(s, n);
}

// Both half-waves' values of a register (lanes L and L ^ 32) without an LDS round trip: v_permlane32_swap exchanges the upper
// half of one copy with the lower half of another, leaving {v[L & 31], v[(L & 31) + 32]} in every lane (a VALU op; the
// ds_bpermute form of __shfl_xor(v, 32) costs an LDS-crossbar round trip on the critical path of every tile).
__device__ __forceinline__ void flmr_both_halves(float v, float& lo, float& hi) {
    float a = v, b = v;
    // (written as an instruction: the compiler's builtin returned the first result twice; two read-write operands also keep
    // the copies in two registers -- with one register for both the instruction swaps it with itself)
    // volatile: the instruction reads lanes of BOTH halves, so it must not be sunk into a branch that only one half takes
    // (its only consumer often is: `if (h == 0) store`)
    // The wait states are the compiler's job for instructions it knows; inside an asm block they are ours: the operands are
    // usually written by the VALU instruction just before (DPP-class hazard), and read by the one just after.
    asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
__device__ __forceinline__ float flmr_xhalf_max(float v) {
    float a, b, r;
    flmr_both_halves(v, a, b);
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // (fmaxf would first canonicalise both asm outputs: two more VALU ops)
    return r;
}
// max(a, b) as the instruction (no canonicalisation of the operands)
__device__ __forceinline__ float flmr_fmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The "keep one, send one" exchange across lane bit 4 of a transpose-reduce: a lane with bit 4 clear keeps x0 and needs its
// partner's (lane ^ 16) x0, a lane with bit 4 set keeps x1 and needs its partner's x1.  v_permlane16_swap_b32 a, b swaps lanes
// 16..31 of `a` with lanes 0..15 of `b` in each half-wave: afterwards `a` holds {own x0, partner's x1} in {low, high} rows and
// `b` {partner's x0, own x1} -- the maximum of the two is what both kinds of lane want.  One VALU instruction instead of two
// selects, an LDS-crossbar round trip and its wait.
__device__ __forceinline__ float flmr_x16_max(float x0, float x1) {
    float a = x0, b = x1;
    asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    return flmr_fmax_raw(a, b);
}
// the value of lane ^ 4 by two DPP moves: row_shl:4 (lane i reads lane i + 4) into the lanes with bit 2 clear -- banks 0 and 2 of a
// row of 16 -- and row_shr:4 into the others
__device__ __forceinline__ float flmr_dpp_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104 /* row_shl:4 */, 0xF, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x114 /* row_shr:4 */, 0xF, 0xA, false);
    return __int_as_float(t);
}
// (eight pairs at once: the wait states around the swaps are paid once)
__device__ __forceinline__ void flmr_x16_max8(float (&a)[8], float (&b)[8], float (&out)[8]) {
    asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %8\n\tv_permlane16_swap_b32 %1, %9\n\tv_permlane16_swap_b32 %2, %10\n\t"
                 "v_permlane16_swap_b32 %3, %11\n\tv_permlane16_swap_b32 %4, %12\n\tv_permlane16_swap_b32 %5, %13\n\t"
                 "v_permlane16_swap_b32 %6, %14\n\tv_permlane16_swap_b32 %7, %15\n\ts_nop 3"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#pragma unroll
    for (int r = 0; r < 8; r++) out[r] = flmr_fmax_raw(a[r], b[r]);
}
__device__ __forceinline__ float flmr_xhalf_sum(float v) { float a, b; flmr_both_halves(v, a, b); return a + b; }  // = v + other half's v

// ---- the reference's CUDA-path numerics (FLMR_NUMERICS_GPU_FP16; index_storage.py:113-149, colbert.py:235-263) ------------
// There the centroid scores are an fp16 tensor, padding is -9999 stored in fp16 (= -10000) and `.sum(-1)` of an fp16 tensor
// accumulates in fp32 and rounds the result to fp16 (overflow -> -inf).  Rounding is monotone, so max over a column of
// rounded values = the rounded maximum: the kernels keep their fp32 column maxima and round ONCE per (passage, column) where
// the maxima are summed.  `f16` is a kernel argument (wave-uniform).
__device__ __forceinline__ float flmr_round_f16(float x) { return (float)(_Float16)x; }   // RNE; |x| > 65504 -> +-inf
__device__ __forceinline__ float flmr_pad_score(int f16) { return f16 ? -10000.0f : -9999.0f; }
__device__ __forceinline__ float flmr_miss_score(int nqc, int f16) {   // a passage with no qualifying code: nqc x the padding value
    float s = 0.0f;
    for (int q = 0; q < nqc; q++) s += flmr_pad_score(f16);
    return f16 ? flmr_round_f16(s) : s;
}

// sequential fp32 sum of per-column maxima, the reference's `score += per_doc_approx_scores[k]` order
// (filter_pids.cpp:59-63): kept strictly k-ascending so pruning decisions are bit-identical to the CPU path.
__device__ __forceinline__ float flmr_seq_sum(const float* v, int n, int f16 = 0) {
    float s = 0.0f;
    if (f16) {   // fp16 column maxima, fp32 accumulation, fp16 result
        for (int k = 0; k < n; k++) s += flmr_round_f16(v[k]);
        return flmr_round_f16(s);
    }
    if (n <= 32) {
        // the common case (one column tile): all loads are issued before the first add -- a loop of dependent LDS reads costs
        // one LDS round trip per column (~3000 cycles per passage, as much as a token tile), the adds alone ~250.  Columns
        // >= n contribute +0.0f, which leaves the sum's bits unchanged (x + 0.0f == x for every x but -0.0f, which a sum
        // that started from +0.0f never is).
        float x[32];
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = k < n ? v[k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 32; k++) s += x[k];
        return s;
    }
    for (int k = 0; k < n; k++) s += v[k];
    return s;
}
