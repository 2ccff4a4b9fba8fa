// Stage 1 for DENSE survivor sets (more surviving centroids than the list-scatter forms of flmr_candidates.hip take): the
// centroid-only MaxSim of every candidate restricted to the centroids above the threshold.
//
// Reference: TPC/search/filter_pids.cpp -- maxsim() :27-69 with the `idx` mask of TPC/search/index_storage.py:116,
// filter_pids_helper() :71-124 (top-ndocs by (score, pid)).
//   per candidate:  per_k = max over the DISTINCT surviving codes of the passage of centroid_scores[code][k]  (init -9999)
//                   score = sum_k per_k, k ascending in fp32 (:59-63)
//
// What the round-5 scan (filter_stage1_kernel) cost on an index built from overlapping clusters (1 M passages, ~1.5 k surviving
// centroids, 31.5 k candidates per query, 43 hits per candidate): 29 ms per 1024 queries, of which 15 ms "reading the
// candidates' codes".  profiles/microbench/gather_probe shows that was never a memory floor: the chip delivers 6.3-7.4 TB/s
// for saturated random 256/512-byte records (16 GB per step = 2.5 ms).  The time went into the scan's structure -- one
// round of 8 passages per wave at a time behind a dependent pid -> offset -> codes chain, a fresh 128-byte fp32 score row from
// L2 per hit, duplicates of a code scored again.  This file restates the scan around three facts:
//   1. DISTINCT codes.  A maximum is idempotent (filter_pids.cpp:50-63 skips repeated codes for the same reason), and the
//      index keeps a per-passage ascending copy of the codes whose distinct values come first (`codes_sorted` + `doc_ulen`,
//      built at flmr_index_open): 57 instead of 128 codes per passage on that index.
//   2. ROWS IN LDS.  A query's surviving rows are few (<= ~2 k) but 32 x fp32 = 128 bytes each: too big for LDS as they are.
//      Their fp16 images ROUNDED UP fit (64 bytes a row, 2032 rows at K = 131072).  max commutes with a monotone rounding, so
//      the column maxima of images are the images of the column maxima, and U(p) = sum_k up16(per_k) satisfies
//      S(p) - eps <= U(p) <= S(p) + E for the exact score S, with E = 32 x (largest rounding step of the query's rows) + eps
//      and eps a bound on the two summations' roundoff: both computed per query while the images are made.
//   3. APPROXIMATE, THEN REFINE (as stage 2 does with its hi products, flmr_filter.hip): with u* = the ndocs-th largest U,
//      at least ndocs passages have S >= u* - E, so the ndocs-th largest exact score is >= u* - E and every passage of the exact
//      top ndocs has U >= u* - E - eps.  That BAND (ndocs + a few per cent) is rescored exactly -- fp32 rows, ascending-k sum --
//      by the second form of this kernel, and the top ndocs of the band by exact (score, pid) key is, as a set with its keys,
//      bit for bit what the full scan selects.
//
// Two kernels (s1_image_kernel, s1_exact_kernel):
//   IMG   = the approximate pass: rows' fp16 images in LDS, packed fp16 maxima, U keys for every candidate;
//   EXACT = fp32 rows gathered through L2 (8 lanes x 16 bytes per row, a batch in flight), per-column maxima summed k-ascending
//           along a DPP chain: the reference's arithmetic.  Runs over the band of an IMG query, or over the whole candidate
//           list of a query whose rows do not fit LDS (mode 2: also what replaces the round-5 scan there).
//
// MI355X shape: one persistent 16-wave workgroup per CU (the images take the LDS), item = (query, part of its candidates),
// items dealt XCD-aware (workgroup L runs on XCD L % 8: a query's parts share an L2 for its rows).  A wave takes 64
// candidates at a time (lane = candidate for pid / offset / length, the next group's prefetched), and inside a group rounds
// of 64 / LPC candidates, LPC lanes x 4 codes each (LPC = 16 or 32 by the index's mean number of distinct codes); the codes of
// round r + 2 are requested when round r's have been listed.  Listing: one probe of the query's K-bit mask in LDS per code,
// hits compacted by a DPP row scan into a per-wave list of row ids (rank = prefix[word] + popcount below).
#include "flmr_device.h"

#define D1_WAVES 16
#ifdef D1_NT_CODES   // development switch: codes by non-temporal loads (gather_probe: plain loads are the faster ones)
#define D1_LOAD(p) __builtin_nontemporal_load(p)
#else
#define D1_LOAD(p) (*(p))
#endif
#define D1_THREADS (64 * D1_WAVES)
#define D1_CODE_PAD FLMR_CODE_PAD   // ints readable beyond the last passage's codes (flmr_build_sorted_codes pads the copy)
#define D1_XCDS 8

typedef uint32_t d1u4 __attribute__((ext_vector_type(4)));
typedef uint32_t d1u2 __attribute__((ext_vector_type(2)));
typedef int d1i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 d1h2 __attribute__((ext_vector_type(2)));
typedef float d1f4 __attribute__((ext_vector_type(4)));

// DPP helpers (row = 16 lanes).  bound_ctrl = true: lanes without a source read 0.
#define D1_DPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xF, 0xF, true)
#define D1_ROW_SHR(n) (0x110 + (n))
#define D1_ROW_ROR(n) (0x120 + (n))

// smallest fp16 >= x (x finite, |x| <= 65504)
__device__ __forceinline__ _Float16 d1_up16(float x) {
    _Float16 h = (_Float16)x;   // RNE
    if ((float)h < x) {         // step to the next value above
        uint16_t b = __builtin_bit_cast(uint16_t, h);
        if (b == 0x8000u) b = 0x0001u;             // -0 -> smallest positive subnormal
        else if (b & 0x8000u) b--;                 // negative: smaller magnitude
        else b++;                                  // positive: larger magnitude
        h = __builtin_bit_cast(_Float16, b);
    }
    // no subnormal images: the image sums are v_dot2_f32_f16 instructions, which may flush them -- a positive one becomes the
    // smallest normal number (still >= x; the step enters the query's E like any other), a negative one -0 (>= x)
    const uint16_t hb = __builtin_bit_cast(uint16_t, h);
    if ((hb & 0x7C00u) == 0u) h = __builtin_bit_cast(_Float16, (uint16_t)(x > 0.0f ? 0x0400u : 0x8000u));
    return h;
}

// (written as the instruction: the builtin maximum first canonicalises each operand -- three instructions instead of one)
__device__ __forceinline__ uint32_t d1_pk_max(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// LDS byte address of the row whose id is the low (hi = 0) or high (hi = 1) half of q: 64 * id + base
__device__ __forceinline__ uint32_t d1_row_addr(uint32_t q, int hi, uint32_t base) {
    uint32_t r;
    if (hi) asm("v_mad_u32_u16 %0, %1, 64, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(q), "v"(base));
    else asm("v_mad_u32_u16 %0, %1, 64, %2" : "=v"(r) : "v"(q), "v"(base));
    return r;
}

// byte offset of fp32 row `id` (the low / high half of q) from the query's rows: 128 * id + base
__device__ __forceinline__ uint32_t d1x_row_off(uint32_t q, int hi, uint32_t base) {
    uint32_t r;
    const uint32_t rb = 128u;   // (no literal operand in a VOP3 of this family)
    if (hi) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(q), "s"(rb), "v"(base));
    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(rb), "v"(base));
    return r;
}
// (written as the instructions: the builtin maximum first canonicalises operands it cannot prove canonical -- a v_max x, x each)
__device__ __forceinline__ float d1_fmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float d1_fmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// LDS address of entry `wi` of the {mask word, rank} table (written as the instruction: the compiler forms it as shift, and, add)
__device__ __forceinline__ uint32_t d1_tab_addr(uint32_t wi, uint32_t base) {
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(wi), "s"(base));
    return r;
}
// the hit bits of a lane's codes, one per step: bit 0 of x enters at the top of hm (after CPL steps: hm >> (32 - CPL))
__device__ __forceinline__ uint32_t d1_push_bit(uint32_t hm, uint32_t x) { return __builtin_amdgcn_alignbit(x, hm, 1); }

// ---- the image form -----------------------------------------------------------------------------------------------------
// LPC lanes per candidate x CPL codes per lane (16 x 4, 16 x 8 or 32 x 8 by the index's usual number of distinct codes per passage).
// Every load of the round loop is compiler-visible and UNCONDITIONAL (s1_exact_kernel's header says why): the two code buffers'
// requests are then the only vector-memory operations in flight and the compiler's wait at the top of a round is a vmcnt(CPL / 4).
template <int LPC, int CPL>
__global__ __launch_bounds__(D1_THREADS) void s1_image_kernel(flmr_s1d_args a) {
    constexpr int R = 64 / LPC;            // candidates per round
    constexpr int LISTCAP = LPC * CPL;     // codes (= most hits) of one candidate chunk
    constexpr int NV = CPL / 4;            // 16-byte requests per lane and round
    constexpr int LPR = 4;                 // lanes per image row (16 bytes = 8 fp16 columns each)
    constexpr int HPI = LPC / LPR;         // hits folded per iteration and candidate
    constexpr int NIT = LISTCAP / HPI;     // list entries of one hit group (16 or 32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int scan_lds[17];
    __shared__ int s_nscan;
    __shared__ float s_red[2 * D1_WAVES];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / LPC, l = lane % LPC;
    const int hg = l / LPR, pr = l % LPR;
    // LDS: [scan list: nqueries i32][bits: idx_words u32][prefix: idx_words u16 (padded to 16 B)][lists: per wave R * LISTCAP + 64 u16][images]
    int* scan_list = reinterpret_cast<int*>(smem);
    uint32_t* lbits = reinterpret_cast<uint32_t*>(scan_list + ((a.nqueries + 3) & ~3));
    uint16_t* lpre = reinterpret_cast<uint16_t*>(lbits + a.idx_words);
    uint16_t* lists = lpre + ((a.idx_words + 7) & ~7);
    uint16_t* my_list = lists + (size_t)wave * (R * LISTCAP + 64);   // (+ one scratch entry per lane)
    char* img = reinterpret_cast<char*>(lists + (size_t)D1_WAVES * (R * LISTCAP + 64));

    if (a.any && a.any[0] == 0) return;   // (uniform: no query of the batch takes this pass)
    if (tid == 0) s_nscan = 0;
    __syncthreads();
    {
        int base = 0;
        for (int q0 = 0; q0 < a.nqueries; q0 += D1_THREADS) {
            const int q = q0 + tid;
            const int need = (q < a.nqueries && a.mode[q] == FLMR_S1D_IMAGE) ? 1 : 0;
            int total;
            const int pos = base + flmr_block_exclusive_scan(need, scan_lds, &total);
            if (need) scan_list[pos] = q;
            base += total;
        }
        if (tid == 0) s_nscan = base;
    }
    __syncthreads();
    const int nscan = s_nscan;
    if (nscan == 0) return;
    const int nscan8 = (nscan + D1_XCDS - 1) & ~(D1_XCDS - 1);
    const int G = a.parts;
    const int nitems = nscan8 * G;
    const int gsz = 64;

    for (int t = blockIdx.x; t < nitems; t += gridDim.x) {
        const int sidx = ((t >> 3) / G) * D1_XCDS + (t & 7), part = (t >> 3) % G;
        if (sidx >= nscan) continue;   // (block-uniform)
        const int b = scan_list[sidx];
        const int P = a.cand_count[b];
        const int32_t* const src = a.cand + (size_t)b * a.cand_stride;
        uint64_t* const keys_b = a.keys + (size_t)b * a.cand_stride;
        const int qlen = a.q_lens ? a.q_lens[b] : a.nq_cand;
        const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
        const float miss = flmr_miss_score(nqc, 0);
        const int n = a.nqual[b] < a.row_cap ? a.nqual[b] : a.row_cap;
        const float* const rows_b = a.rows + (size_t)b * a.row_cap * 32;
        d1h2 colw[4];   // 1 for the columns of this lane's 16 bytes of a row that count (< nqc), else 0
#pragma unroll
        for (int e = 0; e < 4; e++) {
            colw[e].x = (_Float16)(pr * 8 + 2 * e < nqc ? 1.0f : 0.0f);
            colw[e].y = (_Float16)(pr * 8 + 2 * e + 1 < nqc ? 1.0f : 0.0f);
        }
        __syncthreads();   // (the previous item's readers of the LDS tables are done)
        // ---- the query's mask and ranks ----
        {
            const uint32_t* gb = a.idx_bits + (size_t)b * a.idx_words;
            const uint32_t* gp = a.idx_prefix + (size_t)b * a.idx_words;
            for (int w = tid; w < a.idx_words; w += D1_THREADS) {
                lbits[w] = gb[w];
                const uint32_t p = gp[w];
                lpre[w] = (uint16_t)(p < 65535u ? p : 65535u);
            }
        }
        // ---- images: row r, columns 2j / 2j+1 -> dword 16 r + j; row n = the padding row (-inf: never wins) ----
        bool usable;
        {
            float dmax = 0.0f, amax = 0.0f;
            bool bad = false;
            if (tid == 0) s_bad = 0;
            const bool fits = n <= a.img_rows;
            if (fits) {
                uint32_t* im = reinterpret_cast<uint32_t*>(img);
                for (int e = tid; e < n * 16; e += D1_THREADS) {
                    const float2 v = *reinterpret_cast<const float2*>(rows_b + (size_t)e * 2);
                    bad |= !(fabsf(v.x) <= 60000.0f) || !(fabsf(v.y) <= 60000.0f);   // (also NaN)
                    const _Float16 hx = d1_up16(bad ? 0.0f : v.x), hy = d1_up16(bad ? 0.0f : v.y);
                    dmax = fmaxf(dmax, fmaxf((float)hx - v.x, (float)hy - v.y));
                    amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    d1h2 pk; pk.x = hx; pk.y = hy;
                    im[e] = __builtin_bit_cast(uint32_t, pk);
                }
                if (tid < 16) im[n * 16 + tid] = 0xFC00FC00u;
            }
            dmax = flmr_wave_max_f32(dmax);
            amax = flmr_wave_max_f32(amax);
            if (lane == 0) { s_red[wave] = dmax; s_red[D1_WAVES + wave] = amax; }
            __syncthreads();
            if (bad) s_bad = 1;
            __syncthreads();
            usable = fits && !s_bad;
            if (tid == 0 && part == 0) {
                float d = 0.0f, am = 0.0f;
                for (int w = 0; w < D1_WAVES; w++) { d = fmaxf(d, s_red[w]); am = fmaxf(am, s_red[D1_WAVES + w]); }
                // E: 32 rounding steps; eps: two 32-term fp32 sums of terms <= am, each within 32 * 2^-24 * (32 am) -- taken 8 x larger
                a.img_err[b] = usable ? 32.0f * d + 2.0f * 32.0f * 32.0f * am * 4.8e-7f : __builtin_huge_valf();
            }
        }
        const uint32_t pad2 = (uint32_t)n | ((uint32_t)n << 16);
        const int ngroups = (P + gsz - 1) / gsz;
        const int per = (ngroups + G - 1) / G;
        const int gbeg = part * per, gend = (gbeg + per) < ngroups ? (gbeg + per) : ngroups;
        if (!usable) {   // no images for this query: every candidate joins the band (E = inf)
            for (int i = gbeg * gsz + tid; i < gend * gsz && i < P; i += D1_THREADS) keys_b[i] = flmr_make_key(0.0f, src[i]);
            continue;
        }
        // token offsets fit 32 bits (checked by the launcher): a passage's run starts at codes + off
        auto meta = [&](int g, int& pid, uint32_t& off, int& len) {
            pid = 0; off = 0; len = 0;
            const int i = g * gsz + lane;
            if (g < gend && i < P) {
                pid = src[i];
                // (the LOW words only: a 64-bit load whose high half is unused leaves a register the compiler re-uses while the
                // load is pending -- and waits for it with a vmcnt(0) inside the round loop)
                const uint32_t* o32 = reinterpret_cast<const uint32_t*>(a.offsets) + 2 * (size_t)pid;
                off = o32[0];
                len = a.ulen ? (int)a.ulen[pid] : (int)(o32[2] - off);
            }
        };
        int pid, len; uint32_t off;
        meta(gbeg + wave, pid, off, len);
        for (int g = gbeg + wave; g < gend; g += D1_WAVES) {
            int npid, nlen; uint32_t noff;
            meta(g + D1_WAVES, npid, noff, nlen);   // next group's, while this one is processed
            const int ndoc = (P - g * gsz) < gsz ? (P - g * gsz) : gsz;
            const int nrounds = (ndoc + R - 1) / R;
            float ukeep = 0.0f;   // lane j: the score of candidate j of the group
            // Codes of round r (first chunk): CPL codes of this lane's candidate, requested TWO ROUNDS AHEAD and ALWAYS (past the
            // group's end some candidate's codes are read again): what a lane reads beyond its passage's end -- the next passage's
            // codes, the sorted copy's FLMR_CODE_PAD words of padding -- is masked by the listing
            auto request_codes = [&](d1i4u (&c_)[NV], int r, int t0, int& jlen) {
                const int j = r * R + sub;
                const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)off);
                const int jl = __builtin_amdgcn_ds_bpermute(j << 2, len);
                jlen = j < ndoc ? jl : 0;
                const d1i4u* at = reinterpret_cast<const d1i4u*>(a.codes + ((uint64_t)o + (uint32_t)(t0 + CPL * l)));
#pragma unroll
                for (int v = 0; v < NV; v++) c_[v] = D1_LOAD(at + v);
            };
            d1i4u cdA[NV], cdB[NV];
            int lenA = 0, lenB = 0;
            request_codes(cdA, 0, 0, lenA);
            request_codes(cdB, 1, 0, lenB);
            auto round = [&](d1i4u (&cd)[NV], int& jlen, int r) {
                const int my_len = jlen;
                uint32_t acc[4];     // 8 fp16 column maxima of this lane's 16 bytes of a row
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = 0xFC00FC00u;
                int c_[CPL];
#pragma unroll
                for (int v = 0; v < NV; v++) { c_[4 * v] = cd[v].x; c_[4 * v + 1] = cd[v].y; c_[4 * v + 2] = cd[v].z; c_[4 * v + 3] = cd[v].w; }
                for (int t0 = 0;; t0 += LISTCAP) {
                    // ---- list the hits of this chunk ----
                    reinterpret_cast<uint2*>(my_list)[lane] = make_uint2(pad2, pad2);   // (first 256 entries = the padding row)
                    if (R * LISTCAP > 256) reinterpret_cast<uint2*>(my_list)[64 + lane] = make_uint2(pad2, pad2);
                    int nv = my_len - (t0 + CPL * l);
                    nv = nv < 0 ? 0 : (nv > CPL ? CPL : nv);
                    uint32_t wd[CPL], wi[CPL], hm = 0u;
#pragma unroll
                    for (int e = 0; e < CPL; e++) {   // (every word read is a code of the index: see the header)
                        wi[e] = (uint32_t)c_[e] >> 5;
                        wd[e] = lbits[wi[e]];
                        hm = d1_push_bit(hm, wd[e] >> (c_[e] & 31));
                    }
                    hm = (hm >> (32 - CPL)) & ((1u << nv) - 1u);
                    const int cnt = __popc(hm);
                    int incl = cnt;
                    incl += D1_DPP(incl, D1_ROW_SHR(1));
                    incl += D1_DPP(incl, D1_ROW_SHR(2));
                    incl += D1_DPP(incl, D1_ROW_SHR(4));
                    incl += D1_DPP(incl, D1_ROW_SHR(8));
                    if (LPC == 32) incl += __builtin_amdgcn_update_dpp(0, incl, 0x142 /* row_bcast15 */, 0xA, 0xF, false);
                    int nmax;
                    if (LPC == 16) {
                        const int n0 = __builtin_amdgcn_readlane(incl, 15), n1 = __builtin_amdgcn_readlane(incl, 31);
                        const int n2 = __builtin_amdgcn_readlane(incl, 47), n3 = __builtin_amdgcn_readlane(incl, 63);
                        nmax = max(max(n0, n1), max(n2, n3));
                    } else {
                        const int n0 = __builtin_amdgcn_readlane(incl, 31), n1 = __builtin_amdgcn_readlane(incl, 63);
                        nmax = max(n0, n1);
                    }
                    {   // entry o of a candidate's list = its o-th hit; block b of the fold = entries [4 HPI b, 4 HPI (b + 1)), four per hit group
                        uint16_t* at = my_list + sub * LISTCAP + (incl - cnt);
                        uint16_t* const scratch = my_list + R * LISTCAP + lane;
#pragma unroll
                        for (int e = 0; e < CPL; e++) {
                            // (a hit's rank is below n by construction; a miss goes to the lane's own scratch entry; the bit-field
                            // extract takes its width from the low five bits of the code)
                            const int rid = (int)lpre[wi[e]] + __popc(__builtin_amdgcn_ubfe(wd[e], 0u, (uint32_t)c_[e]));
                            const uint32_t bit = __builtin_amdgcn_ubfe(hm, (uint32_t)e, 1u);
                            *(bit ? at : scratch) = (uint16_t)rid;
                            at += bit;
                        }
                    }
                    // the codes are dead: request the chunk-0 codes of round r + 2 into the same registers
                    const bool more_chunks = __ballot(my_len > t0 + LISTCAP) != 0ull;   // wave-uniform
                    if (t0 == 0) request_codes(cd, r + 2, 0, jlen);
                    // ---- fold the listed rows ----
                    if (nmax > 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        // blocks of four entries per lane (a candidate's hits are dealt to its hit groups four at a time: every group has
                        // work while there are hits), the list entries of four blocks read together, every index static, the four rows
                        // of a block requested together (an entry beyond a candidate's hits names the padding row)
                        const uint2* lp = reinterpret_cast<const uint2*>(my_list + sub * LISTCAP) + hg;   // block b: lp[b * HPI]
                        const uint32_t pb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)img + pr * 16;
#pragma unroll 1
                        for (int b0 = 0; b0 < NIT / 4; b0 += 4) {
                            if (b0 * 4 * HPI >= nmax) break;   // (wave-uniform)
                            uint2 q[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) q[u] = lp[(b0 + u) * HPI];
#pragma unroll
                            for (int bb = 0; bb < 4; bb++) {
                                if ((b0 + bb) * 4 * HPI < nmax) {   // (wave-uniform)
                                    d1u4 v[4];
                                    const uint32_t w2[2] = {q[bb].x, q[bb].y};
#pragma unroll
                                    for (int u = 0; u < 4; u++) {
                                        const uint32_t at = d1_row_addr(w2[u >> 1], u & 1, pb);
                                        v[u] = *reinterpret_cast<const __attribute__((address_space(3))) d1u4*>((uintptr_t)at);
                                    }
#pragma unroll
                                    for (int u = 0; u < 4; u++) {
                                        acc[0] = d1_pk_max(acc[0], v[u].x); acc[1] = d1_pk_max(acc[1], v[u].y);
                                        acc[2] = d1_pk_max(acc[2], v[u].z); acc[3] = d1_pk_max(acc[3], v[u].w);
                                    }
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();   // (the list is rewritten by the next chunk / round)
                    }
                    if (!more_chunks) break;
                    {   // further chunks of long passages: on demand (their wait drains the requests ahead: rare)
                        d1i4u cx[NV];
                        int dl;
                        request_codes(cx, r, t0 + LISTCAP, dl);
#pragma unroll
                        for (int v = 0; v < NV; v++) { c_[4 * v] = cx[v].x; c_[4 * v + 1] = cx[v].y; c_[4 * v + 2] = cx[v].z; c_[4 * v + 3] = cx[v].w; }
                        // (a use HERE: pending where the paths of the chunk loop meet, these loads turn the wait of the round's end -- on
                        // the common path none at all: the codes of round r + 2 stay in flight -- into a vmcnt(0))
#pragma unroll
                        for (int e = 0; e < CPL; e++) asm volatile("" : "+v"(c_[e]));
                    }
                }
                // ---- this round's candidates: combine the hit groups, sum the columns, keep the score for lane j ----
                if (HPI >= 2) {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = d1_pk_max(acc[e], (uint32_t)__builtin_amdgcn_mov_dpp((int)acc[e], D1_ROW_ROR(4), 0xF, 0xF, false));
                }
                if (HPI >= 4) {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = d1_pk_max(acc[e], (uint32_t)__builtin_amdgcn_mov_dpp((int)acc[e], D1_ROW_ROR(8), 0xF, 0xF, false));
                }
                if (HPI >= 8) {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = d1_pk_max(acc[e], (uint32_t)__shfl_xor((int)acc[e], 16, 64));
                }
                const bool nohit = (acc[0] & 0xffffu) == 0xFC00u;   // a real row is finite: -inf = only padding rows were folded
                // this lane's eight column maxima summed in fp32 by dot products with ones (zeros for the columns >= nqc): one
                // instruction per pair instead of two conversions and two additions; the products are exact, the additions fp32
                // (their order inside the instruction is the hardware's: E's roundoff term, 8 x the bound of any order, covers it)
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; e++) s = __builtin_amdgcn_fdot2(__builtin_bit_cast(d1h2, acc[e]), colw[e], s, false);
                s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
                s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
                const float sc = nohit ? miss : s;
                // the finished score sits in lane 0 of each candidate's lanes: hand it to lane j of the group
#pragma unroll
                for (int u = 0; u < R; u++) {
                    const int v = __builtin_amdgcn_readlane(__float_as_int(sc), u * LPC);
                    const int dst = __builtin_amdgcn_readfirstlane(r * R + u);
                    // (no builtin for v_writelane in this toolchain; the s_nop covers the SGPR hazards the compiler cannot see)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(ukeep) : "s"(v), "s"(dst) : "m0");
                }
            };
            for (int r = 0; r < nrounds; r += 2) {
                round(cdA, lenA, r);
                if (r + 1 < nrounds) round(cdB, lenB, r + 1);
            }
            if (lane < ndoc) keys_b[g * gsz + lane] = flmr_make_key(ukeep, pid);
            pid = npid; off = noff; len = nlen;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The EXACT form as its own kernel (round 6, second version).  The first one (s1_dense_kernel<false, .>) shares the image
// form's skeleton, and on the regime it is for -- a planted corpus at centroid_score_threshold 0.25: 8.7 k surviving centroids
// (no room for images), 60.6 k candidates of 128 distinct codes, 9 hits each -- it ran 17 ms per 1024 queries, VALU-bound: two
// candidates per round (32 lanes x 4 codes), a row id computed for every code, a 48-instruction DPP chain per round for the
// ascending-k sums, and the compiler's vmcnt(0) for every batch of row loads draining the codes requested ahead.  Here:
//   * CPL codes per lane (4 or 8: one or two 16-byte requests): 16 lanes x 8 codes hold a 128-code passage, FOUR candidates a round;
//   * hits are rare on this regime (7 % of the codes): a lane walks ITS hits (`while (hm)`), a row id is computed per hit;
//   * every load of the round loop is UNCONDITIONAL (a load under a lane test makes the compiler's count of outstanding loads
//     unknown and each of its waits a vmcnt(0)) and compiler-visible (asm requests into registers were tried: the compiler copies
//     an asm statement's output registers where it likes -- also before the hand-written wait).  Loads return in order, so a
//     request issued before the row loads would be waited for with them: the codes of round r + 2 are requested right AFTER the
//     last block of row loads of round r, and the compiler's wait for that block is a vmcnt(CPL / 4) -- the codes stay in flight
//     across the fold (checked in the generated code: profiles/r06/s1_exact_waits.txt);
//   * the per-column maxima of a candidate (8 lanes x 4 columns) go to a per-wave LDS table, and at the end of a group lane j sums
//     candidate j's 32 columns in ascending k (filter_pids.cpp:59-63): one 16-byte LDS store per round instead of the chain.
// Same arithmetic, same keys, bit for bit (tests/test_hip_regressions.py::test_stage1_dense_forms_*).
// ------------------------------------------------------------------------------------------------
#define D1X_GROUP 32   // candidates per wave and group (the LDS table holds 32 x 36 floats per wave)
#define D1X_TRS 36

template <int LPC, int CPL>
__global__ __launch_bounds__(D1_THREADS) void s1_exact_kernel(flmr_s1d_args a) {
    constexpr int R = 64 / LPC;            // candidates per round
    constexpr int LISTCAP = LPC * CPL;     // codes (= most hits) of one candidate chunk
    constexpr int NV = CPL / 4;            // 16-byte requests per lane and round
    constexpr int LPR = 8;                 // lanes per fp32 row (16 bytes each)
    constexpr int HPI = LPC / LPR;         // hits folded per iteration and candidate
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / LPC, l = lane % LPC;
    const int hg = l / LPR, pr = l % LPR;
    // LDS: [mask + ranks: idx_words x {u32, u32}][scan list][lists: per wave R * LISTCAP u16][tr: per wave a.group x D1X_TRS f32][stash: per wave CPL x 64 i32]
    // (the mask first: its address is then a compile-time offset of the probes; a word and the rank of its first bit side by side:
    // one 8-byte read per hit in the walk)
    uint2* ltab = reinterpret_cast<uint2*>(smem);
    const uint32_t ltab_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    int* scan_list = reinterpret_cast<int*>(ltab + a.idx_words);
    uint16_t* lists = reinterpret_cast<uint16_t*>(scan_list + ((a.nqueries + 3) & ~3));
    uint16_t* my_list = lists + (size_t)wave * (R * LISTCAP);
    float* tr = reinterpret_cast<float*>(lists + (size_t)D1_WAVES * (R * LISTCAP)) + (size_t)wave * a.group * D1X_TRS;
    // a lane's codes of the round, [code][lane]: the hit walk reads code e of its lane with one LDS load (a register array
    // indexed by a lane's own e is a chain of CPL compare-and-selects)
    int* stash = reinterpret_cast<int*>(reinterpret_cast<float*>(lists + (size_t)D1_WAVES * (R * LISTCAP)) + (size_t)D1_WAVES * a.group * D1X_TRS) +
                 (size_t)wave * (CPL * 64) + lane;
    // (no static __shared__ in this kernel: the dynamic block -- the mask -- then starts at LDS address 0 and a probe's address is
    // the shifted code itself; the block scan's scratch and two scalars sit behind the stash)
    int* const scan_lds = reinterpret_cast<int*>(reinterpret_cast<float*>(lists + (size_t)D1_WAVES * (R * LISTCAP)) + (size_t)D1_WAVES * a.group * D1X_TRS) +
                          (size_t)D1_WAVES * (CPL * 64);
    int& s_nscan = scan_lds[17];
    int& s_item = scan_lds[18];

    if (a.any && a.any[1] == 0) return;   // (uniform: no query of the batch takes this pass)
    if (tid == 0) s_nscan = 0;
    // (list entries past a candidate's hits are read as row ids before they are known to be unused: never beyond row_cap)
    for (int i = tid; i < D1_WAVES * R * LISTCAP / 2; i += D1_THREADS) reinterpret_cast<uint32_t*>(lists)[i] = 0u;
    __syncthreads();
    {
        int base = 0;
        for (int q0 = 0; q0 < a.nqueries; q0 += D1_THREADS) {
            const int q = q0 + tid;
            const int need = (q < a.nqueries && a.mode[q] != FLMR_S1D_SKIP) ? 1 : 0;
            int total;
            const int pos = base + flmr_block_exclusive_scan(need, scan_lds, &total);
            if (need) scan_list[pos] = q;
            base += total;
        }
        if (tid == 0) s_nscan = base;
    }
    __syncthreads();
    const int nscan = s_nscan;
    if (nscan == 0) return;
    const int nscan8 = (nscan + D1_XCDS - 1) & ~(D1_XCDS - 1);
    const int G = a.parts;
    const int nitems = nscan8 * G;

    // Items are taken from a counter per XCD (workgroup L runs on XCD L % 8 as far as the dispatcher keeps its round robin: affinity
    // for speed only): a band (ndocs + a few per cent) fills a few of a query's G parts, a whole list all of them, and a static
    // deal -- workgroup L sees the same part number in every one of its items when 256 / 8 is a multiple of G -- left 10 of 16
    // workgroups without work on a batch of bands.  Consecutive items of a counter are the parts of one query: they run side by
    // side on the XCD whose L2 holds the query's rows.
    const int xcd = blockIdx.x & (D1_XCDS - 1);
    const int per_xcd = (nscan8 / D1_XCDS) * G;
    for (int t = blockIdx.x;; t += gridDim.x) {
        int it;
        if (a.any) {
            __syncthreads();   // (the previous item's readers of s_item and of the LDS tables are done)
            if (tid == 0) s_item = atomicAdd(const_cast<int32_t*>(a.any) + 2 + xcd, 1);
            __syncthreads();
            it = s_item;
            if (it >= per_xcd) break;
        } else {   // (a caller without counters -- the stand-alone probe: the static deal)
            if (t >= nitems) break;
            it = t >> 3;
        }
        const int sidx = (it / G) * D1_XCDS + (a.any ? xcd : (t & 7)), part = it % G;
        if (sidx >= nscan) continue;   // (block-uniform)
        const int b = __builtin_amdgcn_readfirstlane(scan_list[sidx]);
        const bool from_band = a.mode[b] == FLMR_S1D_IMAGE;
        const int P = from_band ? a.band_count[b] : a.cand_count[b];
        // a band is short (ndocs + a few per cent): groups of 16 so that every wave of the item has some; whole lists: 32
        const int gsz = from_band ? 16 : a.group;
        const int32_t* const src = (from_band ? a.band : a.cand) + (size_t)b * a.cand_stride;
        uint64_t* const keys_b = a.keys + (size_t)b * a.cand_stride;
        const int qlen = a.q_lens ? a.q_lens[b] : a.nq_cand;
        const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
        const float miss = flmr_miss_score(nqc, 0);
        const int n = a.nqual[b] < a.row_cap ? a.nqual[b] : a.row_cap;
        const float* const rows_b = a.rows + (size_t)b * a.row_cap * 32;
        const int ngroups = (P + gsz - 1) / gsz;
        // a part takes at least one group per wave: a band (ndocs + a few per cent) is a few parts' work, a whole list all G parts'
        const int per = max((ngroups + G - 1) / G, D1_WAVES);
        if (part * per >= ngroups) continue;   // (block-uniform: this part of a short list is empty -- no table load)
        __syncthreads();   // (the previous item's readers of the LDS tables are done)
        {
            const uint32_t* gb = a.idx_bits + (size_t)b * a.idx_words;
            const uint32_t* gp = a.idx_prefix + (size_t)b * a.idx_words;
            for (int w = tid; w < a.idx_words; w += D1_THREADS) {
                ltab[w] = make_uint2(gb[w], gp[w]);
            }
        }
        __syncthreads();
        const int gbeg = part * per, gend = (gbeg + per) < ngroups ? (gbeg + per) : ngroups;
        auto meta = [&](int g, int& pid, uint32_t& off, int& len) {
            pid = 0; off = 0; len = 0;
            const int i = g * gsz + lane;
            if (g < gend && lane < gsz && i < P) {
                pid = src[i];
                const uint32_t* o32 = reinterpret_cast<const uint32_t*>(a.offsets) + 2 * (size_t)pid;   // (low words: see s1_dense_kernel)
                off = o32[0];
                len = a.ulen ? (int)a.ulen[pid] : (int)(o32[2] - off);
            }
        };
        int pid, len; uint32_t off;
        meta(gbeg + wave, pid, off, len);
        for (int g = gbeg + wave; g < gend; g += D1_WAVES) {
            int npid, nlen; uint32_t noff;
            meta(g + D1_WAVES, npid, noff, nlen);
            const int ndoc = (P - g * gsz) < gsz ? (P - g * gsz) : gsz;
            const int nrounds = (ndoc + R - 1) / R;
            auto request_codes = [&](d1i4u (&c_)[NV], int r, int& jlen) {   // always issued (see s1_dense_kernel): NV requests
                const int j = r * R + sub;
                const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)off);
                const int jl = __builtin_amdgcn_ds_bpermute(j << 2, len);
                jlen = j < ndoc ? jl : 0;
                const int32_t* at = a.codes + ((uint64_t)o + (uint32_t)(CPL * l));
#pragma unroll
                for (int v = 0; v < NV; v++) c_[v] = D1_LOAD(reinterpret_cast<const d1i4u*>(at) + v);
            };
            d1i4u cdA[NV], cdB[NV];
            int lenA = 0, lenB = 0;
            request_codes(cdA, 0, lenA);
            request_codes(cdB, 1, lenB);
            const uint32_t prb = (uint32_t)pr * 16u;
            const char* const rbytes = reinterpret_cast<const char*>(rows_b);
            uint16_t* const lst = my_list + sub * LISTCAP;
            const d1u4* const lp = reinterpret_cast<const d1u4*>(lst) + hg;   // block blk of this lane's hit group: lp[blk * HPI]
            auto round = [&](d1i4u (&cd)[NV], int& jlen, int r) {
                const int my_len = jlen;
                d1f4 facc;
                facc.x = facc.y = facc.z = facc.w = -9999.0f;   // filter_pids.cpp:30-33
                int nh_total = 0;
                int c_[CPL];
#pragma unroll
                for (int v = 0; v < NV; v++) { c_[4 * v] = cd[v].x; c_[4 * v + 1] = cd[v].y; c_[4 * v + 2] = cd[v].z; c_[4 * v + 3] = cd[v].w; }
                for (int t0 = 0;; t0 += LISTCAP) {
                    // ---- the hit mask of this lane's CPL codes (what a lane reads past its passage -- the next passage's codes, the
                    // copy's padding -- is a code of the index: a word of the mask; its bit is dropped with `nv`) ----
                    int nv = my_len - (t0 + CPL * l);
                    nv = nv < 0 ? 0 : (nv > CPL ? CPL : nv);
                    uint32_t hm = 0u;
                    {
                        uint32_t wd[CPL];   // (all CPL words requested before the first is used: one LDS round trip, not CPL / 2)
#pragma unroll
#ifdef D1X_ABL_NOMASK   // development switch: no mask reads (a hash of the code instead), no stash
                        for (int e = 0; e < CPL; e++) wd[e] = ((uint32_t)c_[e] * 2654435761u) >> 28 == 0u ? ~0u : 0u;
#elif defined(D1X_ABL_LINMASK)   // development switch: the probes read conflict-free addresses (wrong words: timing only)
                        for (int e = 0; e < CPL; e++) wd[e] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)d1_tab_addr((uint32_t)(lane + 64 * e) + ((uint32_t)c_[e] >> 31), ltab_lds));
#pragma unroll
                        for (int e = 0; e < CPL; e++) stash[e * 64] = c_[e];
#else
                        for (int e = 0; e < CPL; e++) wd[e] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)d1_tab_addr((uint32_t)c_[e] >> 5, ltab_lds));
#ifndef D1X_ABL_NOSTASH   // development switch: no stash writes (the walk reads stale codes: timing only)
#pragma unroll
                        for (int e = 0; e < CPL; e++) stash[e * 64] = c_[e];
#endif
#endif
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int e = 0; e < CPL; e++) hm = d1_push_bit(hm, wd[e] >> (c_[e] & 31));
                    }
                    hm = (hm >> (32 - CPL)) & ((1u << nv) - 1u);
                    const int cnt = __popc(hm);
                    int incl = cnt;
                    incl += D1_DPP(incl, D1_ROW_SHR(1));
                    incl += D1_DPP(incl, D1_ROW_SHR(2));
                    incl += D1_DPP(incl, D1_ROW_SHR(4));
                    incl += D1_DPP(incl, D1_ROW_SHR(8));
                    if (LPC == 32) incl += __builtin_amdgcn_update_dpp(0, incl, 0x142 /* row_bcast15 */, 0xA, 0xF, false);
                    // (the candidate's count: the scan's value in the last lane of its lanes)
                    const int nh_own = __builtin_amdgcn_ds_bpermute((lane | (LPC - 1)) << 2, incl);
                    int nmax;
                    if (LPC == 16) {
                        const int n0 = __builtin_amdgcn_readlane(incl, 15), n1 = __builtin_amdgcn_readlane(incl, 31);
                        const int n2 = __builtin_amdgcn_readlane(incl, 47), n3 = __builtin_amdgcn_readlane(incl, 63);
                        nmax = max(max(n0, n1), max(n2, n3));
                    } else {
                        const int n0 = __builtin_amdgcn_readlane(incl, 31), n1 = __builtin_amdgcn_readlane(incl, 63);
                        nmax = max(n0, n1);
                    }
                    nh_total += nh_own;
                    // ---- a lane walks its hits (7 % of the codes on the regime this form is for: listing every code costs more
                    // instructions than the walk's ~2.7 iterations): entry o of a candidate's list = the row id (rank of the centroid
                    // among the survivors) of its o-th hit ----
                    {
                        uint16_t* at = lst + (incl - cnt);
                        uint32_t left = hm;
#ifdef D1X_ABL_NOWALK   // development switch: one list entry per lane with a hit, no rank
                        if (left) { *at = (uint16_t)(c_[0] & 1023); left = 0u; }
#endif
                        while (left) {
                            const int e = __ffs(left) - 1;
                            left &= left - 1u;
                            const int c = stash[e * 64];
                            const uint32_t wi = (uint32_t)c >> 5;
                            const d1u2 wp = *reinterpret_cast<const __attribute__((address_space(3))) d1u2*>((uintptr_t)d1_tab_addr(wi, ltab_lds));
                            int rid = (int)wp.y + __popc(wp.x & ((1u << (c & 31)) - 1u));
                            rid = rid < n ? rid : n - 1;
                            *at++ = (uint16_t)rid;
                        }
                    }
                    const bool more_chunks = __ballot(my_len > t0 + LISTCAP) != 0ull;   // wave-uniform
                    // ---- fold the listed rows: blocks of EIGHT entries (16 bytes of the list) per lane, LPC slots per candidate, hit
                    // group hg takes entries [blk * LPC + 8 hg, + 8); the rows of a block are requested together (a block is one L2
                    // round trip whatever it holds).  No entry is tested: the slots of a candidate's blocks past its hits are filled
                    // with its FIRST hit (a maximum is idempotent), and a candidate without a hit in the chunk drops what its lanes
                    // folded (`has`).  The codes of round r + 2 are requested behind the LAST block's row loads (in-order returns: a
                    // request issued before them would be waited for with them), at ONE place in the program ----
                    if (nmax > 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int nblocks = (nmax + LPC - 1) / LPC;   // (wave-uniform, >= 1)
                        {
                            const uint16_t first = lst[0];
                            for (int slot = nh_own + l; slot < nblocks * LPC; slot += LPC) lst[slot] = first;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        d1f4 fch;
                        fch.x = fch.y = fch.z = fch.w = -9999.0f;
                        auto rows_of = [&](int blk, d1f4 (&v)[8]) {
                            const d1u4 e4 = lp[blk * HPI];
                            const uint32_t w[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
#ifdef D1X_ABL_NOROWS   // development switch (profiles/r06/s1_exact_ablation.txt): no row loads
                            for (int u = 0; u < 8; u++) { v[u].x = v[u].y = v[u].z = v[u].w = __int_as_float((int)d1x_row_off(w[u >> 1], u & 1, prb)); }
#elif defined(D1X_ABL_HALFROWS)   // development switch: half of the row loads
                            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const d1f4*>(rbytes + d1x_row_off(w[u >> 1], u & 1, prb));
                            for (int u = 4; u < 8; u++) { v[u].x = v[u].y = v[u].z = v[u].w = __int_as_float((int)d1x_row_off(w[u >> 1], u & 1, prb)); }
#elif defined(D1X_ABL_ROWS64)   // development switch: what a pass over 64-byte (fp16) rows would move -- four loads a lane, rows 64 bytes apart
                            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const d1f4*>(rbytes + (d1x_row_off(w[u >> 1], u & 1, 0u) >> 1) + (prb & 48u));
                            for (int u = 4; u < 8; u++) { v[u].x = v[u].y = v[u].z = v[u].w = __int_as_float((int)d1x_row_off(w[u >> 1], u & 1, prb)); }
#elif defined(D1X_ABL_SAMEROWS)   // development switch: entries 4 .. 7 read ONE line (row 0) in every lane
                            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const d1f4*>(rbytes + d1x_row_off(w[u >> 1], u & 1, prb));
                            for (int u = 4; u < 8; u++) v[u] = *reinterpret_cast<const d1f4*>(rbytes + (d1x_row_off(w[u >> 1], u & 1, prb) & 127u));
#else
                            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const d1f4*>(rbytes + d1x_row_off(w[u >> 1], u & 1, prb));
#endif
                        };
                        auto take = [&](const d1f4 (&v)[8]) {
#pragma unroll
                            for (int u = 0; u < 8; u += 2) {
                                fch.x = d1_fmax3(fch.x, v[u].x, v[u + 1].x); fch.y = d1_fmax3(fch.y, v[u].y, v[u + 1].y);
                                fch.z = d1_fmax3(fch.z, v[u].z, v[u + 1].z); fch.w = d1_fmax3(fch.w, v[u].w, v[u + 1].w);
                            }
                        };
#pragma unroll 1
                        for (int blk = 0; blk + 1 < nblocks; blk++) {
                            d1f4 v[8];
                            rows_of(blk, v);
                            take(v);
                        }
                        {
                            d1f4 v[8];
                            rows_of(nblocks - 1, v);
                            asm volatile("" ::: "memory");   // (the order of the requests is the point)
                            request_codes(cd, r + 2, jlen);   // (in a later chunk of long passages: the same request again -- unconditional)
                            take(v);
                        }
                        {   // (a select, not a branch: under a lane test the compiler sinks the row loads into it)
                            const bool has = nh_own > 0;
                            facc.x = d1_fmax(facc.x, has ? fch.x : -9999.0f); facc.y = d1_fmax(facc.y, has ? fch.y : -9999.0f);
                            facc.z = d1_fmax(facc.z, has ? fch.z : -9999.0f); facc.w = d1_fmax(facc.w, has ? fch.w : -9999.0f);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();   // (the list is rewritten by the next chunk / round)
                    } else {
                        request_codes(cd, r + 2, jlen);   // (a round / chunk without a hit)
                    }
                    if (!more_chunks) break;
                    {   // further chunks of long passages: on demand, compiler-visible (its wait drains the requests ahead: rare)
                        const int j = r * R + sub;
                        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)off);
                        const d1i4u* at = reinterpret_cast<const d1i4u*>(a.codes + ((uint64_t)o + (uint32_t)(t0 + LISTCAP + CPL * l)));
#pragma unroll
                        for (int v = 0; v < NV; v++) { const d1i4u x = D1_LOAD(at + v); c_[4 * v] = x.x; c_[4 * v + 1] = x.y; c_[4 * v + 2] = x.z; c_[4 * v + 3] = x.w; }
                        // (a use HERE: pending at the loop's header, these loads would turn the header's wait into a vmcnt(0))
#pragma unroll
                        for (int e = 0; e < CPL; e++) asm volatile("" : "+v"(c_[e]));
                    }
                }
                // ---- this round's candidates: combine the hit groups; the 8 lanes of a candidate park their 4 columns in the table ----
                if (HPI >= 2) {
                    facc.x = d1_fmax(facc.x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(facc.x), D1_ROW_ROR(8), 0xF, 0xF, false)));
                    facc.y = d1_fmax(facc.y, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(facc.y), D1_ROW_ROR(8), 0xF, 0xF, false)));
                    facc.z = d1_fmax(facc.z, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(facc.z), D1_ROW_ROR(8), 0xF, 0xF, false)));
                    facc.w = d1_fmax(facc.w, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(facc.w), D1_ROW_ROR(8), 0xF, 0xF, false)));
                }
                if (HPI >= 4) {
                    facc.x = d1_fmax(facc.x, __shfl_xor(facc.x, 16, 64)); facc.y = d1_fmax(facc.y, __shfl_xor(facc.y, 16, 64));
                    facc.z = d1_fmax(facc.z, __shfl_xor(facc.z, 16, 64)); facc.w = d1_fmax(facc.w, __shfl_xor(facc.w, 16, 64));
                }
                const int j = r * R + sub;
                if (l < 8 && j < gsz) {
                    if (nh_total == 0) facc.x = facc.y = facc.z = facc.w = __int_as_float(0x7fc00000);   // (NaN marks "no surviving centroid")
                    *reinterpret_cast<d1f4*>(tr + j * D1X_TRS + 4 * l) = facc;
                }
            };
            for (int r = 0; r < nrounds; r += 2) {
                round(cdA, lenA, r);
                if (r + 1 < nrounds) round(cdB, lenB, r + 1);
            }
            // (the requests past the group's end must have landed before their registers mean anything else to the compiler)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < ndoc) {
                // ascending-k sum of the candidate's column maxima (filter_pids.cpp:59-63); columns >= nqc add +0.0f (bits unchanged)
                const d1f4* row = reinterpret_cast<const d1f4*>(tr + lane * D1X_TRS);
                d1f4 x[8];
#pragma unroll
                for (int w = 0; w < 8; w++) x[w] = row[w];
                float sc = 0.0f;
                const bool nohit = x[0].x != x[0].x;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    sc += 4 * w < nqc ? x[w].x : 0.0f; sc += 4 * w + 1 < nqc ? x[w].y : 0.0f;
                    sc += 4 * w + 2 < nqc ? x[w].z : 0.0f; sc += 4 * w + 3 < nqc ? x[w].w : 0.0f;
                }
                keys_b[g * gsz + lane] = flmr_make_key(nohit ? miss : sc, pid);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the table is rewritten by the next group)
            pid = npid; off = noff; len = nlen;
        }
    }
}

// (lanes per candidate, codes per lane) by the index's usual number of distinct codes per passage
static void d1_shape(double mean_codes, int* lpc, int* cpl) {
    *lpc = mean_codes > 144.0 ? 32 : 16;
    *cpl = mean_codes > 72.0 ? 8 : 4;
}
static size_t d1x_lds(int nqueries, int idx_words, int lpc, int cpl, int group) {
    return (size_t)((nqueries + 3) & ~3) * 4 + (size_t)idx_words * 8 +
           (size_t)D1_WAVES * (64 / lpc) * (lpc * cpl) * 2 + (size_t)D1_WAVES * group * D1X_TRS * 4 + (size_t)D1_WAVES * cpl * 64 * 4 + 32 * 4;
}

// the exact pass: (lanes per candidate, codes per lane) by the index's usual number of distinct codes per passage
int flmr_launch_s1_exact(const flmr_s1d_args& a_in, double mean_codes, hipStream_t st) {
    flmr_s1d_args a = a_in;
    int lpc, cpl;
    d1_shape(mean_codes, &lpc, &cpl);
    // candidates per wave and group: 32 where the table fits (and the caller does not ask for 16), else 16
    a.group = (a.group == 16 || d1x_lds(a.nqueries, a.idx_words, lpc, cpl, D1X_GROUP) > (size_t)160 * 1024 - 1024) ? 16 : D1X_GROUP;
    const size_t lds = d1x_lds(a.nqueries, a.idx_words, lpc, cpl, a.group);
    if (lds > (size_t)160 * 1024 - 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dense stage 1: K = %d does not fit the LDS form", a.idx_words * 32);
    if (a.codes_len > 0xffffffffLL) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dense stage 1: token offsets beyond 32 bits");
    if (a.parts < 1) a.parts = 8;
    int64_t grid = 256;
    const int64_t max_items = (int64_t)((a.nqueries + D1_XCDS - 1) & ~(D1_XCDS - 1)) * a.parts;
    if (grid > max_items) grid = max_items;
    dim3 g((unsigned)grid), block(D1_THREADS);
#define D1X_LAUNCH(L, C) do { \
        if (lds > 48 * 1024) FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s1_exact_kernel<L, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((s1_exact_kernel<L, C>), g, block, lds, st, a); } while (0)
    if (lpc == 16 && cpl == 4) D1X_LAUNCH(16, 4);
    else if (lpc == 16) D1X_LAUNCH(16, 8);
    else D1X_LAUNCH(32, 8);
#undef D1X_LAUNCH
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// LDS the image launch needs besides the images; rows of images that fit beside it
static size_t d1_fixed_lds(int nqueries, int idx_words, int lpc, int cpl) {
    return (size_t)((nqueries + 3) & ~3) * 4 + (size_t)idx_words * 4 + (size_t)((idx_words + 7) & ~7) * 2 +
           (size_t)D1_WAVES * ((64 / lpc) * (lpc * cpl) + 64) * 2;
}
int flmr_s1_dense_image_rows(int nqueries, int idx_words, double mean_codes) {
    int lpc, cpl;
    d1_shape(mean_codes, &lpc, &cpl);
    const size_t budget = (size_t)160 * 1024 - 1024;   // static __shared__ of the kernel and alignment
    const size_t fixed = d1_fixed_lds(nqueries, idx_words, lpc, cpl);
    if (fixed + 64 * 65 > budget || d1x_lds(nqueries, idx_words, lpc, cpl, 16) > budget) return 0;
    const size_t rows = (budget - fixed) / 64 - 1;   // (+ the padding row)
    return (int)(rows > 65000 ? 65000 : rows);
}

int flmr_launch_s1_image(const flmr_s1d_args& a_in, double mean_codes, hipStream_t st) {
    flmr_s1d_args a = a_in;
    int lpc, cpl;
    d1_shape(mean_codes, &lpc, &cpl);
    a.img_rows = flmr_s1_dense_image_rows(a.nqueries, a.idx_words, mean_codes);
    if (a.img_rows < 1) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dense stage 1: the centroid mask leaves no room for score-row images in LDS");
    const size_t lds = d1_fixed_lds(a.nqueries, a.idx_words, lpc, cpl) + (size_t)(a.img_rows + 1) * 64;
    if (a.codes_len > 0xffffffffLL) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "dense stage 1: token offsets beyond 32 bits");
    if (a.parts < 1) a.parts = 4;
    int64_t grid = 256;   // one persistent workgroup per CU; a multiple of 8 so that workgroup L and its items share L % 8
    const int64_t max_items = (int64_t)((a.nqueries + D1_XCDS - 1) & ~(D1_XCDS - 1)) * a.parts;
    if (grid > max_items) grid = max_items;
    dim3 g((unsigned)grid), block(D1_THREADS);
#define D1I_LAUNCH(L, C) do { \
        if (lds > 48 * 1024) FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s1_image_kernel<L, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((s1_image_kernel<L, C>), g, block, lds, st, a); } while (0)
    if (lpc == 16 && cpl == 4) D1I_LAUNCH(16, 4);
    else if (lpc == 16) D1I_LAUNCH(16, 8);
    else D1I_LAUNCH(32, 8);
#undef D1I_LAUNCH
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// one block: mode / scan_skip per query, and any[0] = some query takes the image form, any[1] = some query takes the exact pass (a band
// or a whole list): the dense kernels read their word first and leave at once when it is 0 (the usual case on the tuned path)
__global__ __launch_bounds__(1024) void s1_dense_modes_kernel(const int32_t* skip, const int32_t* nqual, const int32_t* row_ovf, int32_t nqueries,
                                                              int32_t img_rows, int32_t exact_too, int32_t* mode, int32_t* scan_skip, int32_t* any) {
    int any_img = 0, any_exact = 0;
    for (int q = threadIdx.x; q < nqueries; q += 1024) {
        int m = FLMR_S1D_SKIP, sk = 1;
        if (!(skip && skip[q]) && !(row_ovf && row_ovf[q])) {
            if (nqual[q] <= img_rows) m = FLMR_S1D_IMAGE;
            else if (exact_too) m = FLMR_S1D_EXACT;
            else sk = 0;   // the scan takes the query
        }
        mode[q] = m;
        scan_skip[q] = sk;
        any_img |= m == FLMR_S1D_IMAGE;
        any_exact |= m != FLMR_S1D_SKIP;
    }
    any_img = __syncthreads_or(any_img);
    any_exact = __syncthreads_or(any_exact);
    if (threadIdx.x == 0) { any[0] = any_img; any[1] = any_exact; }
    if (threadIdx.x >= 2 && threadIdx.x < 16) any[threadIdx.x] = 0;   // (the exact pass's item counters)
}

int flmr_launch_s1_dense_modes(const int32_t* skip, const int32_t* nqual, const int32_t* row_ovf, int32_t nqueries, int32_t img_rows,
                               int32_t exact_too, int32_t* mode, int32_t* scan_skip, int32_t* any, hipStream_t st) {
    hipLaunchKernelGGL(s1_dense_modes_kernel, dim3(1), dim3(1024), 0, st, skip, nqual, row_ovf, nqueries, img_rows, exact_too, mode,
                       scan_skip, any);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
