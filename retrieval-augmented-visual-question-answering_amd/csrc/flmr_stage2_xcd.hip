// Stage 2 (filter_pids.cpp:77-110: the all-centroid MaxSim of the top-ndocs survivors), with the centroid table cut into ONE
// L2-RESIDENT SLICE PER XCD.
//
// Why.  The gather form (filter_stage2_lds_kernel) fetches one 256-byte fp16 centroid row per survivor token: 134 M rows =
// 34 GB per 1024 queries at BASELINE's shape, from a 33.5 MB table that no L2 (4 MB per XCD) holds -- every XCD touches every
// row, the rows come from the Infinity Cache, and the chip delivers 9.3-9.6 TB/s for that access (profiles/microbench/
// s2_design_probe: "table 32 MB").  The same probe with each workgroup confined to slice (blockIdx.x % 8) of the table --
// workgroups are dealt to the eight XCDs round-robin, so every XCD's L2 sees 4 MB of it -- delivers 23-30 TB/s ("table 32 MB
// in 8 XCD slices").  So the work is cut by CODE RANGE: block L handles only the tokens whose centroid id lies in slice L % 8.
//
// How.
//   * `codes_sorted` (per-passage ascending copy of the codes, built at flmr_index_open for the walk kernel) makes the slice-s
//     tokens of a passage a contiguous run; `doc_splits[p][s]` (8 x u16 per passage, built here at open) says where it starts.
//   * A wave takes 64 survivors of one query and ONE slice.  Each passage's run is padded to a multiple of 8 tokens (the pad
//     repeats its last token -- a max is not changed by a repeat) and the runs are concatenated into a flat stream of
//     8-token octets; a 32-row MFMA tile is four octets.  The 32x32 accumulator puts rows 8g..8g+7 in registers 4g..4g+3 of the
//     two half-waves, so the maximum over an octet is a STATIC register group: no row masks, no per-row selects.
//   * Rows go global -> LDS by DMA two tiles ahead; their codes (an HBM stream, ~2 us away) go global -> LDS six tiles ahead
//     into a small ring, so that no step waits for a code; both from inline assembly with hand-counted vmcnt waits (see maxsim_f16_dma_kernel in flmr_maxsim.hip for the rules and why the compiler's own wait
//     insertion cannot be used with a deep DMA pipeline).
//   * Per (passage, slice) the wave writes the 32 column maxima ("partial", -9999 start as filter_pids.cpp:30-33) to
//     part[query][slot][slice][32]; s2_combine_kernel takes the maximum over the slices that hold tokens and sums the columns
//     k-ascending exactly like the gather kernel.  max is exact and every token's 32 scores come from the same MFMA sequence as
//     in the gather kernel (one output row depends on its own A row only), so the keys are bit-identical.
//
// Cost: the partial maxima are 1 KB per (query, survivor) written and read once (1 GB per 1024 x 1024), against 34 GB of row
// gathers served from L2 instead of the fabric.
#include "flmr_common.h"
#include "flmr_device.h"
#include <type_traits>

#define X2_XCDS 8        // L2 domains a launch's workgroups are dealt to round-robin (SPX mode: checked at index open)
#define X2_MAX_SLICES 32 // slices of the centroid table (a multiple of 8, chosen at index open so that a slice fits an L2)
#define X2_WAVES 4
#define X2_DOCS 64  // survivors per wave (one per lane)
#define X2_AHEAD 6  // the codes of a tile are requested six tiles before its rows are consumed (four before they are requested)
#define X2_RING 8   // code ring slots (256 B each) per wave
#define X2_OCT 256  // octets (of 8 tokens) per batch of passages: the per-wave octet table
#define X2_WAVE_LDS (16384 + X2_RING * 256 + X2_OCT * 6)  // row buffers + code ring + octet table

typedef _Float16 x2h8 __attribute__((ext_vector_type(8)));
typedef float x2f16 __attribute__((ext_vector_type(16)));
typedef uint32_t x2u4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// doc_splits[p * nsl + s] = number of codes of passage p below s * slice_rows (position in the sorted copy), s = 0..nsl-1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void doc_splits_kernel(const int32_t* __restrict__ codes_sorted, const int64_t* __restrict__ offsets,
                                                         const uint16_t* __restrict__ ulen,
                                                         int64_t npass, int slice_rows, int nsl, uint16_t* __restrict__ splits) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = t / nsl;
    const int s = (int)(t % nsl);
    if (p >= npass) return;
    const int64_t off = offsets[p];
    const int len = ulen ? (int)ulen[p] : (int)(offsets[p + 1] - off);   // (the distinct prefix of the run, when the index keeps its length)
    const int bound = s * slice_rows;
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (codes_sorted[off + mid] < bound) lo = mid + 1; else hi = mid;
    }
    splits[p * nsl + s] = (uint16_t)lo;
}

// ------------------------------------------------------------------------------------------------
// Which L2 does workgroup L of a 1-D grid run on?  The sliced kernel confines the workgroups with L % 8 == x to table
// slices x, x + 8, ...: that only helps if those workgroups share an L2, i.e. if the dispatcher deals consecutive
// workgroups to the eight XCDs round-robin (SPX partition mode).  Checked ONCE at index open instead of assumed: every
// workgroup of a probe grid reads its XCC_ID hardware register; the sliced form is enabled only if the device shows
// all 8 XCDs and L % 8 determines the XCD.
// ------------------------------------------------------------------------------------------------
__global__ void xcc_probe_kernel(int32_t* __restrict__ out) {
    if (threadIdx.x == 0) {
        uint32_t v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int32_t)(v & 0xf);
    }
}

static int x2_xcd_mapping_ok(bool* ok) {
    *ok = false;
    const int n = 4096;
    int32_t* dev = nullptr;
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(&dev), n * sizeof(int32_t)));
    int32_t host[4096];
    bool good = true;
    for (int rep = 0; rep < 2 && good; rep++) {   // twice: the mapping must not depend on where the previous launch stopped
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(n), dim3(64), 0, 0, dev);
        if (hipGetLastError() != hipSuccess || hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) { good = false; break; }
        int of_residue[X2_XCDS];
        bool seen[16] = {};
        for (int r = 0; r < X2_XCDS; r++) { of_residue[r] = host[r]; seen[host[r] & 15] = true; }
        int distinct = 0;
        for (int x = 0; x < 16; x++) distinct += seen[x] ? 1 : 0;
        if (distinct != X2_XCDS) good = false;
        for (int L = 0; L < n && good; L++) good = host[L] == of_residue[L % X2_XCDS];
    }
    (void)hipFree(dev);
    (void)hipGetLastError();
    *ok = good;
    return FLMR_OK;
}

// slices: the smallest multiple of 8 whose slices fit an XCD's L2 next to the streams passing through it (4 MB L2; 4.2 MB
// slices -- K = 131072 in 8 -- still measured at the resident rate, 8 MB slices at a third of it: DESIGN.md section 4)
static int x2_slices_for(int64_t K) {
    const size_t table = (size_t)K * FLMR_DIM * sizeof(_Float16);
    const size_t per_slice = ((size_t)4 << 20) + ((size_t)1 << 18);
    int n = X2_XCDS;
    while (n < X2_MAX_SLICES && table > per_slice * (size_t)n) n += X2_XCDS;
#ifdef X2_FORCE_SLICES   // development probe (profiles/build_variant.py): e.g. 16 half-L2 slices at K = 131072
    if (n < X2_FORCE_SLICES) n = X2_FORCE_SLICES;
#endif
    return n;
}

int flmr_build_doc_splits(flmr_index* ix) {
    ix->doc_splits = nullptr;
    ix->nslices = x2_slices_for(ix->K);
    ix->slice_rows = (int32_t)flmr_ceil_div(ix->K, ix->nslices);
    ix->xcd_round_robin = 0;
    if (!ix->codes_sorted || ix->max_doclen > 65535) return FLMR_OK;
    bool ok = false;
    const int rc = x2_xcd_mapping_ok(&ok);
    if (rc) return rc;
    ix->xcd_round_robin = ok ? 1 : 0;
    if (hipMalloc(reinterpret_cast<void**>(&ix->doc_splits), (size_t)ix->num_passages * ix->nslices * sizeof(uint16_t)) != hipSuccess) {
        (void)hipGetLastError();   // the sliced form is then unavailable: stage 2 takes the gather form
        ix->doc_splits = nullptr;
        return FLMR_OK;
    }
    const int64_t threads = ix->num_passages * ix->nslices;
    hipLaunchKernelGGL(doc_splits_kernel, dim3((unsigned)flmr_ceil_div(threads, 256)), dim3(256), 0, 0, ix->codes_sorted,
                       ix->doc_offsets, ix->doc_ulen, ix->num_passages, ix->slice_rows, ix->nslices, ix->doc_splits);
    FLMR_LAUNCH_CHECK();
    FLMR_HIP(hipDeviceSynchronize());
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void x2_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ float x2_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float x2_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// grid = nsl * nqueries * G (1-D; block L: XCD x = L & 7, query (L >> 3) % nqueries, survivor group ((L >> 3) / nqueries) % G,
// pass hf = (L >> 3) / (nqueries * G), slice 8 * hf + x: with more than 8 slices the grid works through them eight at a time,
// so that an L2 holds one slice at any moment), block = 256 (4 waves x 64 survivors);
// dynamic LDS = 4 x (16 KB row buffers + 2 KB code ring + 1.5 KB octet table).
// HI_ONLY (FLMR_NUMERICS_GPU_FP16: q_lo = 0): the lo products and the hi / lo combine are left out, same values
template <bool HI_ONLY>
__global__ __launch_bounds__(256, 2) void filter_stage2_xcd_kernel(flmr_filter_args f, const int32_t* __restrict__ pids, int64_t pid_stride,
                                                                    const int32_t* __restrict__ counts, float* __restrict__ part,
                                                                    int64_t part_stride, const _Float16* __restrict__ cen16,
                                                                    const _Float16* __restrict__ q_hi, const _Float16* __restrict__ q_lo,
                                                                    const int32_t* __restrict__ codes_sorted,
                                                                    const uint16_t* __restrict__ splits, int nsl, int G
#ifdef X2_PROFILE
                                                                    , long long* prof
#endif
) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef X2_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long ptiles = 0;
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#define X2_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define X2_STAMP(k) do { } while (0)
#endif
    const int L = blockIdx.x;
    const int rest = L >> 3;
    const int b = rest % f.nqueries, gg = rest / f.nqueries;
    const int g = gg % G;
    const int sl = (gg / G) * X2_XCDS + (L & (X2_XCDS - 1));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int cnt = counts[b];
    const int slot0 = (g * X2_WAVES + wave) * X2_DOCS;
    if (slot0 >= cnt) return;  // (no block-wide barrier below)
    const int nd = cnt - slot0 < X2_DOCS ? cnt - slot0 : X2_DOCS;
    // per wave: row buffers 2 x 8 KB | code ring 8 x 256 B | octet table: position of the octet's first token (u32) [X2_OCT],
    // then (tokens in the octet - 1) | passage lane << 3 (u16) [X2_OCT]
    char* const wbase = smem + (size_t)wave * X2_WAVE_LDS;
    char* const rowbuf = wbase;
    char* const ring = wbase + 16384;
    uint32_t* const opos = reinterpret_cast<uint32_t*>(wbase + 16384 + X2_RING * 256);
    uint16_t* const ometa = reinterpret_cast<uint16_t*>(wbase + 16384 + X2_RING * 256 + X2_OCT * 4);
    const uint32_t rowbuf_lds =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)rowbuf);
    const uint32_t ring_lds = rowbuf_lds + 16384;

    // ---- this lane's passage: its run of slice-`sl` tokens in the sorted copy ----
    int run_len = 0;
    uint32_t run_base = 0;  // positions fit 31 bits (flmr_build_sorted_codes)
    if (lane < nd) {
        const int pid = pids[(size_t)b * pid_stride + slot0 + lane];
        const int64_t off = f.offsets[pid];
        const int len = f.ulen_sorted ? (int)f.ulen_sorted[pid] : (int)(f.doclens ? f.doclens[pid] : (f.offsets[pid + 1] - off));
        const int start = splits[(size_t)pid * nsl + sl];
        const int end = sl < nsl - 1 ? (int)splits[(size_t)pid * nsl + sl + 1] : len;
        run_len = end - start;
        run_base = (uint32_t)(off + start);
    }
    const int noct = (run_len + 7) >> 3;  // octets of this passage
    int oend = noct;                      // -> inclusive prefix over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(oend, d, 64);
        if (lane >= d) oend += t;
    }
    if (__builtin_amdgcn_readlane(oend, 63) == 0) return;

    x2h8 bh[8], bl[8];
    {
        const x2h8* ph = reinterpret_cast<const x2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
        const x2h8* pl = reinterpret_cast<const x2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            bh[s] = ph[s];
            if constexpr (!HI_ONLY) bl[s] = pl[s];
        }
    }
    // every compiler-visible load lands here, before the first hand-counted one is issued
#pragma unroll
    for (int s = 0; s < 8; s++) {
        asm volatile("" : "+v"(bh[s])::"memory");
        if constexpr (!HI_ONLY) asm volatile("" : "+v"(bl[s])::"memory");
    }
    // part[query][survivor slot][slice][32]: the eight partial rows of a survivor are one contiguous KB for the combine kernel
    float* const prow = part + (((size_t)b * part_stride + slot0) * nsl + sl) * 32;
    const int pstep = nsl * 32;   // (<= 64 x 1024 floats: 32-bit arithmetic in the flush)
    uint32_t piece_off[8];  // byte offset, inside its row, of the 16-byte piece this lane moves in DMA instruction gq
#pragma unroll
    for (int gq = 0; gq < 8; gq++) piece_off[gq] = (uint32_t)(((lane & 15) ^ ((4 * gq + (lane >> 4)) & 15)) << 4);
    X2_STAMP(0);

    // Passages are taken in batches whose octets fit the table (one batch unless the runs are long: 256 octets = 64 tiles).
    int jstart = 0, obase = 0;
    while (jstart < 64) {
        const unsigned long long fits = __ballot(lane >= jstart && oend - obase <= X2_OCT);
        const int jend = jstart + __popcll(fits);  // oend is monotone: the lanes that fit are jstart .. jend-1 (>= 1: a run has <= 256 octets)
        const int ntot = __builtin_amdgcn_readlane(oend, jend - 1) - obase;  // octets of this batch
        if (lane >= jstart && lane < jend) {
            const int first = oend - noct - obase;
            for (int k = 0; k < noct; k++) {
                opos[first + k] = run_base + 8 * k;
                const int left = run_len - 8 * k;
                ometa[first + k] = (uint16_t)(((left < 8 ? left : 8) - 1) | (lane << 3));
            }
        }
        const int ntiles = (ntot + 3) >> 2;
#ifdef X2_PROFILE
        ptiles += ntiles;
#endif
        if (ntiles > 0) {
            // position (in codes_sorted) of the token in row (lane & 31) of tile t; octets past the end repeat the last one, a pad
            // repeats the octet's last token
            auto tile_pos = [&](int t) -> uint32_t {
                int o = 4 * t + ((lane >> 3) & 3);
                o = o < ntot ? o : ntot - 1;
                const int e = lane & 7, nv = ometa[o] & 7;
                return opos[o] + (e < nv ? e : nv);
            };
            // codes of tile t -> ring slot t % 8, straight to LDS (256 B: lanes 32..63 repeat lanes 0..31)
            auto issue_codes = [&](int t) {
                const int32_t* src = codes_sorted + tile_pos(t);
                const uint32_t dst = ring_lds + (t & (X2_RING - 1)) * 256;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
            };
            // 32 rows of tile t -> rowbuf[t & 1]: piece p of row r at position p ^ (r & 15); the codes come from the ring.
            // SGPR base + 32-bit byte offset (the table is K * 256 bytes < 4 GB); the offset's VALU op is the wait state
            // between the write of M0 and the DMA that reads it.
            auto issue_rows = [&](int t) {
#ifdef X2_NO_DMA   // development ablation (profiles/s2_ceiling.sh): no row gathers, everything else as it is
                return;
#endif
                const int* cr = reinterpret_cast<const int*>(ring + (t & (X2_RING - 1)) * 256) + (lane >> 4);
                int c[8];
#pragma unroll
                for (int gq = 0; gq < 8; gq++) c[gq] = cr[4 * gq];
#pragma unroll
                for (int gq = 0; gq < 8; gq++) {
                    const uint32_t dst = rowbuf_lds + (t & 1) * 8192 + gq * 1024;
                    uint32_t voff;
                    asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                                 : "=&v"(voff)
                                 : "s"(cen16), "s"(dst), "v"(c[gq]), "v"(piece_off[gq])
                                 : "memory", "m0");
                }
            };
            // VMEM issue order (every operation is ALWAYS issued -- positions past the end of the stream repeat its last octet,
            // whose rows are L2 hits -- so the counts below do not depend on the stream length):
            //   prologue  C0 .. C5, R0 (8), R1 (8)          Ct = codes of tile t (1 operation), Rt = rows of tile t (8 operations)
            //   step t    C(t+6), R(t+2)
            // top of step t: Rt must have landed; younger than it are R1 (t = 0) or all of step t-1: vmcnt(8) / vmcnt(9).
            // R(t+2) needs C(t+2), the first operation of step t-4 (or of the prologue): at least 19 operations older than the
            // youngest, so the wait at the top of the step has already covered it.
            // (The compiler's own stores of column maxima only add to the number of younger operations: a count stays safe.)
#pragma unroll
            for (int t = 0; t < X2_AHEAD; t++) issue_codes(t);
            x2_wait_vm<X2_AHEAD - 2>();
            asm volatile("" ::: "memory");
            issue_rows(0);
            issue_rows(1);
            // passage whose column maxima `cm` carries (per lane: over the rows of its own half-wave); starts at the first octet's
            int jcur = __builtin_amdgcn_readfirstlane((int)(ometa[0] >> 3));
            float cm = -9999.0f;
            auto flush = [&](int j) {
                const float v = flmr_xhalf_max(cm);
                if (h == 0) prow[(unsigned)(j * pstep) + i] = v;
            };
            for (int t = 0; t < ntiles; t++) {
                // ---- tile t's rows ----
                if (t == 0) x2_wait_vm<8>(); else x2_wait_vm<9>();
                X2_STAMP(1);
                // ONE LDS round trip per step: this tile's rows, its octets' passages, the codes of tile t+2 (their request, four
                // steps old, is older than anything the wait above lets through) and the octet entries of tile t+6
                x2h8 av[8];
#pragma unroll
                for (int s = 0; s < 8; s++)
                    av[s] = *reinterpret_cast<const x2h8*>(rowbuf + (t & 1) * 8192 + i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
                int om;  // lanes 0..3: passage lane of octets 4t .. 4t+3
                {
                    int o = 4 * t + (lane & 3);
                    o = o < ntot ? o : ntot - 1;
                    om = ometa[o] >> 3;
                }
                int c[8];
                {
                    const int* cr = reinterpret_cast<const int*>(ring + ((t + 2) & (X2_RING - 1)) * 256) + (lane >> 4);
#pragma unroll
                    for (int gq = 0; gq < 8; gq++) c[gq] = cr[4 * gq];
                }
                const uint32_t cpos = tile_pos(t + X2_AHEAD);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the row buffer has been read: it may be refilled)
                X2_STAMP(2);
                // ---- keep the pipeline full: codes of tile t+6, rows of tile t+2 ----
                {
                    const uint32_t dst = ring_lds + ((t + X2_AHEAD) & (X2_RING - 1)) * 256;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(codes_sorted + cpos), "s"(dst) : "memory", "m0");
                }
                X2_STAMP(3);
                X2_STAMP(4);
#if !defined(X2_LATE_DMA) && !defined(X2_NO_DMA)
#pragma unroll
                for (int gq = 0; gq < 8; gq++) {
                    const uint32_t dst = rowbuf_lds + (t & 1) * 8192 + gq * 1024;
                    uint32_t voff;
                    asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                                 : "=&v"(voff)
                                 : "s"(cen16), "s"(dst), "v"(c[gq]), "v"(piece_off[gq])
                                 : "memory", "m0");
                }
#endif
                X2_STAMP(5);
                // ---- 32 tokens x 32 query tokens, fp16-split: same MFMA sequence as stage 0 and the gather kernel ----
                x2f16 ah, al;
#pragma unroll
                for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#ifndef X2_NO_MFMA   // development ablation: no matrix products (the accumulators stay zero)
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[s], ah, 0, 0, 0);
                    if constexpr (!HI_ONLY) al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[s], al, 0, 0, 0);
                }
#else
#pragma unroll
                for (int s = 0; s < 8; s++) asm volatile("" :: "v"(av[s]));   // (the LDS reads of the rows stay)
#endif
#ifdef X2_NO_FOLD   // development ablation: the products are made and dropped (no octet maxima, no per-passage fold, no flush)
                asm volatile("" :: "v"(ah));
                if constexpr (!HI_ONLY) asm volatile("" :: "v"(al));
                continue;
#endif
                // rows 8k .. 8k+7 (octet k) live in registers 4k .. 4k+3 of the two half-waves: each lane keeps the maximum over
                // ITS four rows; the two halves are only combined when a passage is flushed (one LDS-crossbar op per passage
                // instead of four per tile)
                float mq[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (HI_ONLY) {
                        // (folded straight into `cm` below, as a v_max3 chain from an already canonical value: a maximum of raw
                        // MFMA outputs makes the compiler canonicalise each of them first -- five instructions per octet, not two)
                        mq[k] = 0.0f;
                    } else {
                        const float v0 = fmaf(al[4 * k], 1.0f / 2048.0f, ah[4 * k]), v1 = fmaf(al[4 * k + 1], 1.0f / 2048.0f, ah[4 * k + 1]);
                        const float v2 = fmaf(al[4 * k + 2], 1.0f / 2048.0f, ah[4 * k + 2]), v3 = fmaf(al[4 * k + 3], 1.0f / 2048.0f, ah[4 * k + 3]);
                        mq[k] = x2_max(x2_max3(v0, v1, v2), v3);
                    }
                }
#ifdef X2_PROFILE
                asm volatile("" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]));
#endif
                X2_STAMP(6);
                // ---- fold the octets into their passages (wave-uniform control flow) ----
                // (octets past the end of the stream -- last tile only -- repeat the last octet, passage and tokens alike: folding
                // them again changes nothing, so no octet needs a validity test)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = __builtin_amdgcn_readlane(om, k);
                    if (j != jcur) {
                        flush(jcur);
                        cm = -9999.0f;  // filter_pids.cpp:30-33
                        jcur = j;
                    }
                    if constexpr (HI_ONLY) {
                        // (compiler-visible: an inline-asm instruction reading the accumulators right after the MFMAs would have to
                        // carry the MFMA -> VALU wait states itself)
                        cm = fmaxf(fmaxf(cm, ah[4 * k]), ah[4 * k + 1]);
                        cm = fmaxf(fmaxf(cm, ah[4 * k + 2]), ah[4 * k + 3]);
                    } else {
                        cm = x2_max(cm, mq[k]);
                    }
#ifdef X2_LATE_DMA   // development form: the row pieces of tile t+2 issued in the fold's VALU-only stretches, two per octet
#pragma unroll
                    for (int gq = 2 * k; gq < 2 * k + 2; gq++) {
                        const uint32_t dst = rowbuf_lds + (t & 1) * 8192 + gq * 1024;
                        uint32_t voff;
                        asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                                     : "=&v"(voff)
                                     : "s"(cen16), "s"(dst), "v"(c[gq]), "v"(piece_off[gq])
                                     : "memory", "m0");
                    }
#endif
                }
                X2_STAMP(7);
            }
            flush(jcur);
            x2_wait_vm<0>();  // (also: the DMA writes of the tiles requested past the end have landed before the LDS is reused)
            asm volatile("" ::: "memory");
        }
        obase += ntot;
        jstart = jend;
        if (obase >= __builtin_amdgcn_readlane(oend, 63)) break;
    }
#ifdef X2_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)&prof[k], (unsigned long long)pt[k]);
        atomicAdd((unsigned long long*)&prof[8], (unsigned long long)ptiles);
        atomicAdd((unsigned long long*)&prof[9], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// The same kernel with the rows IN FLIGHT IN REGISTERS instead of LDS (build flag -DX2_ROWS_IN_REGS=<depth>, development
// form).  Why: the DMA form keeps two 8 KB row buffers per wave in LDS for as long as their rows are in flight, so LDS (160 KB)
// caps a CU at 8 waves x 2 tiles = 128 KB of outstanding rows -- and throughput = outstanding bytes / latency.  Here a tile is
// requested with plain 16-byte loads in the DMA's lane order (16 consecutive lanes per 256-byte row: the order the address path
// coalesces, profiles/microbench/s2_reg_probe), stays in 32 VGPRs per lane while in flight, and only passes THROUGH an 8 KB
// per-wave transit buffer (ds_write_b128 x 8, ds_read_b128 x 8 in MFMA operand order) when it is consumed: 9.5 KB of LDS per
// wave, `depth` tiles in flight per wave, occupancy set by registers.  Every load is compiler-visible (no hand-counted waits):
// the codes of tile t are requested 2 x depth tiles ahead, BEFORE the rows of tile t - 2 depth, so the in-order wait for those
// rows has always covered them; the register rings are indexed statically (the loop is unrolled 2 x depth times).
// Same rows, same MFMA sequence, same fold: bit-identical partial maxima.
// ------------------------------------------------------------------------------------------------
#ifdef X2_ROWS_IN_REGS
#ifndef X2_REG_MINW
#define X2_REG_MINW 2
#endif
#define X2R_WAVE_LDS (8192 + X2_OCT * 6)   // transit buffer + octet table
template <bool HI_ONLY>
__global__ __launch_bounds__(256, X2_REG_MINW) void filter_stage2_xreg_kernel(flmr_filter_args f, const int32_t* __restrict__ pids, int64_t pid_stride,
                                                                               const int32_t* __restrict__ counts, float* __restrict__ part,
                                                                               int64_t part_stride, const _Float16* __restrict__ cen16,
                                                                               const _Float16* __restrict__ q_hi, const _Float16* __restrict__ q_lo,
                                                                               const int32_t* __restrict__ codes_sorted,
                                                                               const uint16_t* __restrict__ splits, int nsl, int G) {
    constexpr int D = X2_ROWS_IN_REGS, CR = 2 * D, UN = CR;
    static_assert(D >= 1 && D <= 4, "row tiles in flight per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = blockIdx.x;
    const int rest = L >> 3;
    const int b = rest % f.nqueries, gg = rest / f.nqueries;
    const int g = gg % G;
    const int sl = (gg / G) * X2_XCDS + (L & (X2_XCDS - 1));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int cnt = counts[b];
    const int slot0 = (g * X2_WAVES + wave) * X2_DOCS;
    if (slot0 >= cnt) return;
    const int nd = cnt - slot0 < X2_DOCS ? cnt - slot0 : X2_DOCS;
    char* const wbase = smem + (size_t)wave * X2R_WAVE_LDS;
    char* const transit = wbase;
    uint32_t* const opos = reinterpret_cast<uint32_t*>(wbase + 8192);
    uint16_t* const ometa = reinterpret_cast<uint16_t*>(wbase + 8192 + X2_OCT * 4);
    int run_len = 0;
    uint32_t run_base = 0;
    if (lane < nd) {
        const int pid = pids[(size_t)b * pid_stride + slot0 + lane];
        const int64_t off = f.offsets[pid];
        const int len = f.ulen_sorted ? (int)f.ulen_sorted[pid] : (int)(f.doclens ? f.doclens[pid] : (f.offsets[pid + 1] - off));
        const int start = splits[(size_t)pid * nsl + sl];
        const int end = sl < nsl - 1 ? (int)splits[(size_t)pid * nsl + sl + 1] : len;
        run_len = end - start;
        run_base = (uint32_t)(off + start);
    }
    const int noct = (run_len + 7) >> 3;
    int oend = noct;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(oend, d, 64);
        if (lane >= d) oend += t;
    }
    if (__builtin_amdgcn_readlane(oend, 63) == 0) return;
    x2h8 bh[8], bl[8];
    {
        const x2h8* ph = reinterpret_cast<const x2h8*>(q_hi + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
        const x2h8* pl = reinterpret_cast<const x2h8*>(q_lo + ((size_t)b * f.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            bh[s] = ph[s];
            if constexpr (!HI_ONLY) bl[s] = pl[s];
        }
    }
    float* const prow = part + (((size_t)b * part_stride + slot0) * nsl + sl) * 32;
    const int pstep = nsl * 32;
    // lane -> (row 4 gq + (lane >> 4), 16-byte piece pbase ^ (4 gq & 15)) of load gq: piece p of row r lands at slot p ^ (r & 15)
    const uint32_t pbase = (uint32_t)((lane & 15) ^ (lane >> 4));
    const char* const cenb = reinterpret_cast<const char*>(cen16);

    int jstart = 0, obase = 0;
    while (jstart < 64) {
        const unsigned long long fits = __ballot(lane >= jstart && oend - obase <= X2_OCT);
        const int jend = jstart + __popcll(fits);
        const int ntot = __builtin_amdgcn_readlane(oend, jend - 1) - obase;
        if (lane >= jstart && lane < jend) {
            const int first = oend - noct - obase;
            for (int k = 0; k < noct; k++) {
                opos[first + k] = run_base + 8 * k;
                const int left = run_len - 8 * k;
                ometa[first + k] = (uint16_t)(((left < 8 ? left : 8) - 1) | (lane << 3));
            }
        }
        const int ntiles = (ntot + 3) >> 2;
        if (ntiles > 0) {
            auto tile_pos = [&](int t) -> uint32_t {
                int o = 4 * t + ((lane >> 3) & 3);
                o = o < ntot ? o : ntot - 1;
                const int e = lane & 7, nv = ometa[o] & 7;
                return opos[o] + (e < nv ? e : nv);
            };
            x2u4 vr[D][8];
            int cdreg[CR];
            auto load_rows = [&](x2u4 (&dst)[8], int code) __attribute__((always_inline)) {
#pragma unroll
                for (int gq = 0; gq < 8; gq++) {
                    const uint32_t c = (uint32_t)__shfl(code, 4 * gq + (lane >> 4), 64);
                    const uint32_t voff = (c << 8) | ((pbase ^ (uint32_t)((4 * gq) & 15)) << 4);
                    dst[gq] = *reinterpret_cast<const x2u4*>(cenb + voff);
                }
            };
            // prologue in the steady state's issue order (codes of the first 2 depth tiles, then the rows tile by tile), pinned
            // by compiler barriers: the loop header merges this state with the back edge's
#pragma unroll
            for (int t = 0; t < CR; t++) cdreg[t] = codes_sorted[tile_pos(t)];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < D; t++) {
                load_rows(vr[t], cdreg[t]);
                asm volatile("" ::: "memory");
            }
            int jcur = __builtin_amdgcn_readfirstlane((int)(ometa[0] >> 3));
            float cm = -9999.0f;
            auto flush = [&](int j) {
                const float v = flmr_xhalf_max(cm);
                if (h == 0) prow[(unsigned)(j * pstep) + i] = v;
            };
            auto step = [&](auto U_, int t) __attribute__((always_inline)) {
                constexpr int U = decltype(U_)::value;
                // ---- tile t: registers -> transit buffer (the DMA's layout: 1 KB per load, lanes in order) ----
#pragma unroll
                for (int gq = 0; gq < 8; gq++) *reinterpret_cast<x2u4*>(transit + gq * 1024 + lane * 16) = vr[U % D][gq];
                // ---- keep the pipeline full: codes of tile t + 2 depth, rows of tile t + depth ----
                cdreg[U % CR] = codes_sorted[tile_pos(t + CR)];
                load_rows(vr[U % D], cdreg[(U + D) % CR]);
                // ---- tile t in MFMA operand order, its octets' passages ----
                x2h8 av[8];
#pragma unroll
                for (int s = 0; s < 8; s++)
                    av[s] = *reinterpret_cast<const x2h8*>(transit + i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
                int om;
                {
                    int o = 4 * t + (lane & 3);
                    o = o < ntot ? o : ntot - 1;
                    om = ometa[o] >> 3;
                }
                x2f16 ah, al;
#pragma unroll
                for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[s], ah, 0, 0, 0);
                    if constexpr (!HI_ONLY) al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[s], al, 0, 0, 0);
                }
                float mq[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (HI_ONLY) {
                        mq[k] = 0.0f;
                    } else {
                        const float v0 = fmaf(al[4 * k], 1.0f / 2048.0f, ah[4 * k]), v1 = fmaf(al[4 * k + 1], 1.0f / 2048.0f, ah[4 * k + 1]);
                        const float v2 = fmaf(al[4 * k + 2], 1.0f / 2048.0f, ah[4 * k + 2]), v3 = fmaf(al[4 * k + 3], 1.0f / 2048.0f, ah[4 * k + 3]);
                        mq[k] = x2_max(x2_max3(v0, v1, v2), v3);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = __builtin_amdgcn_readlane(om, k);
                    if (j != jcur) {
                        flush(jcur);
                        cm = -9999.0f;  // filter_pids.cpp:30-33
                        jcur = j;
                    }
                    if constexpr (HI_ONLY) {
                        cm = fmaxf(fmaxf(cm, ah[4 * k]), ah[4 * k + 1]);
                        cm = fmaxf(fmaxf(cm, ah[4 * k + 2]), ah[4 * k + 3]);
                    } else {
                        cm = x2_max(cm, mq[k]);
                    }
                }
            };
            // (a step is only reachable through the one before it: the compiler's wait counts then know what is in flight)
            for (int t0 = 0; t0 < ntiles; t0 += UN) {
                step(std::integral_constant<int, 0>{}, t0);
#define X2R_STEP(U) if constexpr (U < UN) { if (t0 + U >= ntiles) break; step(std::integral_constant<int, U>{}, t0 + U); }
                X2R_STEP(1) X2R_STEP(2) X2R_STEP(3) X2R_STEP(4) X2R_STEP(5) X2R_STEP(6) X2R_STEP(7)
#undef X2R_STEP
            }
            flush(jcur);
        }
        obase += ntot;
        jstart = jend;
        if (obase >= __builtin_amdgcn_readlane(oend, 63)) break;
    }
}
#endif

// one half-wave per survivor: maximum over the slices that hold tokens, k-ascending sum of the first nqc columns, key.
// grid = (nqueries, ceil(max_count / 8)), block = 256
template <int NSL>
__global__ __launch_bounds__(256) void s2_combine_kernel(flmr_filter_args f, const int32_t* __restrict__ pids, int64_t pid_stride,
                                                         const int32_t* __restrict__ counts, const float* __restrict__ part,
                                                         int64_t part_stride, const uint16_t* __restrict__ splits,
                                                         uint64_t* __restrict__ keys, int64_t key_stride) {
    __shared__ float tr[8][33];
    const int b = blockIdx.x;
    const int grp = threadIdx.x >> 5, i = threadIdx.x & 31;
    const int d = blockIdx.y * 8 + grp;
    const bool live = d < counts[b];
    const int qlen = f.q_lens ? f.q_lens[b] : f.nq_cand;
    const int nqc = qlen < f.nq_cand ? qlen : f.nq_cand;
    float m = -9999.0f;
    int pid = 0;
    if (live) {
        pid = pids[(size_t)b * pid_stride + d];
        const int64_t off = f.offsets[pid];
        const int len = f.ulen_sorted ? (int)f.ulen_sorted[pid] : (int)(f.doclens ? f.doclens[pid] : (f.offsets[pid + 1] - off));
        uint32_t w[NSL / 2];
#pragma unroll
        for (int q4 = 0; q4 < NSL / 8; q4++) {
            const uint4 sp = *reinterpret_cast<const uint4*>(splits + (size_t)pid * NSL + 8 * q4);
            w[4 * q4] = sp.x; w[4 * q4 + 1] = sp.y; w[4 * q4 + 2] = sp.z; w[4 * q4 + 3] = sp.w;
        }
        // all rows are requested before the first is used (a slice without tokens was never written: its row is read and dropped)
        float v[NSL];
#pragma unroll
        for (int s = 0; s < NSL; s++) v[s] = __builtin_nontemporal_load(part + (((size_t)b * part_stride + d) * NSL + s) * 32 + i);
#pragma unroll
        for (int s = 0; s < NSL; s++) {
            const int start = (int)((w[s >> 1] >> (16 * (s & 1))) & 0xffffu);
            const int end = s < NSL - 1 ? (int)((w[(s + 1) >> 1] >> (16 * ((s + 1) & 1))) & 0xffffu) : len;
            m = end > start ? fmaxf(m, v[s]) : m;
        }
    }
    tr[grp][i] = m;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live && i == 0) keys[(size_t)b * key_stride + d] = flmr_make_key(flmr_seq_sum(tr[grp], nqc, f.f16_round), pid);
}

#ifdef X2_PROFILE
long long* x2_prof_buffer = nullptr;  // set by the stand-alone harness (profiles/microbench/s2_xcd_probe.hip)
#endif

// the sliced form pays when the centroid table does not fit an XCD's L2 anyway (the gather kernel then reads it from the
// fabric) AND consecutive workgroups really land on the eight XCDs in turn (probed at index open)
bool flmr_stage2_xcd_pays(const flmr_index* ix) {
    const size_t table = (size_t)ix->K * FLMR_DIM * sizeof(_Float16);
    return ix->doc_splits && ix->codes_sorted && ix->xcd_round_robin && table > ((size_t)6 << 20) &&
           table < ((size_t)1 << 32);  // 32-bit row offsets
}

size_t flmr_stage2_xcd_part_floats(const flmr_index* ix, int64_t nqueries, int64_t ndocs) {
    return (size_t)nqueries * (size_t)ix->nslices * ndocs * 32;
}

int flmr_launch_filter_stage2_xcd(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                  int32_t max_count, uint64_t* keys, int64_t key_stride, const flmr_index* ix,
                                  const _Float16* q_hi, const _Float16* q_lo, float* part, int64_t part_stride, hipStream_t st) {
    return flmr_launch_filter_stage2_xcd_ex(f, pids, pid_stride, counts, max_count, keys, key_stride, ix, q_hi, q_lo, part, part_stride,
                                            f.f16_round != 0, st);
}

int flmr_launch_filter_stage2_xcd_ex(const flmr_filter_args& f, const int32_t* pids, int64_t pid_stride, const int32_t* counts,
                                     int32_t max_count, uint64_t* keys, int64_t key_stride, const flmr_index* ix,
                                     const _Float16* q_hi, const _Float16* q_lo, float* part, int64_t part_stride, bool hi_only,
                                     hipStream_t st) {
    if (max_count <= 0) return FLMR_OK;
    if (f.ncol != 32) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 recompute needs one column tile");
    if (!ix->doc_splits || !ix->codes_sorted || !part) FLMR_FAIL(FLMR_ERR_INVALID, "stage-2 sliced kernel needs the sorted codes and their split table");
    if ((size_t)ix->K * FLMR_DIM * sizeof(_Float16) >= ((size_t)1 << 32)) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "stage-2 sliced kernel: centroid table >= 4 GB");
    const int G = (int)flmr_ceil_div(max_count, X2_WAVES * X2_DOCS);
    const int nsl = ix->nslices;
    const int64_t grid = (int64_t)nsl * f.nqueries * G;
    if (grid > 0x7fffffffLL) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "stage-2 grid too large");
#ifdef X2_ROWS_IN_REGS
    const size_t lds = (size_t)X2_WAVES * X2R_WAVE_LDS;
#define X2_KERNEL filter_stage2_xreg_kernel
#elif defined(X2_LDS_PAD)   // development probe: a larger request leaves one workgroup per CU (half the tiles in flight, one wave per SIMD)
    const size_t lds = (size_t)X2_WAVES * X2_WAVE_LDS + X2_LDS_PAD;
#define X2_KERNEL filter_stage2_xcd_kernel
#else
    const size_t lds = (size_t)X2_WAVES * X2_WAVE_LDS;
#define X2_KERNEL filter_stage2_xcd_kernel
#endif
    if (hi_only) {
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(X2_KERNEL<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(X2_KERNEL<true>, dim3((unsigned)grid), dim3(256), lds, st, f, pids, pid_stride, counts, part, part_stride,
                           ix->centroids_f16, q_hi, q_lo, ix->codes_sorted, ix->doc_splits, nsl, G
#ifdef X2_PROFILE
                           , x2_prof_buffer
#endif
        );
    } else {
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(X2_KERNEL<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(X2_KERNEL<false>, dim3((unsigned)grid), dim3(256), lds, st, f, pids, pid_stride, counts, part, part_stride,
                           ix->centroids_f16, q_hi, q_lo, ix->codes_sorted, ix->doc_splits, nsl, G
#ifdef X2_PROFILE
                           , x2_prof_buffer
#endif
        );
    }
    const dim3 cgrid(f.nqueries, (unsigned)flmr_ceil_div(max_count, 8));
    if (nsl == 8)
        hipLaunchKernelGGL(s2_combine_kernel<8>, cgrid, dim3(256), 0, st, f, pids, pid_stride, counts, part, part_stride, ix->doc_splits, keys, key_stride);
    else if (nsl == 16)
        hipLaunchKernelGGL(s2_combine_kernel<16>, cgrid, dim3(256), 0, st, f, pids, pid_stride, counts, part, part_stride, ix->doc_splits, keys, key_stride);
    else if (nsl == 24)
        hipLaunchKernelGGL(s2_combine_kernel<24>, cgrid, dim3(256), 0, st, f, pids, pid_stride, counts, part, part_stride, ix->doc_splits, keys, key_stride);
    else
        hipLaunchKernelGGL(s2_combine_kernel<32>, cgrid, dim3(256), 0, st, f, pids, pid_stride, counts, part, part_stride, ix->doc_splits, keys, key_stride);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
