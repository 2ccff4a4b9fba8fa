// Stage 0 of the search path: centroid scores, probed cells, IVF union -> candidate pids.
//
// Reference: TPC/search/candidate_generation.py:12-20 (get_cells: centroids @ Q.T, per-token top-ncells,
// unique), :31-37 + TPC/search/segmented_lookup.cpp (IVF ragged gather), :45-64 (sort + unique_consecutive)
// and TPC/search/index_storage.py:116 (idx = max_j score >= thr).
//
// MI355X design
//   * centroid scores are the one dense contraction of the path: [K,128] x [128, nqueries*nq_cand].  It runs on
//     the fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain) with the centroid tile held in
//     registers for the whole query loop, so the 4*128*K-byte centroid matrix is read once per row block, not once
//     per query.  The epilogue fuses (a) the fp32 score-table store (the only HBM-sized output: 4*K*ncol bytes
//     per query, read back sparsely by S1/S2), (b) the row-max >= thr test packed to one bit per centroid
//     (a wave's 32 rows = one 32-bit word), (c) the per-column top-ncells of the block's 128 rows.
//   * cells -> candidates: IVF lists are OR-ed into a per-query passage bitmap (idempotent, so duplicates
//     across cells cost nothing) and compacted with popcount prefix sums into an ascending pid list: the
//     reference's concatenate + sort + unique without a sort.
#include "flmr_device.h"

// ------------------------------------------------------------------------------------------------
// cross-wave merge of the per-column top lists + store of the block partial
// ------------------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void s0_block_merge_store(flmr_toplist<NC>& tl, float* lds_v, int* lds_i, int wave, int lane,
                                                     float* part_val, int32_t* part_idx, size_t part_base) {
    // lds layout [4 waves][32 cols][NC]
    if (lane < 32) {
#pragma unroll
        for (int t = 0; t < NC; t++) {
            lds_v[(wave * 32 + lane) * NC + t] = tl.v[t];
            lds_i[(wave * 32 + lane) * NC + t] = tl.id[t];
        }
    }
    __syncthreads();
    if (wave == 0 && lane < 32) {
        for (int w = 1; w < 4; w++) {
#pragma unroll
            for (int t = 0; t < NC; t++) tl.insert(lds_v[(w * 32 + lane) * NC + t], lds_i[(w * 32 + lane) * NC + t]);
        }
#pragma unroll
        for (int t = 0; t < NC; t++) {
            part_val[part_base + (size_t)lane * NC + t] = tl.v[t];
            part_idx[part_base + (size_t)lane * NC + t] = tl.id[t];
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// S0a (MFMA): grid = (ceil(K/128), QSPLIT), block = 256 (4 waves x 32 centroid rows)
// A operand (centroids) lane (i = lane&31, h = lane>>5) holds dims 64h..64h+63 of row i: the MFMA
// contraction index is a permutation of the embedding dims (k-step s pairs dims s and 64+s), which is
// just another valid fp32 summation order.
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void s0_centroid_scores_mfma(flmr_s0_args a) {
    __shared__ float lds_v[4 * 32 * NC];
    __shared__ int lds_i[4 * 32 * NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 128 + wave * 32;
    const int T = a.ncol >> 5;

    float av[64];
    {
        const int arow = row0 + i;
        if (arow < a.K) {
            const float4* p = reinterpret_cast<const float4*>(a.centroids + (size_t)arow * FLMR_DIM + 64 * h);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                float4 v = p[t];
                av[4 * t + 0] = v.x; av[4 * t + 1] = v.y; av[4 * t + 2] = v.z; av[4 * t + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 64; t++) av[t] = 0.0f;
        }
    }

    for (int b = blockIdx.y; b < a.nqueries; b += gridDim.y) {
        const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
        const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
        float rmax[16];
#pragma unroll
        for (int r = 0; r < 16; r++) rmax[r] = FLMR_NEG_INF;
        float* cs_b = a.cs + (size_t)b * a.K * a.ncol;

        for (int ct = 0; ct < T; ct++) {
            const int col = ct * 32 + i;
            const bool colok = col < nqc;
            float bv[64];
            if (colok) {
                const float4* p = reinterpret_cast<const float4*>(a.Q + ((size_t)b * a.nq + col) * FLMR_DIM + 64 * h);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    float4 v = p[t];
                    bv[4 * t + 0] = v.x; bv[4 * t + 1] = v.y; bv[4 * t + 2] = v.z; bv[4 * t + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 64; t++) bv[t] = 0.0f;
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 64; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);

            flmr_toplist<NC> tl;
            tl.init();
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout of the 32x32 MFMA
                const int grow = row0 + row;
                const float v = acc[r];
                if (grow < a.K) {
                    cs_b[(size_t)grow * a.ncol + col] = v;
                    if (colok) tl.insert(v, grow);
                }
                rmax[r] = fmaxf(rmax[r], flmr_half_wave_max(colok ? v : FLMR_NEG_INF));
            }
            tl.merge_xor(32);
            const size_t part_base = (((size_t)b * a.nblk + blockIdx.x) * a.ncol + ct * 32) * NC;
            s0_block_merge_store<NC>(tl, lds_v, lds_i, wave, lane, a.part_val, a.part_idx, part_base);
        }
        // idx bits: this wave's 32 rows are exactly one word
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row0 + row < a.K && rmax[r] >= a.thr) w |= 1u << row;
        }
        w |= (uint32_t)__shfl_xor((int)w, 32, 64);
        if (lane == 0 && row0 < a.K) a.idx_bits[(size_t)b * a.idx_words + (row0 >> 5)] = w;
    }
}

// ------------------------------------------------------------------------------------------------
// S0a (fp16-split MFMA, default): centroids of a reference-format index are fp16 values (centroids.pt is saved
// as half, residual.py:161), so C is EXACT in fp16; Q is split as q_hi + q_lo*2^-11 with both halves fp16.  Every
// fp16 x fp16 product is exact in fp32 and the MFMA accumulates in fp32, so
//     C.q  ~=  mfma16(C, q_hi) + 2^-11 * mfma16(C, q_lo)
// carries ~2^-22 relative error per term (fp32-roundoff class, far inside the 1e-4 score tolerance) at 1/8 of
// the fp32-MFMA issue cost (2 x v_mfma_f32_32x32x16_f16 @ 32 cycles vs 8 x v_mfma_f32_32x32x2_f32 @ 64 per 16 dims).
// The MFMA pairs element e of k-half h of A with element e of k-half h of B, so any (h,e)->dim map used for
// BOTH operands is a valid contraction order: lane (i, h) holds dims 64h..64h+63 of its row, step s uses
// dims 64h+8s..64h+8s+7.
// A wave owns 128 centroid rows (4 row tiles, A image = 128 VGPRs) for the whole query loop and loads the query
// tile once per 128 rows; the per-column top lists are carried across the 4 row tiles in registers, so the kernel
// has no LDS and no block barrier.  grid = (ceil(K/512), QSPLIT), block = 256.
// ------------------------------------------------------------------------------------------------

// hi_only (FLMR_NUMERICS_GPU_FP16): the query is ROUNDED to fp16 like the reference's `Q.cuda().half()`
// (candidate_generation.py:50-52): q_lo = 0 and every split kernel then computes the fp32-accumulated fp16 product.
__global__ __launch_bounds__(256) void s0_split_q(const float* Q, const int32_t* q_lens, int nq, int nq_cand, int ncol,
                                                  _Float16* q_hi, _Float16* q_lo, int hi_only) {
    const int b = blockIdx.y;
    const int qlen = q_lens ? q_lens[b] : nq;
    const int nqc = qlen < nq_cand ? qlen : nq_cand;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ncol * FLMR_DIM; e += gridDim.x * blockDim.x) {
        const int col = e / FLMR_DIM;
        float v = 0.0f;
        if (col < nqc) v = Q[((size_t)b * nq + col) * FLMR_DIM + (e % FLMR_DIM)];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = hi_only ? (_Float16)0.0f : (_Float16)((v - (float)hi) * 2048.0f);
        q_hi[(size_t)b * ncol * FLMR_DIM + e] = hi;
        q_lo[(size_t)b * ncol * FLMR_DIM + e] = lo;
    }
}

#ifndef S0_RT
#define S0_RT 2                 // row tiles (of 32 centroids) per wave: A image = 64 VGPRs
#endif
#ifndef S0_CH
#define S0_CH 6                 // (query, column-tile) items whose B operand is staged in LDS per block barrier
#endif
#ifndef S0_WAVES
#define S0_WAVES 8               // waves per block (512 threads): two per SIMD, so one wave's MFMAs overlap the other's epilogue
#endif
#ifndef S0_OCC
#define S0_OCC 2                // workgroups per CU the register budget is sized for
#endif
#define S0_BROW 136             // halfs per staged B row (128 + 8 pad: conflict-free ds_read_b128 across rows)
#ifndef S0_DMA_B
#define S0_DMA_B 1              // 1: B operands DMA-ed into two alternating LDS buffers of S0_DCH items each (global_load_lds,
#endif                          //    rows unpadded, 16-byte pieces XOR-swizzled by row), the next chunk in flight while the
#define S0_DCH 3                //    current one is multiplied; 0: staged through registers, one buffer of S0_CH items
#define S0_LDS_STRIDE 36        // floats per staged row: 16-byte aligned rows for ds_read_b128

// ARGMAX = true (index build, flmr_nearest_centroids): additionally tracks the row index of every block maximum in
// part_idx (first row on ties); the search path instantiates ARGMAX = false and pays nothing for it.
// SPARSE = true (single column tile, only the rows of surviving centroids are stored): the ">= thr" test is a v_cmp per
// accumulator register whose 64-bit lane mask splits into the two rows the register holds (lanes 0-31 / 32-63), so the idx
// tile-level "anything survives?" test is scalar; only a tile that does hold a surviving row (rare) goes through the LDS
// staging + row stores of the dense epilogue.  That halves the kernel's LDS traffic, which at 16 KB of B fragments +
// 8 KB of staging per (wave, query) was as long as its MFMA time.
template <bool ARGMAX, bool SPARSE>
__global__ __launch_bounds__(64 * S0_WAVES, S0_OCC) void s0_centroid_scores_f16(flmr_s0_args a) {
    // dynamic LDS: [4 waves][32 rows][36 f32] staging tiles for the row-contiguous table stores, then the fp16 B
    // operands (q_hi, q_lo) of S0_CH (query, column-tile) items, loaded once per block and shared by its 4 waves.
    // Keeping B out of the global-load path matters: on CDNA4 vmcnt counts loads AND stores and retires in order, so
    // a wave that waits for a B load issued after its table stores also waits for those stores to be acknowledged
    // (~us each iteration); with B in LDS the only vmcnt waits left are the 63-deep rolling window of the stores.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * S0_LDS_STRIDE;  // (!SPARSE only)
    _Float16* bq = reinterpret_cast<_Float16*>(smem + S0_WAVES * 32 * S0_LDS_STRIDE * sizeof(float));
    const int wtile = blockIdx.x * S0_WAVES + wave;  // (32*S0_RT)-row tile index == partial block index
    const int row0 = wtile * 32 * S0_RT;
    const bool active = row0 < a.K;  // the launcher guarantees K % (32*S0_RT) == 0; idle waves still join the barriers
    const int T = a.ncol >> 5;

    f16x8 av[S0_RT][8];
#pragma unroll
    for (int rt = 0; rt < S0_RT; rt++) {
        const float4* p = reinterpret_cast<const float4*>(a.centroids + (size_t)((active ? row0 : 0) + rt * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const float4 x = p[2 * s], y = p[2 * s + 1];
            av[rt][s][0] = (_Float16)x.x; av[rt][s][1] = (_Float16)x.y; av[rt][s][2] = (_Float16)x.z; av[rt][s][3] = (_Float16)x.w;
            av[rt][s][4] = (_Float16)y.x; av[rt][s][5] = (_Float16)y.y; av[rt][s][6] = (_Float16)y.z; av[rt][s][7] = (_Float16)y.w;
        }
    }

    // flattened (query, column-tile) item loop, S0_CH items per LDS chunk
    const int nb = (a.nqueries - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int niter = nb * T;
    uint32_t idxw[S0_RT];  // the 32-bit idx word of each row tile, OR-ed over the column tiles of one query
#pragma unroll
    for (int rt = 0; rt < S0_RT; rt++) idxw[rt] = 0u;
#if S0_DMA_B
    // B staging by LDS-DMA, double-buffered: piece z of a chunk is 1 KB = rows 4g .. 4g+3 of one (item, hi|lo) image, contiguous
    // in q_hi / q_lo; wave w issues pieces 6w .. 6w+5 of the 48.  One barrier per chunk: behind it every wave's pieces of THIS
    // chunk have landed (each wave waits for its own with vmcnt(0) first) and every wave is done reading the OTHER buffer, which
    // the next chunk's DMA then overwrites while this one is multiplied.  (Staged through registers -- the S0_DMA_B=0 path --
    // the load latency and the LDS writes of every chunk were exposed between two barriers: a third of the wave cycles of this
    // kernel were spent parked, profiles/r02_pmc_summary.csv.)
    static_assert(S0_WAVES * 6 == S0_DCH * 16, "piece assignment assumes 48 pieces per chunk over 8 waves");
    char* const bqb = reinterpret_cast<char*>(bq);
    auto dma_chunk = [&](int c0, int buf) {
#pragma unroll
        for (int kz = 0; kz < 6; kz++) {
            const int z = wave * 6 + kz, item = z >> 4, hl = (z >> 3) & 1, g = z & 7;
            const int itx = c0 + item;
            if (itx < niter) {  // wave-uniform
                const int bb = blockIdx.y + (itx / T) * gridDim.y, row = 4 * g + (lane >> 4);
                const _Float16* src = (hl ? a.q_lo : a.q_hi) + ((size_t)bb * a.ncol + (itx % T) * 32 + row) * FLMR_DIM +
                                      (((lane & 15) ^ (row & 15)) << 3);
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(bqb + (((buf * S0_DCH + item) * 2 + hl) * 8192 + g * 1024)),
                                                 16, 0, 0);
            }
        }
    };
    dma_chunk(0, 0);
    for (int c0 = 0, cit = 0; c0 < niter; c0 += S0_DCH, cit++) {
      const int nch = (niter - c0) < S0_DCH ? (niter - c0) : S0_DCH;
      const int buf = cit & 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the chunk (and its older stores) are done
      __syncthreads();
      if (c0 + S0_DCH < niter) dma_chunk(c0 + S0_DCH, buf ^ 1);
      if (active)
      for (int j = 0; j < nch; j++) {
        const int it = c0 + j;
        const int b = blockIdx.y + (it / T) * gridDim.y, ct = it % T;
        f16x8 bh[8], bl[8];
        {
            const char* ph = bqb + (((buf * S0_DCH + j) * 2 + 0) * 8192) + i * 256;
            const char* pl = bqb + (((buf * S0_DCH + j) * 2 + 1) * 8192) + i * 256;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                bh[s] = *reinterpret_cast<const f16x8*>(ph + (((8 * h + s) ^ (i & 15)) << 4));
                bl[s] = *reinterpret_cast<const f16x8*>(pl + (((8 * h + s) ^ (i & 15)) << 4));
            }
        }
#else
    for (int c0 = 0; c0 < niter; c0 += S0_CH) {
      const int nch = (niter - c0) < S0_CH ? (niter - c0) : S0_CH;
      __syncthreads();  // every wave is done reading the previous chunk
      for (int e = threadIdx.x; e < nch * 1024; e += 64 * S0_WAVES) {  // 16-byte pieces: [item][hi|lo][32 rows][16 pieces]
          const int piece = e & 15, row = (e >> 4) & 31, hl = (e >> 9) & 1, item = e >> 10;
          const int itx = c0 + item;
          const int bb = blockIdx.y + (itx / T) * gridDim.y, colx = (itx % T) * 32 + row;
          const _Float16* src = (hl ? a.q_lo : a.q_hi) + ((size_t)bb * a.ncol + colx) * FLMR_DIM + piece * 8;
          *reinterpret_cast<f16x8*>(bq + ((item * 2 + hl) * 32 + row) * S0_BROW + piece * 8) = *reinterpret_cast<const f16x8*>(src);
      }
      __syncthreads();
      if (active)
      for (int j = 0; j < nch; j++) {
        const int it = c0 + j;
        const int b = blockIdx.y + (it / T) * gridDim.y, ct = it % T;
        f16x8 bh[8], bl[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            bh[s] = *reinterpret_cast<const f16x8*>(bq + ((j * 2 + 0) * 32 + i) * S0_BROW + 64 * h + 8 * s);
            bl[s] = *reinterpret_cast<const f16x8*>(bq + ((j * 2 + 1) * 32 + i) * S0_BROW + 64 * h + 8 * s);
        }
#endif
        const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
        const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
        float* cs_b = a.cs + (size_t)b * a.K * a.ncol;
        const int col = ct * 32 + i;
        const int c4 = (lane & 7) * 4;             // first of the 4 columns this lane stores
        const int nvalid4 = nqc - (ct * 32 + c4);  // how many of them are real query tokens
        const bool full_cols = nqc >= ct * 32 + 32;
        const bool sparse = !a.full_table && T == 1;
        float cmax = FLMR_NEG_INF;                 // running max of column `col` over this wave's rows
        int carg = 0x7fffffff;                     // (ARGMAX) its row
        // software pipeline over the row tiles: the MFMAs of tile rt+1 are issued BEFORE the epilogue of tile rt, so the
        // epilogue's VALU / LDS / store instructions fill the matrix pipe's shadow (one wave per SIMD cannot rely on
        // another wave for that overlap)
        f32x16 acc_h[2], acc_l[2];
        auto tile_mfma = [&](int rt, f32x16& oh, f32x16& ol) {
#pragma unroll
            for (int r = 0; r < 16; r++) { oh[r] = 0.0f; ol[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                oh = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][s], bh[s], oh, 0, 0, 0);
                ol = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][s], bl[s], ol, 0, 0, 0);
            }
        };
        tile_mfma(0, acc_h[0], acc_l[0]);
#pragma unroll
        for (int rt = 0; rt < S0_RT; rt++) {
            if (rt + 1 < S0_RT) tile_mfma(rt + 1, acc_h[(rt + 1) & 1], acc_l[(rt + 1) & 1]);
            const f32x16& ah = acc_h[rt & 1];
            const f32x16& al = acc_l[rt & 1];
            // epilogue, 3 VALU ops per score: combine hi/lo, column max, stage row-major in LDS
            const int rbase = row0 + rt * 32;
            bool staged = true;
            if constexpr (SPARSE) {
                // lanes whose column is a real query token (both halves): rows only count those (index_storage.py:116)
                const unsigned long long colmask = full_cols ? ~0ull : (((1ull << nqc) - 1ull) * 0x100000001ull);
                // "does any row of this tile survive" == "is any valid lane's maximum over its 16 rows >= thr": the tile
                // maximum (v_max3 chain) also feeds the column maximum, so the test costs one compare per tile
                float tmax = FLMR_NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                    if constexpr (ARGMAX) {
                        if (v > cmax) { cmax = v; carg = rbase + (r & 3) + 8 * (r >> 2) + 4 * h; }
                    }
                    tmax = fmaxf(tmax, v);
                }
                if constexpr (!ARGMAX) cmax = fmaxf(cmax, tmax);
                // wave-uniform and rare: some row of this tile survives -> the staged epilogue below stores it and sets
                // its idx bit (its column max / argmax updates are idempotent)
                staged = (__ballot(tmax >= a.thr) & colmask) != 0ull;
            }
            if (staged) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                if constexpr (ARGMAX) {
                    if (v > cmax) { cmax = v; carg = rbase + lrow; }  // rows are visited in ascending order: first maximum
                } else {
                    cmax = fmaxf(cmax, v);
                }
                stage[lrow * S0_LDS_STRIDE + i] = v;
            }
            __builtin_amdgcn_wave_barrier();  // DS ops of one wave execute in order: only the compiler must not reorder
            // 4 x (8 rows x 128 B) fully coalesced stores: lane l -> row (l>>3)+8m, columns c4..c4+3; the same
            // row-major view gives the ">= thr" test of 8 rows per ballot (index_storage.py:116)
#pragma unroll
            for (int mrow = 0; mrow < 4; mrow++) {
                const int R = (lane >> 3) + 8 * mrow;
                const float4 v4 = *reinterpret_cast<const float4*>(stage + R * S0_LDS_STRIDE + c4);
                float m4;
                if (full_cols) {  // wave-uniform: every column of this tile is a real query token
                    m4 = fmaxf(fmaxf(v4.x, v4.y), fmaxf(v4.z, v4.w));
                } else {
                    m4 = nvalid4 > 0 ? v4.x : FLMR_NEG_INF;
                    m4 = fmaxf(m4, nvalid4 > 1 ? v4.y : FLMR_NEG_INF);
                    m4 = fmaxf(m4, nvalid4 > 2 ? v4.z : FLMR_NEG_INF);
                    m4 = fmaxf(m4, nvalid4 > 3 ? v4.w : FLMR_NEG_INF);
                }
                unsigned long long bal = __ballot(m4 >= a.thr);  // byte j of `bal` = the 8 lanes of row 8*mrow + j
                bal |= bal >> 4; bal |= bal >> 2; bal |= bal >> 1;
                bal &= 0x0101010101010101ull;
                const uint32_t byte = (uint32_t)((bal * 0x0102040810204080ull) >> 56);  // bit j = row 8*mrow + j hit
                idxw[rt] |= byte << (8 * mrow);
                // table store.  Sparse mode keeps only the rows of surviving centroids (all stage 1 ever reads; stage 2
                // and the cell selection recompute what they need from the fp16 centroids), which removes the
                // 4*K*32-byte-per-query table write -- the largest HBM stream of the whole path.
                // (a.cs == NULL: the sparse path's compact form -- the surviving rows are computed by qualifying_kernel, nothing is stored here)
                if (a.cs && (!sparse || ((byte >> (lane >> 3)) & 1u)))
                    *reinterpret_cast<float4*>(cs_b + (size_t)(rbase + R) * a.ncol + ct * 32 + c4) = v4;
            }
            __builtin_amdgcn_wave_barrier();
            }  // staged
        }
        // block maximum of each column (this wave's 32*S0_RT rows): the cell selection re-reads only the winners
        if constexpr (ARGMAX) {
            const float ov = __shfl_xor(cmax, 32, 64);
            const int oa = __shfl_xor(carg, 32, 64);
            if (ov > cmax || (ov == cmax && oa < carg)) { cmax = ov; carg = oa; }
            if (lane < 32) a.part_idx[((size_t)b * a.nblk + wtile) * a.ncol + col] = carg;
        } else {
            cmax = flmr_xhalf_max(cmax);
        }
        if (lane < 32) a.part_val[((size_t)b * a.nblk + wtile) * a.ncol + col] = (col < nqc) ? cmax : FLMR_NEG_INF;
        if (ct == T - 1) {
#pragma unroll
            for (int rt = 0; rt < S0_RT; rt++) {
                if (lane == 0) a.idx_bits[(size_t)b * a.idx_words + ((row0 + rt * 32) >> 5)] = idxw[rt];
                idxw[rt] = 0u;
            }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// S0a, QUERY-stationary form (default on the sparse path: one column tile, only surviving rows stored).
// s0_centroid_scores_f16 keeps 64 centroid rows per wave in registers and streams the queries' B operands from LDS: 16 KB of
// ds_read_b128 per (wave, query) for 32 MFMAs -- with 16 waves per CU the LDS pipe is as busy as the matrix pipe.  Here the
// operand that is REUSED stays in registers instead: a wave holds the fp16 hi/lo images of S0Q_QT = 2 queries (128 VGPRs) for
// the whole kernel and the centroid tiles stream through LDS, 8 KB per 32-row tile shared by the 8 waves of the workgroup
// (16 queries): 8 KB of LDS reads per wave for the same 32 MFMAs, a third of the traffic.  The tiles go global -> LDS by DMA
// (one 1 KB piece per wave and tile, XOR-swizzled like stage 2's), S0Q_AHEAD 64-row blocks ahead.
// Same MFMA sequence per (row, column) as the row-stationary kernel: every value is bitwise the same.
//
// A step handles one 64-row block = two tiles (one block barrier per 64 MFMAs of a wave).
// vmcnt: a step's VMEM operations are fixed so that the wait for a block's pieces can leave the younger STORES outstanding
// (stores retire slowly and share the counter): step p issues the two DMA pieces of block p + AHEAD first, then one
// block-maximum store per query -- 2 + QT operations.  The pieces of block p were the first two operations of step
// p - AHEAD = p - 2: younger are the rest of that step (QT) and one full step (2 + QT): vmcnt(2 + 2 QT); in the first two
// steps (pieces from the prologue) 2 and 2 + QT.  (The rare dense epilogue's row and idx-word stores only add to that: the
// idx words are zeroed by the launcher, and only a tile with a surviving row stores one.)
// grid = (ceil(nqueries / 16), slices), block = 512; dynamic LDS = 8 x 4.5 KB staging + S0Q_NBUF x 16 KB blocks.
// ------------------------------------------------------------------------------------------------
#define S0Q_QT 2
#define S0Q_AHEAD 2   // in 64-row blocks
#define S0Q_NBUF 3    // block buffers of 2 x 8 KB

template <int N>
__device__ __forceinline__ void s0q_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// HI_ONLY (FLMR_NUMERICS_GPU_FP16: Q is rounded to fp16, q_lo = 0): the lo products would multiply by zero -- they and the
// hi / lo combine are left out, the values are the same (fma(0, 2^-11, x) = x)
// APPROX ("hi first", the default of the CPU-path numerics): the hi products of a tile come first; their column maxima go to
// the block maxima as they are (s0_select_cells knows they are within q_err of the full values and verifies its choice), and
// the lo products are computed only for a tile in which  ah + q_err >= thr  somewhere -- a tile that can hold a surviving row,
// a few dozen of a query's 4096 -- after which the dense epilogue tests and stores the FULL values exactly as before.
template <bool HI_ONLY, bool APPROX = false>
__global__ __launch_bounds__(512, 1) void s0_centroid_scores_qs(flmr_s0_args a, int rows_per_slice) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform for the compiler too: the queries' addresses stay in scalar registers
    const int i = lane & 31, h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * S0_LDS_STRIDE;
    char* const abuf = smem + 8 * 32 * S0_LDS_STRIDE * sizeof(float);
    const uint32_t abuf_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)abuf);
    const int row_begin = blockIdx.y * rows_per_slice;
    const int row_end = row_begin + rows_per_slice < a.K ? row_begin + rows_per_slice : a.K;  // multiples of 64
    const int ntiles = (row_end - row_begin) >> 5;
    if (ntiles <= 0) return;
    // this wave's queries (a query past the end repeats the last one: same values to the same addresses)
    int bq[S0Q_QT], nqc[S0Q_QT];
    float qe[S0Q_QT];   // APPROX: the bound of this lane's column
    f16x8 bh[S0Q_QT][8], bl[S0Q_QT][8];
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++) {
        const int b = (blockIdx.x * 8 + wave) * S0Q_QT + q;
        bq[q] = b < a.nqueries ? b : a.nqueries - 1;
        const int qlen = a.q_lens ? a.q_lens[bq[q]] : a.nq;
        nqc[q] = qlen < a.nq_cand ? qlen : a.nq_cand;
        qe[q] = APPROX ? a.q_err[(size_t)bq[q] * a.ncol + i] : 0.0f;
        const f16x8* ph = reinterpret_cast<const f16x8*>(a.q_hi + ((size_t)bq[q] * a.ncol + i) * FLMR_DIM + 64 * h);
        const f16x8* pl = reinterpret_cast<const f16x8*>(a.q_lo + ((size_t)bq[q] * a.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            bh[q][s] = ph[s];
            if constexpr (!HI_ONLY) bl[q][s] = pl[s];
        }
    }
    // every compiler-visible load lands here, before the first hand-counted operation
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++)
#pragma unroll
        for (int s = 0; s < 8; s++) {
            asm volatile("" : "+v"(bh[q][s])::"memory");
            if constexpr (!HI_ONLY) asm volatile("" : "+v"(bl[q][s])::"memory");
        }

    // tile t (rows row_begin + 32 t ...) -> buffer t % NBUF; this wave moves rows 4*wave .. 4*wave+3: piece p of row r at
    // position p ^ (r & 15).  Tiles past the end repeat the last tile (the counts above need every step's DMA).
    const int prow = 4 * wave + (lane >> 4);
    const uint32_t poff = (uint32_t)((((lane & 15) ^ (prow & 15)) << 4));
    const int nblocks = ntiles >> 1;
    auto dma_block = [&](int p) {
        const int pp = p < nblocks ? p : nblocks - 1;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t voff = (uint32_t)(row_begin + 64 * pp + 32 * u + prow) * 256u + poff;  // K * 256 < 4 GB (launcher)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(abuf_lds + ((p % S0Q_NBUF) * 2 + u) * 8192 + wave * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(dst), "s"(a.centroids_f16) : "memory", "m0");
        }
    };
#pragma unroll
    for (int p = 0; p < S0Q_AHEAD; p++) dma_block(p);

    const int c4 = (lane & 7) * 4;  // first of the 4 columns this lane stores in the dense epilogue
    // wave-uniform per query (scalar registers): the columns that count, as a ballot mask
    unsigned long long colmask[S0Q_QT];
    bool full_cols[S0Q_QT];
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++) {
        const int n = __builtin_amdgcn_readfirstlane(nqc[q]);
        full_cols[q] = n >= 32;
        colmask[q] = full_cols[q] ? ~0ull : (((1ull << n) - 1ull) * 0x100000001ull);
    }
    // maximum of a tile's 16 accumulator registers as ONE chain (v_max3): the tree form makes the compiler canonicalise each
    // raw MFMA output first (20 instructions instead of 9)
    auto tile_max = [](const float* v) {
        float m = fmaxf(v[0], v[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) m = fmaxf(fmaxf(m, v[r]), v[r + 1]);
        return m;
    };
    // the rare dense epilogue of (tile, query): FULL values of the tile's rows -> stage, surviving rows -> cs, their idx word
    auto dense = [&](int q, int rbase_, const f32x16& ah, const f32x16& al) {
        int rbase = rbase_;
        asm volatile("" : "+s"(rbase));   // (keeps this branch's address arithmetic inside the branch: it is rare)
        const int b = bq[q];
        float* cs_b = a.cs + (size_t)b * a.K * a.ncol;
        const int nvalid4 = nqc[q] - c4;
        uint32_t idxw = 0u;
#pragma unroll
        for (int r = 0; r < 16; r++) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * S0_LDS_STRIDE + i] = HI_ONLY ? ah[r] : fmaf(al[r], 1.0f / 2048.0f, ah[r]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mrow = 0; mrow < 4; mrow++) {
            const int R = (lane >> 3) + 8 * mrow;
            const float4 v4 = *reinterpret_cast<const float4*>(stage + R * S0_LDS_STRIDE + c4);
            float m4;
            if (full_cols[q]) {
                m4 = fmaxf(fmaxf(v4.x, v4.y), fmaxf(v4.z, v4.w));
            } else {
                m4 = nvalid4 > 0 ? v4.x : FLMR_NEG_INF;
                m4 = fmaxf(m4, nvalid4 > 1 ? v4.y : FLMR_NEG_INF);
                m4 = fmaxf(m4, nvalid4 > 2 ? v4.z : FLMR_NEG_INF);
                m4 = fmaxf(m4, nvalid4 > 3 ? v4.w : FLMR_NEG_INF);
            }
            unsigned long long bal = __ballot(m4 >= a.thr);  // byte j of `bal` = the 8 lanes of row 8*mrow + j
            bal |= bal >> 4; bal |= bal >> 2; bal |= bal >> 1;
            bal &= 0x0101010101010101ull;
            const uint32_t byte = (uint32_t)((bal * 0x0102040810204080ull) >> 56);
            idxw |= byte << (8 * mrow);
            if (a.cs && ((byte >> (lane >> 3)) & 1u)) *reinterpret_cast<float4*>(cs_b + (size_t)(rbase + R) * a.ncol + c4) = v4;
        }
        __builtin_amdgcn_wave_barrier();
        // (the idx words start at zero -- cleared by the launcher -- so only a tile that went through here stores one)
        if (lane == 0 && idxw != 0u) a.idx_bits[(size_t)b * a.idx_words + (rbase >> 5)] = idxw;
    };
    uint32_t aoff[8];   // this lane's eight 16-byte pieces inside a tile buffer (rows XOR-swizzled by the DMA)
#pragma unroll
    for (int s = 0; s < 8; s++) aoff[s] = (uint32_t)(i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
    for (int p = 0; p < nblocks; p++) {
        // ---- block p: this wave's pieces have landed, then everybody's ----
        if (p == 0) s0q_wait_vm<2>(); else if (p == 1) s0q_wait_vm<2 + S0Q_QT>(); else s0q_wait_vm<2 + 2 * S0Q_QT>();
        __syncthreads();
        dma_block(p + S0Q_AHEAD);  // (its buffer was read a step ago at the latest: every wave has passed this barrier since)
        float cmax[S0Q_QT];
#pragma unroll
        for (int q = 0; q < S0Q_QT; q++) cmax[q] = FLMR_NEG_INF;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            f16x8 av[8];
            {
                const char* pa = abuf + __builtin_amdgcn_readfirstlane(((p % S0Q_NBUF) * 2 + u) * 8192);   // (scalar: one add per read)
#pragma unroll
                for (int s = 0; s < 8; s++) av[s] = *reinterpret_cast<const f16x8*>(pa + aoff[s]);
            }
            const int rbase = row_begin + 64 * p + 32 * u;
            if constexpr (HI_ONLY || APPROX) {
                // one product per score: the MFMAs of all the wave's queries first, then the maxima -- the second query's
                // MFMAs run under the first one's reduction
                f32x16 ahq[S0Q_QT];
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) ahq[q][r] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 8; s++) ahq[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[q][s], ahq[q], 0, 0, 0);
                }
                float tmax[S0Q_QT];
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = ahq[q][r];
                    tmax[q] = tile_max(v);
                    cmax[q] = fmaxf(cmax[q], tmax[q]);
                }
                // (every reduction is complete -- the accumulators are dead -- before the first rare branch)
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) asm volatile("" : "+v"(tmax[q]));
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) {
                    if ((__ballot(tmax[q] + qe[q] >= a.thr) & colmask[q]) != 0ull) {  // wave-uniform and rare: some row of this tile survives (APPROX: may survive)
                        // both products of this tile, recomputed (the A fragments are still in registers; the same sequences, so
                        // the same bits): keeping the hi accumulators of all queries alive across this branch would spill
                        f32x16 ah2, al2;
#pragma unroll
                        for (int r = 0; r < 16; r++) { ah2[r] = 0.0f; al2[r] = 0.0f; }
#pragma unroll
                        for (int s = 0; s < 8; s++) {
                            ah2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[q][s], ah2, 0, 0, 0);
                            if constexpr (APPROX) al2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[q][s], al2, 0, 0, 0);
                        }
                        dense(q, rbase, ah2, al2);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) {
                    f32x16 ah, al;
#pragma unroll
                    for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[q][s], ah, 0, 0, 0);
                        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[q][s], al, 0, 0, 0);
                    }
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                    const float tmax = tile_max(v);
                    cmax[q] = fmaxf(cmax[q], tmax);
                    if ((__ballot(tmax >= a.thr) & colmask[q]) != 0ull) dense(q, rbase, ah, al);
                }
            }
            if (u == 1) {  // end of the 64-row block: its column maxima, for the cell selection
#pragma unroll
                for (int q = 0; q < S0Q_QT; q++) {
                    const float m = flmr_xhalf_max(cmax[q]);
                    if (lane < 32) a.part_val[((size_t)bq[q] * a.nblk + (rbase >> 6)) * a.ncol + i] = (i < nqc[q]) ? m : FLMR_NEG_INF;
                }
            }
        }
    }
    s0q_wait_vm<0>();  // the DMA of the repeated tiles past the end must have landed before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// S0a, query-stationary, "hi first" (APPROX) and fp16-numerics (HI_ONLY) paths: the main loop has NO rare branch.
// s0_centroid_scores_qs<.., APPROX> above tests every (tile, query) and runs the dense epilogue -- lo products, staging, row
// stores -- inline; that keeps the queries' lo images (64 VGPRs) resident for a branch taken by < 1 % of the tiles, leaves
// one register set for the A fragments, and -- the workgroup's eight waves being phase-locked by the block barrier -- makes
// every tile an LDS phase (all waves fetch their fragments: 64 KB at 128 B/clk) FOLLOWED by an MFMA phase: the matrix pipe
// measured 40 % busy.  Here a flagged (tile, query) only appends its index to a per-wave list; the main loop holds the hi
// images and TWO fragment sets -- every tile's fragments are requested while the MFMAs of the tile before it run (the first
// tile of the next step during this one: the barrier of step P certifies super-block P + 1) -- and nothing else; it also
// leaves one column maximum per S0Q2_GRP blocks for the cell selection.  After the loop every wave works through its own list without any block
// barrier: the queries' lo images are loaded then (the fragment registers are free), the tile's rows come straight from the
// table into fragment layout, both products are recomputed with the same MFMA sequences -- the same bits -- and the dense
// epilogue is the one above.  Block maxima (hi-only values, as s0_select_cells expects on this path) and idx words /
// surviving rows are identical to the inline form.
// grid = (ceil(nqueries / (2 S0Q2_WAVES)), slices <= 8192 rows each), block = 64 S0Q2_WAVES;
// dynamic LDS = S0Q2_NBUF x S0Q2_SBT x 8 KB tile buffers (reused as 8 x 4.5 KB staging rows by the deferred pass) + S0Q2_WAVES x
// S0Q2_FCAP flagged-tile entries (u16).
// ------------------------------------------------------------------------------------------------
#define S0Q2_MAX_SLICE_ROWS 8192
#ifndef S0Q2_WAVES
#define S0Q2_WAVES 8    // (12 -- three waves per SIMD, 24 queries per workgroup -- measured slower: the waves that do not move tiles wait for the movers)
#endif
#ifdef S0Q_PROFILE   // development only: s_memtime clocks of wave 0 of every workgroup per phase, printed by the launcher
__device__ unsigned long long s0q_prof[8];
#define S0Q_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define S0Q_STAMP(k) do { } while (0)
#endif
#define S0Q2_FCAP (S0Q2_MAX_SLICE_ROWS / 32 * S0Q_QT)   // every (tile, query) of a slice could be flagged
#ifndef S0Q2_SBT
#define S0Q2_SBT 4     // tiles per step ("super-block" of 128 rows): one workgroup barrier per 4 x 16 MFMAs of a wave (2: measured the same)
#endif
#define S0Q2_AHEAD 3   // super-blocks requested ahead: at the barrier of step P super-block P + 1 has landed (its first tile is fetched one step early)
#ifdef S0_NO_GROUPS   // development A/B: no coarse level
#define S0Q2_GROUPS false
#else
#define S0Q2_GROUPS true
#endif
#define S0Q2_GRP 8     // 64-row blocks per group of the coarse column maxima (512 rows; a multiple of the blocks of a step)
#define S0Q2_NBUF 4    // super-block buffers of S0Q2_SBT x 8 KB (the staging rows of the deferred pass reuse them)

template <bool HI_ONLY>
__global__ __launch_bounds__(64 * S0Q2_WAVES, 1) void s0_centroid_scores_qs2(flmr_s0_args a, int rows_per_slice) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * S0_LDS_STRIDE;   // deferred pass only: aliases the tile buffers
    char* const abuf = smem;
    uint16_t* const flist = reinterpret_cast<uint16_t*>(abuf + S0Q2_NBUF * S0Q2_SBT * 8192) + wave * S0Q2_FCAP;
    const uint32_t abuf_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)abuf);
    const int row_begin = blockIdx.y * rows_per_slice;
    const int row_end = row_begin + rows_per_slice < a.K ? row_begin + rows_per_slice : a.K;  // multiples of 64
    const int ntiles = (row_end - row_begin) >> 5;
    if (ntiles <= 0) return;
#ifdef S0Q_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
    int bq[S0Q_QT], nqc[S0Q_QT];
    float qe[S0Q_QT];
    f16x8 bh[S0Q_QT][8];
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++) {
        const int b = (blockIdx.x * S0Q2_WAVES + wave) * S0Q_QT + q;
        bq[q] = b < a.nqueries ? b : a.nqueries - 1;
        const int qlen = a.q_lens ? a.q_lens[bq[q]] : a.nq;
        nqc[q] = qlen < a.nq_cand ? qlen : a.nq_cand;
        qe[q] = HI_ONLY ? 0.0f : a.q_err[(size_t)bq[q] * a.ncol + i];
        const f16x8* ph = reinterpret_cast<const f16x8*>(a.q_hi + ((size_t)bq[q] * a.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) bh[q][s] = ph[s];
    }
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++)
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(bh[q][s])::"memory");

    const int prow = 4 * wave + (lane >> 4);
    const uint32_t poff = (uint32_t)((((lane & 15) ^ (prow & 15)) << 4));
    const int nblocks = ntiles / S0Q2_SBT;   // super-blocks (the launcher cuts slices in multiples of 32 * S0Q2_SBT rows)
    const bool mover = wave < 8;   // (wave-uniform)
    auto dma_block = [&](int p) {
        if (!mover) return;
        const int pp = p < nblocks ? p : nblocks - 1;
#pragma unroll
        for (int u = 0; u < S0Q2_SBT; u++) {
            const uint32_t voff = (uint32_t)(row_begin + 32 * S0Q2_SBT * pp + 32 * u + prow) * 256u + poff;  // K * 256 < 4 GB (launcher)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(abuf_lds + ((p % S0Q2_NBUF) * S0Q2_SBT + u) * 8192 + wave * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(dst), "s"(a.centroids_f16) : "memory", "m0");
        }
    };
#pragma unroll
    for (int p = 0; p < S0Q2_AHEAD; p++) dma_block(p);

    unsigned long long colmask[S0Q_QT];
    bool full_cols[S0Q_QT];
    float qemax[S0Q_QT];   // the largest column bound of the query (wave-uniform): a row maximum is within it of the full one
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++) {
        const int n = __builtin_amdgcn_readfirstlane(nqc[q]);
        full_cols[q] = n >= 32;
        colmask[q] = full_cols[q] ? ~0ull : (((1ull << n) - 1ull) * 0x100000001ull);
        qemax[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(flmr_half_wave_max(i < n ? qe[q] : 0.0f))));
    }
    auto tile_max = [](const f32x16& v) {   // one chain (v_max3): a tree makes the compiler canonicalise every raw MFMA output
        float m = fmaxf(v[0], v[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) m = fmaxf(fmaxf(m, v[r]), v[r + 1]);
        return m;
    };
    uint32_t aoff[8];
#pragma unroll
    for (int s = 0; s < 8; s++) aoff[s] = (uint32_t)(i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
    int nflag = 0;   // wave-uniform
    S0Q_STAMP(0);

    // Software pipeline over the tiles: the fragments of a tile are fetched while the MFMAs of the tile before it run (the first
    // tile of super-block P + 1 during step P) -- otherwise the eight waves, phase-locked by the barrier, all fetch (64 KB at
    // 128 B/clk) and then all multiply.  For that the barrier of step P certifies super-block P + 1.
    // VMEM order of a moving wave: prologue D0 D1 D2 (S0Q2_SBT operations each); step P: D(P+3), then one block-maximum store per
    // query and 64-row block.  Own pieces of super-block P + 1 (the first operations of step P - 2): younger are that step's
    // stores and one full step.
    constexpr int ST = S0Q2_SBT / 2 * S0Q_QT;   // stores of a step
    f16x8 avA[8], avB[8];
    if (mover) s0q_wait_vm<S0Q2_SBT * (S0Q2_AHEAD - 1)>();   // D0
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; s++) avA[s] = *reinterpret_cast<const f16x8*>(abuf + aoff[s]);
    auto tile = [&](const f16x8 (&av)[8], int t, float (&cmax)[S0Q_QT]) {   // MFMAs + maxima + flags of tile t (of the slice)
        f32x16 acc[S0Q_QT];
#pragma unroll
        for (int q = 0; q < S0Q_QT; q++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 8; s++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[q][s], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < S0Q_QT; q++) {
            const float tmax = tile_max(acc[q]);
            cmax[q] = fmaxf(cmax[q], tmax);
            // wave-uniform and rare on a corpus whose queries meet few centroids (1 % of the tiles at the policy thresholds of the
            // synthetic corpus; MOST tiles once thousands of centroids pass the threshold): some row of this tile survives (or,
            // with the lo products still missing, may survive).  The tile's idx word -- one bit per row: does the row's maximum
            // over the query's columns reach the threshold -- is decided HERE from the hi products whenever no row's maximum lies
            // within the bound of the threshold: a transpose-reduce leaves row (i >> 1)'s maximum over the 32 columns in lane i
            // (16 cross-lane exchanges for the 16 x 32 values of a half-wave, not 16 x 5), a ballot collects the bits.  Only a
            // tile with an ambiguous row goes on the list of the deferred pass (both products, dense epilogue).
            if ((__ballot(tmax + qe[q] >= a.thr) & colmask[q]) != 0ull) {
#ifdef S0Q2_NO_INLINE_IDX   // development A/B: every flagged tile goes to the deferred pass (the form before round 4)
                if (lane == 0) flist[nflag] = (uint16_t)(t * S0Q_QT + q);
                nflag++;
                continue;
#endif
                // (the exchanges as VALU operations where the instruction set has one -- v_permlane16_swap across lane bit 4, DPP
                // row_ror:8 across bit 3, row_shl:4 / row_shr:4 by bank across bit 2, quad_perm across bits 1 and 0: the same lanes, the same
                // values as the __shfl_xor forms (profiles/microbench/xor_exchange_check.hip).  MOST tiles come through here
                // once thousands of centroids pass the threshold, and 16 crossbar round trips per tile were half of this kernel's time there)
                float v8[8];
                {
                    float x0[8], x1[8];   // lanes with bit 4 clear keep registers 0..7, the others 8..15
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        x0[r] = acc[q][r]; x1[r] = acc[q][r + 8];
                        if (!full_cols[q]) { x0[r] = i < nqc[q] ? x0[r] : FLMR_NEG_INF; x1[r] = i < nqc[q] ? x1[r] : FLMR_NEG_INF; }
                    }
                    flmr_x16_max8(x0, x1, v8);
                }
                float v4[4], v2[2];
                const bool up3 = (i & 8) != 0, up2 = (i & 4) != 0, up1 = (i & 2) != 0;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    v4[r] = flmr_fmax_raw(up3 ? v8[r + 4] : v8[r], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(up3 ? v8[r] : v8[r + 4]), 0x128 /* row_ror:8 */, 0xF, 0xF, false)));
#pragma unroll
                for (int r = 0; r < 2; r++) v2[r] = flmr_fmax_raw(up2 ? v4[r + 2] : v4[r], flmr_dpp_xor4(up2 ? v4[r] : v4[r + 2]));
                float rm = flmr_fmax_raw(up1 ? v2[1] : v2[0], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(up1 ? v2[0] : v2[1]), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false)));
                rm = flmr_fmax_raw(rm, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(rm), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false)));   // lane (i, h): the maximum of row r = i >> 1 (of this half's 16) over all columns
                const unsigned long long amb = HI_ONLY ? 0ull : __ballot(rm + qemax[q] >= a.thr && rm - qemax[q] < a.thr);
                if (amb != 0ull) {
                    if (lane == 0) flist[nflag] = (uint16_t)(t * S0Q_QT + q);
                    nflag++;
                } else {
                    // register r = 4 a + b of half h is row b + 8 a + 4 h: lane 32 h + 8 a + 2 b (and its odd twin) -> bit 8 a + 4 h + b
                    unsigned long long y = __ballot(rm >= a.thr) & 0x5555555555555555ull;
                    y = (y | (y >> 1)) & 0x3333333333333333ull;
                    y = (y | (y >> 2)) & 0x0f0f0f0f0f0f0f0full;
                    const uint32_t word = (uint32_t)y | ((uint32_t)(y >> 32) << 4);
                    if (lane == 0 && word != 0u) a.idx_bits[(size_t)bq[q] * a.idx_words + ((row_begin >> 5) + t)] = word;
                }
            }
        }
    };
    auto fetch = [&](f16x8 (&av)[8], int buf_tile) {   // buf_tile = (super-block % NBUF) * SBT + tile
        const char* const pt_ = abuf + __builtin_amdgcn_readfirstlane(buf_tile * 8192);
#pragma unroll
        for (int s = 0; s < 8; s++) av[s] = *reinterpret_cast<const f16x8*>(pt_ + aoff[s]);
    };
    // the coarse level: one column maximum per S0Q2_GRP blocks (s0_select_cells reads these 32 KB per query first, and block maxima
    // of the best groups only), when the launcher cut the slices on group boundaries
    const bool groups = a.grp_blocks == S0Q2_GRP;
    float gmax[S0Q_QT];
#pragma unroll
    for (int q = 0; q < S0Q_QT; q++) gmax[q] = FLMR_NEG_INF;
    for (int p = 0; p < nblocks; p++) {
        if (mover) {
            if (p == 0) s0q_wait_vm<S0Q2_SBT>(); else if (p == 1) s0q_wait_vm<S0Q2_SBT + ST>(); else s0q_wait_vm<S0Q2_SBT + 2 * ST>();
        }
        S0Q_STAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's reads of the buffer about to be refilled are complete)
        __syncthreads();
        S0Q_STAMP(2);
        dma_block(p + S0Q2_AHEAD);
        const int bt = (p % S0Q2_NBUF) * S0Q2_SBT, bn = ((p + 1) % S0Q2_NBUF) * S0Q2_SBT;
#pragma unroll
        for (int u2 = 0; u2 < S0Q2_SBT / 2; u2++) {   // one 64-row block: tiles 2 u2 (in A) and 2 u2 + 1 (B)
            fetch(avB, bt + 2 * u2 + 1);
            float cmax[S0Q_QT];
#pragma unroll
            for (int q = 0; q < S0Q_QT; q++) cmax[q] = FLMR_NEG_INF;
            tile(avA, S0Q2_SBT * p + 2 * u2, cmax);
            fetch(avA, 2 * u2 + 2 < S0Q2_SBT ? bt + 2 * u2 + 2 : bn);   // (the last one: first tile of the next super-block)
            tile(avB, S0Q2_SBT * p + 2 * u2 + 1, cmax);
#pragma unroll
            for (int q = 0; q < S0Q_QT; q++) {   // the 64-row block's column maxima, for the cell selection
                const float m = flmr_xhalf_max(cmax[q]);
                gmax[q] = fmaxf(gmax[q], m);
                if (lane < 32)
                    a.part_val[((size_t)bq[q] * a.nblk + ((row_begin + 32 * S0Q2_SBT * p + 64 * u2) >> 6)) * a.ncol + i] = (i < nqc[q]) ? m : FLMR_NEG_INF;
            }
        }
        if (groups && (p + 1) % (S0Q2_GRP * 2 / S0Q2_SBT) == 0) {   // (wave-uniform) a group of S0Q2_GRP blocks is complete
            const int g = (row_begin + 32 * S0Q2_SBT * p) / (64 * S0Q2_GRP);
            float* const grp = reinterpret_cast<float*>(a.part_idx);
#pragma unroll
            for (int q = 0; q < S0Q_QT; q++) {
                if (lane < 32) grp[((size_t)bq[q] * (a.nblk / S0Q2_GRP) + g) * a.ncol + i] = (i < nqc[q]) ? gmax[q] : FLMR_NEG_INF;
                gmax[q] = FLMR_NEG_INF;
            }
        }
        S0Q_STAMP(3);
    }
    s0q_wait_vm<0>();  // (the DMA of the repeated tiles past the end has landed; nothing hand-counted is outstanding below)
    asm volatile("" ::: "memory");
    __syncthreads();   // the tile buffers become the waves' staging rows
    S0Q_STAMP(4);
#ifdef S0Q_PROFILE
    auto prof_out = [&]() {
        if (threadIdx.x == 0) {
            for (int k = 0; k < 6; k++) atomicAdd(&s0q_prof[k], (unsigned long long)pt[k]);
            atomicAdd(&s0q_prof[6], (unsigned long long)nblocks);
            atomicAdd(&s0q_prof[7], 1ull);
        }
    };
    if (nflag == 0) { prof_out(); return; }
#else
    if (nflag == 0) return;
#endif

    // ---- the flagged tiles: both products, dense epilogue (per wave, no block barrier) ----
    // Latency-bound (a few entries per wave): every query's lo image is requested up front, and the rows of entry e + 1 while
    // entry e is worked on.
    const int c4 = (lane & 7) * 4;
    f16x8 bl[S0Q_QT][8];
    if constexpr (!HI_ONLY) {
#pragma unroll
        for (int q = 0; q < S0Q_QT; q++) {
            const f16x8* pl = reinterpret_cast<const f16x8*>(a.q_lo + ((size_t)bq[q] * a.ncol + i) * FLMR_DIM + 64 * h);
#pragma unroll
            for (int s = 0; s < 8; s++) bl[q][s] = pl[s];
        }
    }
    auto rows_of = [&](int e, f16x8 (&av)[8]) {   // fragment layout straight from the table
        const int ent = __builtin_amdgcn_readfirstlane((int)flist[e < nflag ? e : nflag - 1]);
        const f16x8* pr = reinterpret_cast<const f16x8*>(a.centroids_f16 + (size_t)(row_begin + 32 * (ent / S0Q_QT) + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) av[s] = pr[s];
    };
    f16x8 av[8], avn[8];
    rows_of(0, av);
    for (int e = 0; e < nflag; e++) {
        const int ent = __builtin_amdgcn_readfirstlane((int)flist[e]);
        const int rbase = row_begin + 32 * (ent / S0Q_QT);
        rows_of(e + 1, avn);   // (past the end: the last entry again)
#pragma unroll
        for (int q = 0; q < S0Q_QT; q++) {
            if (ent % S0Q_QT != q) continue;   // wave-uniform
            const int b = bq[q];
            float* const cs_b = a.cs + (size_t)b * a.K * a.ncol;
            const int nvalid4 = nqc[q] - c4;
            f32x16 ah, al;
#pragma unroll
            for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bh[q][s], ah, 0, 0, 0);
                if constexpr (!HI_ONLY) al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bl[q][s], al, 0, 0, 0);
            }
            uint32_t idxw = 0u;
#pragma unroll
            for (int r = 0; r < 16; r++) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * S0_LDS_STRIDE + i] = HI_ONLY ? ah[r] : fmaf(al[r], 1.0f / 2048.0f, ah[r]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mrow = 0; mrow < 4; mrow++) {
                const int R = (lane >> 3) + 8 * mrow;
                const float4 v4 = *reinterpret_cast<const float4*>(stage + R * S0_LDS_STRIDE + c4);
                float m4;
                if (full_cols[q]) {
                    m4 = fmaxf(fmaxf(v4.x, v4.y), fmaxf(v4.z, v4.w));
                } else {
                    m4 = nvalid4 > 0 ? v4.x : FLMR_NEG_INF;
                    m4 = fmaxf(m4, nvalid4 > 1 ? v4.y : FLMR_NEG_INF);
                    m4 = fmaxf(m4, nvalid4 > 2 ? v4.z : FLMR_NEG_INF);
                    m4 = fmaxf(m4, nvalid4 > 3 ? v4.w : FLMR_NEG_INF);
                }
                unsigned long long bal = __ballot(m4 >= a.thr);  // byte j of `bal` = the 8 lanes of row 8*mrow + j
                bal |= bal >> 4; bal |= bal >> 2; bal |= bal >> 1;
                bal &= 0x0101010101010101ull;
                const uint32_t byte = (uint32_t)((bal * 0x0102040810204080ull) >> 56);
                idxw |= byte << (8 * mrow);
                if (a.cs && ((byte >> (lane >> 3)) & 1u)) *reinterpret_cast<float4*>(cs_b + (size_t)(rbase + R) * a.ncol + c4) = v4;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the staged tile has been read before the next one overwrites it)
            if (lane == 0 && idxw != 0u) a.idx_bits[(size_t)b * a.idx_words + (rbase >> 5)] = idxw;
        }
#pragma unroll
        for (int s = 0; s < 8; s++) av[s] = avn[s];
    }
#ifdef S0Q_PROFILE
    S0Q_STAMP(5);
    prof_out();
#endif
}

__global__ void check_f16_exact_kernel(const float* x, size_t n, int* flag) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float v = x[e];
        if ((float)(_Float16)v != v) atomicAnd(flag, 0);
    }
}

__global__ void convert_f16_kernel(const float* x, size_t n, _Float16* out) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) out[e] = (_Float16)x[e];
}

// max over the rows of ||row||_2 (fp32, non-negative: the float's bit pattern orders like an int), rounded up
__global__ __launch_bounds__(256) void max_row_norm_kernel(const float* __restrict__ x, int64_t rows, int* out_bits) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float n = 0.0f;
    if (r < rows) {
        float ss = 0.0f;
        const float4* p = reinterpret_cast<const float4*>(x + (size_t)r * FLMR_DIM);
#pragma unroll 8
        for (int j = 0; j < FLMR_DIM / 4; j++) {
            const float4 v = p[j];
            ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
        n = sqrtf(ss);
    }
    if (n == n) atomicMax(out_bits, __float_as_int(n));
}

int flmr_max_row_norm(const float* dev, int64_t rows, float* host_result) {
    int* bits = nullptr;
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(&bits), sizeof(int)));
    FLMR_HIP(hipMemset(bits, 0, sizeof(int)));
    hipLaunchKernelGGL(max_row_norm_kernel, dim3((unsigned)flmr_ceil_div(rows > 0 ? rows : 1, 256)), dim3(256), 0, 0, dev, rows, bits);
    int b = 0;
    FLMR_HIP(hipMemcpy(&b, bits, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(bits);
    float f;
    memcpy(&f, &b, 4);
    *host_result = f * 1.0001f + 1e-30f;   // (the fp32 sum of squares is within 128 * 2^-24 of the true one)
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// "hi first" stage 0: per (query, column) a RIGOROUS bound on how far the hi-only score  ah = sum_j c_j qh_j  (what the MFMA
// accumulates in fp32) can be from the value the full sequence stores,  s = fl(ah + al * 2^-11),  al = fl(sum_j c_j ql_j):
//     |s - ah| <= |al| / 2048 + ulp/2(s),   |al| <= (1 + 128 * 2^-24) * ||c|| ||ql||   (Cauchy-Schwarz on the exact products),
//     ulp/2(s) <= 2^-24 |s| <= 2^-24 ||c|| ||q||.
// err = cen_norm_max * (||ql|| / 2048 * 1.001 + ||q|| * 2.5e-7), norms accumulated in fp32 and rounded up by the 1.001.
// With it the kernels decide where the lo products are needed at all (a tile can hold a surviving row only if ah + err >= thr)
// and s0_select_cells verifies its block choice; every stored value and every decision is still the full sequence's.
// grid = nqueries, block = 64 (lane = column, two columns per lane for ncol = 64 .. 128)
// ------------------------------------------------------------------------------------------------
// q_err_sum[query] bounds the difference between a passage's stage-2 score from hi-only column maxima and from the full ones:
// per column |max_t s - max_t ah| <= err, plus what two k-ascending fp32 sums of <= 128 terms of magnitude <= ||c|| ||q|| can differ
// by through rounding (2 * 127 * 2^-24 * sum |terms|).
// (computed by s0_prepare_kernel below, together with the query images)

// ------------------------------------------------------------------------------------------------
// Everything the query-stationary stage-0 kernels need from a query, in ONE launch (one column tile: ncol == 32): the fp16
// hi / lo images (s0_split_q), the per-column bounds of "hi first" and their per-query sum for stage 2 (the expressions above; the 128 squares of a column are summed by eight threads and a shuffle tree instead of one thread in dimension
// order -- the bound's 1e-3 relative slack covers either order's rounding by two orders of magnitude), and the query's idx
// words cleared (the kernels store a word only for a tile with a surviving row).  grid = nqueries, block = 256: thread t owns
// dimensions 16 (t % 8) .. + 15 of column t / 8.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s0_prepare_kernel(const float* __restrict__ Q, const int32_t* __restrict__ q_lens, int nq, int nq_cand,
                                                         _Float16* __restrict__ q_hi, _Float16* __restrict__ q_lo, int hi_only,
                                                         float cen_norm_max, float* __restrict__ q_err, float* __restrict__ q_err_sum,
                                                         uint32_t* __restrict__ idx_bits, int idx_words) {
    __shared__ float s_e[32], s_q[32];
    const int b = blockIdx.x, t = threadIdx.x;
    const int qlen = q_lens ? q_lens[b] : nq;
    const int nqc = qlen < nq_cand ? qlen : nq_cand;
    const int col = t >> 3, d0 = 16 * (t & 7);
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = 0.0f;
    if (col < nqc) {
        const float4* src = reinterpret_cast<const float4*>(Q + ((size_t)b * nq + col) * FLMR_DIM + d0);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float4 x = src[j]; v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w; }
    }
    f16x8 hv[2], lv[2];
    float sl = 0.0f, sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const _Float16 hi = (_Float16)v[j];
        const _Float16 lo = hi_only ? (_Float16)0.0f : (_Float16)((v[j] - (float)hi) * 2048.0f);
        hv[j >> 3][j & 7] = hi;
        lv[j >> 3][j & 7] = lo;
        const float l = (float)lo, x = fmaf(l, 1.0f / 2048.0f, (float)hi);
        sl = fmaf(l, l, sl);
        sq = fmaf(x, x, sq);
    }
    f16x8* ph = reinterpret_cast<f16x8*>(q_hi + ((size_t)b * 32 + col) * FLMR_DIM + d0);
    f16x8* pl = reinterpret_cast<f16x8*>(q_lo + ((size_t)b * 32 + col) * FLMR_DIM + d0);
    ph[0] = hv[0]; ph[1] = hv[1];
    pl[0] = lv[0]; pl[1] = lv[1];
    for (int w = t; w < idx_words; w += 256) idx_bits[(size_t)b * idx_words + w] = 0u;
    if (q_err) {   // (block-uniform)
#pragma unroll
        for (int m = 4; m >= 1; m >>= 1) { sl += __shfl_xor(sl, m, 64); sq += __shfl_xor(sq, m, 64); }
        const float e = cen_norm_max * (sqrtf(sl) * (1.001f / 2048.0f) + sqrtf(sq) * 2.5e-7f);
        if ((t & 7) == 0) {
            q_err[(size_t)b * 32 + col] = e;
            s_e[col] = col < nqc ? e : 0.0f;
            s_q[col] = col < nqc ? sqrtf(sq) : 0.0f;
        }
        __syncthreads();
        if (t < 32) {
            float esum = s_e[t], qmax = s_q[t];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) { esum += __shfl_xor(esum, m, 64); qmax = fmaxf(qmax, __shfl_xor(qmax, m, 64)); }
            if (t == 0 && q_err_sum) q_err_sum[b] = 1.01f * esum + 1.6e-5f * (float)nqc * cen_norm_max * qmax * 1.001f;
        }
    }
}

int flmr_convert_f16(const float* dev, size_t n, _Float16* out) {
    hipLaunchKernelGGL(convert_f16_kernel, dim3(1024), dim3(256), 0, 0, dev, n, out);
    FLMR_HIP(hipDeviceSynchronize());
    return FLMR_OK;
}

int flmr_check_f16_exact(const float* dev, size_t n, int32_t* host_result) {
    int* flag = nullptr;
    FLMR_HIP(hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)));
    int one = 1;
    FLMR_HIP(hipMemcpy(flag, &one, sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_f16_exact_kernel, dim3(1024), dim3(256), 0, 0, dev, n, flag);
    FLMR_HIP(hipMemcpy(&one, flag, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(flag);
    *host_result = one;
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// S0a (VALU cross-check path, FLMR_S0_IMPL=valu): plain k-ascending fp32 dot products, then a
// post-processing kernel derives the idx bits and the block partials from the stored table.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s0_centroid_scores_valu(flmr_s0_args a) {
    const int b = blockIdx.y;
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
    const size_t total = (size_t)a.K * a.ncol;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / a.ncol), col = (int)(e % a.ncol);
        float acc = 0.0f;
        if (col < nqc) {
            const float* c = a.centroids + (size_t)row * FLMR_DIM;
            const float* q = a.Q + ((size_t)b * a.nq + col) * FLMR_DIM;
            for (int k = 0; k < FLMR_DIM; k++) acc = fmaf(c[k], q[k], acc);
        }
        a.cs[(size_t)b * total + e] = acc;
    }
}

template <int NC>
__global__ __launch_bounds__(256) void s0_postprocess_table(flmr_s0_args a) {
    __shared__ float lds_v[8 * 32 * NC];
    __shared__ int lds_i[8 * 32 * NC];
    __shared__ uint32_t lds_bits[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
    const int row0 = blockIdx.x * 128;
    const float* cs_b = a.cs + (size_t)b * a.K * a.ncol;
    // idx bits: thread t < 128 owns row row0+t
    if (tid < 128) {
        const int row = row0 + tid;
        float m = FLMR_NEG_INF;
        if (row < a.K)
            for (int c = 0; c < nqc; c++) m = fmaxf(m, cs_b[(size_t)row * a.ncol + c]);
        const bool flag = (row < a.K) && (m >= a.thr);
        unsigned long long bal = __ballot(flag);
        if ((tid & 63) == 0) { lds_bits[(tid >> 6) * 2] = (uint32_t)bal; lds_bits[(tid >> 6) * 2 + 1] = (uint32_t)(bal >> 32); }
    }
    __syncthreads();
    if (tid < 4 && row0 + tid * 32 < a.K) a.idx_bits[(size_t)b * a.idx_words + (row0 >> 5) + tid] = lds_bits[tid];
    // partial top lists: thread (g = tid>>5, c = tid&31) scans rows g, g+8, ...
    const int g = tid >> 5, c = tid & 31;
    for (int ct = 0; ct < (a.ncol >> 5); ct++) {
        const int col = ct * 32 + c;
        flmr_toplist<NC> tl;
        tl.init();
        if (col < nqc)
            for (int r = g; r < 128; r += 8) {
                const int row = row0 + r;
                if (row < a.K) tl.insert(cs_b[(size_t)row * a.ncol + col], row);
            }
#pragma unroll
        for (int t = 0; t < NC; t++) { lds_v[(g * 32 + c) * NC + t] = tl.v[t]; lds_i[(g * 32 + c) * NC + t] = tl.id[t]; }
        __syncthreads();
        if (g == 0) {
            for (int w = 1; w < 8; w++)
#pragma unroll
                for (int t = 0; t < NC; t++) tl.insert(lds_v[(w * 32 + c) * NC + t], lds_i[(w * 32 + c) * NC + t]);
            const size_t base = (((size_t)b * a.nblk + blockIdx.x) * a.ncol + col) * NC;
#pragma unroll
            for (int t = 0; t < NC; t++) { a.part_val[base + t] = tl.v[t]; a.part_idx[base + t] = tl.id[t]; }
        }
        __syncthreads();
    }
}

enum { S0_F16 = 0, S0_F32 = 1, S0_VALU = 2 };

template <int NC>
static int launch_s0_t(flmr_s0_args& a, hipStream_t st, int impl) {
    const int qsplit = a.nqueries < 8 ? a.nqueries : 8;
    a.nblk = (int)flmr_ceil_div(a.K, impl == S0_F16 ? 32 * S0_RT : 128);
    a.grp_blocks = 0;
    a.part_rows = impl == S0_F16 ? 32 * S0_RT : 0;
    if (impl == S0_F16) {
        const bool sparse_ = !a.full_table && a.ncol == 32 && !flmr_opts().has(FLMR_OPT_S0_STAGED);
        const bool qs_ = sparse_ && a.centroids_f16 && (int64_t)a.K * 256 < (1ll << 32) && !flmr_opts().is(FLMR_OPT_S0_IMPL, "f16rs");
        if (qs_) {   // images, bounds and cleared idx words in one launch
            float* const eb = (a.q_err_buf && !a.q_hi_only) ? a.q_err_buf : nullptr;
            hipLaunchKernelGGL(s0_prepare_kernel, dim3(a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq, a.nq_cand, a.q_hi, a.q_lo,
                               a.q_hi_only, a.cen_norm_max, eb, a.q_err_sum, a.idx_bits, a.idx_words);
        } else {
            hipLaunchKernelGGL(s0_split_q, dim3((a.ncol * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens,
                               a.nq, a.nq_cand, a.ncol, a.q_hi, a.q_lo, a.q_hi_only);
        }
        const size_t lds = (size_t)S0_WAVES * 32 * S0_LDS_STRIDE * sizeof(float) + (S0_DMA_B ? (size_t)2 * S0_DCH * 2 * 8192 : (size_t)S0_CH * 2 * 32 * S0_BROW * sizeof(_Float16));
        const bool sparse = !a.full_table && a.ncol == 32 && !flmr_opts().has(FLMR_OPT_S0_STAGED);
        const dim3 grid((a.nblk + S0_WAVES - 1) / S0_WAVES, qsplit), block(64 * S0_WAVES);
        const bool qs = sparse && a.centroids_f16 && (int64_t)a.K * 256 < (1ll << 32) && !flmr_opts().is(FLMR_OPT_S0_IMPL, "f16rs");
        if (!qs) { a.q_err = nullptr; a.q_err_buf = nullptr; }   // the block maxima of the other kernels are the full values: s0_select_cells must not assume otherwise
        if (qs) {
            // query-stationary: 16 queries per workgroup, the table cut into as many slices (multiples of 64 rows) as fill the chip
            const int ngroups = (int)flmr_ceil_div(a.nqueries, 8 * S0Q_QT);
            int slices = (int)flmr_ceil_div(512, ngroups);  // (256 .. 768 workgroups measure the same)
            const int max_slices = a.K / 64;
            if (slices > max_slices) slices = max_slices;
            const int min_slices = (int)flmr_ceil_div(a.K, S0Q2_MAX_SLICE_ROWS);   // (the flagged-tile lists hold a whole slice)
            if (slices < min_slices) slices = min_slices;
            if (slices < 1) slices = 1;
            const int rows_per_slice = (int)flmr_round_up(flmr_ceil_div(a.K, slices), 64);
            slices = (int)flmr_ceil_div(a.K, rows_per_slice);
            const size_t ldsq = (size_t)8 * 32 * S0_LDS_STRIDE * sizeof(float) + (size_t)S0Q_NBUF * 2 * 8192;
            // (the idx words were cleared and the bounds of this batch's queries written by s0_prepare_kernel above)
            const bool inline_dense = flmr_opts().is(FLMR_OPT_S0_IMPL, "qs1");   // A/B: the form with the dense epilogue in the loop
            if ((a.q_hi_only || a.q_err) && !inline_dense && a.K % (32 * S0Q2_SBT) == 0) {
                const size_t ldsq2 = (size_t)S0Q2_NBUF * S0Q2_SBT * 8192 + (size_t)S0Q2_WAVES * S0Q2_FCAP * sizeof(uint16_t);
                const int ngroups2 = (int)flmr_ceil_div(a.nqueries, S0Q2_WAVES * S0Q_QT);
                int slices2 = 512 / ngroups2;   // (rounded down: a few workgroups over 2 x 256 would cost a third round)
                if (slices2 > a.K / (32 * S0Q2_SBT)) slices2 = a.K / (32 * S0Q2_SBT);
                if (slices2 < min_slices) slices2 = min_slices;
                if (slices2 < 1) slices2 = 1;
                // slices on group boundaries when the table allows it: the kernel then also leaves the coarse column maxima
                const bool grp = a.K % (64 * S0Q2_GRP) == 0 && a.part_idx != nullptr && a.nblk == a.K / 64 && S0Q2_GROUPS;
                const int slice_unit = grp ? 64 * S0Q2_GRP : 32 * S0Q2_SBT;
                if (grp && slices2 > a.K / slice_unit) slices2 = a.K / slice_unit;
                const int rows_per_slice2 = (int)flmr_round_up(flmr_ceil_div(a.K, slices2), slice_unit);
                slices2 = (int)flmr_ceil_div(a.K, rows_per_slice2);
                a.grp_blocks = grp ? S0Q2_GRP : 0;
                const int slices = slices2, rows_per_slice = rows_per_slice2;   // (shadow: this branch's own cut)
                if (a.q_hi_only) {
                    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_qs2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq2));
                    hipLaunchKernelGGL(s0_centroid_scores_qs2<true>, dim3(ngroups2, slices), dim3(64 * S0Q2_WAVES), ldsq2, st, a, rows_per_slice);
                } else {
                    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_qs2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq2));
                    hipLaunchKernelGGL(s0_centroid_scores_qs2<false>, dim3(ngroups2, slices), dim3(64 * S0Q2_WAVES), ldsq2, st, a, rows_per_slice);
                }
#ifdef S0Q_PROFILE
                {
                    unsigned long long hh[8];
                    (void)hipDeviceSynchronize();
                    (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(s0q_prof), sizeof(hh));
                    fprintf(stderr, "[s0q] workgroups %llu x %d slices rows %d; s_memtime ticks per workgroup: prologue %.0f | per block: dma wait %.1f barrier %.1f body %.1f | "
                                    "drain %.0f deferred %.0f (blocks per workgroup %.1f)\n", hh[7], slices, rows_per_slice, (double)hh[0] / hh[7],
                            (double)hh[1] / hh[6], (double)hh[2] / hh[6], (double)hh[3] / hh[6], (double)hh[4] / hh[7], (double)hh[5] / hh[7], (double)hh[6] / hh[7]);
                    unsigned long long z[8] = {};
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(s0q_prof), z, sizeof(z));
                }
#endif
            } else if (a.q_hi_only) {
                FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_qs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
                hipLaunchKernelGGL(s0_centroid_scores_qs<true>, dim3(ngroups, slices), dim3(512), ldsq, st, a, rows_per_slice);
            } else if (a.q_err) {
                FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_qs<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
                hipLaunchKernelGGL((s0_centroid_scores_qs<false, true>), dim3(ngroups, slices), dim3(512), ldsq, st, a, rows_per_slice);
            } else {
                FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_qs<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
                hipLaunchKernelGGL(s0_centroid_scores_qs<false>, dim3(ngroups, slices), dim3(512), ldsq, st, a, rows_per_slice);
            }
        } else if (sparse) {
            FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_f16<false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((s0_centroid_scores_f16<false, true>), grid, block, lds, st, a);
        } else {
            FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_f16<false, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((s0_centroid_scores_f16<false, false>), grid, block, lds, st, a);
        }
    } else if (impl == S0_F32) {
        hipLaunchKernelGGL(s0_centroid_scores_mfma<NC>, dim3(a.nblk, qsplit), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(s0_centroid_scores_valu, dim3(1024, a.nqueries), dim3(256), 0, st, a);
        hipLaunchKernelGGL(s0_postprocess_table<NC>, dim3(a.nblk, a.nqueries), dim3(256), 0, st, a);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ---- index build: nearest centroid of every column (token) = argmax over ALL K rows ----------------------------------
// block partials (value, first row) -> the column's global (max value, lowest row).  grid = nqueries, block = 256:
// thread (col = tid & 31, seg = tid >> 5) walks blocks seg, seg+8, ... (128-byte coalesced rows), LDS merge of the 8 segs.
__global__ __launch_bounds__(256) void s0_argmax_reduce(flmr_s0_args a, int32_t* out_codes) {
    __shared__ float sv[8][32];
    __shared__ int si[8][32];
    const int b = blockIdx.x, col = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
    float bv = FLMR_NEG_INF;
    int bi = 0x7fffffff;
    for (int e = seg; e < a.nblk; e += 8) {
        const size_t off = ((size_t)b * a.nblk + e) * a.ncol + col;
        const float v = a.part_val[off];
        const int id = a.part_idx[off];
        if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
    }
    sv[seg][col] = bv; si[seg][col] = bi;
    __syncthreads();
    if (seg == 0 && col < nqc) {
        for (int s = 1; s < 8; s++) {
            const float v = sv[s][col];
            const int id = si[s][col];
            if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
        }
        out_codes[(size_t)b * a.nq + col] = bi;
    }
}

// a: centroids (fp16-exact, K % 64 == 0), Q = the embeddings viewed as [nqueries, 32, 128], q_lens (last query may be short),
// q_hi/q_lo, part_val/part_idx [nqueries, K/64, 32], idx_bits [nqueries, K/32] (scratch).  out_codes [nqueries * 32].
int flmr_launch_centroid_argmax(flmr_s0_args& a, int32_t* out_codes, hipStream_t st) {
    if (a.K % (32 * S0_RT) != 0 || a.ncol != 32 || a.nq != 32) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "centroid argmax needs K %% 64 == 0 and 32-column tiles");
    a.nblk = a.K / (32 * S0_RT);
    a.part_rows = 32 * S0_RT;
    a.full_table = 0;
    a.thr = __builtin_inff();  // nothing qualifies: no table rows are stored
    const int qsplit = a.nqueries < 8 ? a.nqueries : 8;
    hipLaunchKernelGGL(s0_split_q, dim3((a.ncol * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq,
                       a.nq_cand, a.ncol, a.q_hi, a.q_lo, a.q_hi_only);
    const size_t lds = (size_t)S0_WAVES * 32 * S0_LDS_STRIDE * sizeof(float) + (S0_DMA_B ? (size_t)2 * S0_DCH * 2 * 8192 : (size_t)S0_CH * 2 * 32 * S0_BROW * sizeof(_Float16));
    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(s0_centroid_scores_f16<true, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((s0_centroid_scores_f16<true, true>), dim3((a.nblk + S0_WAVES - 1) / S0_WAVES, qsplit), dim3(64 * S0_WAVES), lds, st, a);
    hipLaunchKernelGGL(s0_argmax_reduce, dim3(a.nqueries), dim3(256), 0, st, a, out_codes);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

int flmr_launch_split_q(const flmr_s0_args& a, hipStream_t st) {
    hipLaunchKernelGGL(s0_split_q, dim3((a.ncol * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq,
                       a.nq_cand, a.ncol, a.q_hi, a.q_lo, a.q_hi_only);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

static int nc_bucket(int ncells) { return ncells <= 1 ? 1 : ncells <= 2 ? 2 : ncells <= 4 ? 4 : 8; }

// FLMR_S0_IMPL = f16 (default when every centroid is fp16-exact) | f32 (fp32 MFMA) | valu (plain FMA cross-check)
int flmr_launch_centroid_scores(flmr_s0_args& a, hipStream_t st) {
    const flmr_options& o = flmr_opts();
    const char* env = o.has(FLMR_OPT_S0_IMPL) ? o.v[FLMR_OPT_S0_IMPL] : nullptr;
    const bool f16_ok = a.centroids_f16_exact && (a.K % (32 * S0_RT) == 0);
    int impl = f16_ok ? S0_F16 : S0_F32;
    if (env && strcmp(env, "valu") == 0) impl = S0_VALU;
    if (env && (strcmp(env, "f32") == 0 || strcmp(env, "mfma") == 0)) impl = S0_F32;
    if (env && (strcmp(env, "f16") == 0 || strcmp(env, "f16rs") == 0) && f16_ok) impl = S0_F16;
    switch (nc_bucket(a.ncells)) {
        case 1: return launch_s0_t<1>(a, st, impl);
        case 2: return launch_s0_t<2>(a, st, impl);
        case 4: return launch_s0_t<4>(a, st, impl);
        default: return launch_s0_t<8>(a, st, impl);
    }
}

// ------------------------------------------------------------------------------------------------
// S0b: merge block partials -> per-token top-ncells -> unique cells.  grid = nqueries, block = 1024.
// ------------------------------------------------------------------------------------------------
#ifdef SC_PROFILE   // development only: clocks of wave 0 per phase, summed over the grid, printed by the launcher
__device__ unsigned long long sc_prof[8];
#define SC_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define SC_STAMP(k) do { } while (0)
#endif
#ifndef SC_WAVES
#define SC_WAVES 8  // 512 threads: the MFMA recompute needs ~190 VGPRs (B 64 + A 64 + accumulators 32 + lists)
#endif
template <int NC>
__global__ __launch_bounds__(64 * SC_WAVES) void s0_select_cells(flmr_s0_args a) {
    __shared__ int raw[1024];
    __shared__ int scan_lds[17];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qlen = a.q_lens ? a.q_lens[b] : a.nq;
    const int nqc = qlen < a.nq_cand ? qlen : a.nq_cand;
#ifdef SC_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
    raw[tid] = 0x7fffffff; raw[tid + 64 * SC_WAVES] = 0x7fffffff;
    // block-maxima partials with a single column tile: every wave scans a slice of the rows for ALL 32 columns with
    // 128-byte coalesced reads (lane = (row parity, column)) and leaves its per-column top-NC blocks in LDS; the per-column
    // loop below then merges 16 short lists instead of walking 2048 rows with a 128-byte stride per column.
    // lists hold NC + 1 entries: the extra one is the best block NOT chosen, the guard of the "hi first" verification below
    constexpr int NL = NC + 1;
    __shared__ float pre_v[SC_WAVES][32][NL];
    __shared__ int pre_i[SC_WAVES][32][NL];
    const bool pre = a.part_rows != 0 && a.ncol == 32;
    // The pipelined recompute below (sparse table, one column tile, fp16 centroid copy): the queries' hi / lo B fragments of
    // all 32 columns sit in LDS in fragment order [hi|lo][k-step][lane] (read just in time, 16 bytes per lane, conflict-free),
    // which leaves the registers for TWO selected blocks' A rows in flight
    const bool piped = pre && !a.full_table && a.centroids_f16 != nullptr && a.part_rows == 32 * S0_RT;
    __shared__ f16x8 bfrag[2][8][64];
    __shared__ int sel[SC_WAVES][4 * NC + 1];   // this wave's (column slot << 24 | block) tasks, valid ones first (+ one scratch slot)
    if (piped) {
        for (int e = tid; e < 2 * 8 * 64; e += 64 * SC_WAVES) {
            const int hl = e >> 9, st = (e >> 6) & 7, ln = e & 63;
            const _Float16* src = (hl ? a.q_lo : a.q_hi) + ((size_t)b * a.ncol + (ln & 31)) * FLMR_DIM + 64 * (ln >> 5) + 8 * st;
            bfrag[hl][st][ln] = *reinterpret_cast<const f16x8*>(src);
        }
    }
    // (with the coarse level of column maxima -- a.grp_blocks -- the pipelined path below reads 32 KB + the best groups' blocks
    // per query instead of all 256 KB: no scan here)
    const bool coarse = a.grp_blocks > 0 && pre && !a.full_table && a.centroids_f16 != nullptr && a.part_rows == 32 * S0_RT;
    if (pre && !coarse) {
        if constexpr (NC <= 2) {
        // 16-byte loads: lane = (row phase lane >> 3, four columns 4 * (lane & 7)), eight in flight per lane.  The scan
        // moves 256 KB per query and measures ~4 TB/s over the chip at 256 queries whatever the depth (8 or 32 in flight,
        // 4- or 16-byte loads): it is bandwidth-bound -- fewer block maxima per query would have to come from S0
        flmr_toplist<NL> bt[4];
#pragma unroll
        for (int c = 0; c < 4; c++) bt[c].init();
        constexpr int RS = 8 * SC_WAVES;  // rows per sweep of the workgroup
        constexpr int SCAN_U = 8;         // loads in flight per lane
        const int cg = lane & 7;
        for (int e0 = wave * 8 + (lane >> 3); e0 < a.nblk; e0 += RS * SCAN_U) {
            float4 pv[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; u++)
                pv[u] = (e0 + RS * u < a.nblk)
                            ? *reinterpret_cast<const float4*>(a.part_val + ((size_t)b * a.nblk + e0 + RS * u) * 32 + 4 * cg)
                            : make_float4(FLMR_NEG_INF, FLMR_NEG_INF, FLMR_NEG_INF, FLMR_NEG_INF);
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {   // a lane meets its rows in ascending order
                const int e = (e0 + RS * u < a.nblk) ? e0 + RS * u : 0x7fffffff;
                bt[0].insert_ascending(pv[u].x, e); bt[1].insert_ascending(pv[u].y, e);
                bt[2].insert_ascending(pv[u].z, e); bt[3].insert_ascending(pv[u].w, e);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            bt[c].merge_xor(8); bt[c].merge_xor(16); bt[c].merge_xor(32);
            if (lane < 8) {
#pragma unroll
                for (int t = 0; t < NL; t++) { pre_v[wave][4 * cg + c][t] = bt[c].v[t]; pre_i[wave][4 * cg + c][t] = bt[c].id[t]; }
            }
        }
        } else {   // longer lists: one column per lane (four lists of NC entries per lane do not fit the registers)
        flmr_toplist<NL> bt;
        bt.init();
        // 8 loads in flight per lane: one workgroup per query means one per CU, so a load -> insert chain per row would
        // expose the full memory latency 64 times per wave
        constexpr int RS = 2 * SC_WAVES;  // rows per sweep of the workgroup
        for (int e0 = wave * 2 + (lane >> 5); e0 < a.nblk; e0 += RS * 8) {
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                pv[u] = (e0 + RS * u < a.nblk) ? a.part_val[((size_t)b * a.nblk + e0 + RS * u) * 32 + (lane & 31)] : FLMR_NEG_INF;
#pragma unroll
            for (int u = 0; u < 8; u++) bt.insert(pv[u], (e0 + RS * u < a.nblk) ? e0 + RS * u : 0x7fffffff);
        }
        bt.merge_xor(32);
        if (lane < 32) {
#pragma unroll
            for (int t = 0; t < NL; t++) { pre_v[wave][lane][t] = bt.v[t]; pre_i[wave][lane][t] = bt.id[t]; }
        }
        }
    }
    __syncthreads();
    SC_STAMP(0);
    if (piped) {
        // ---- which blocks: the top-NC block maxima of each of this wave's columns (wave-uniform lists), as a task list ----
        // "hi first" stage 0 (a.q_err != NULL): the block maxima are hi-only values, each within err = q_err[b][col] of the
        // full one.  The blocks holding the true top-ncells rows are then still the chosen ones PROVIDED every other block's
        // full maximum is below the ncells-th best row found:  guard + err < v_ncells,  guard = the best maximum not chosen (any
        // unchosen block's rows are <= its hi maximum + err <= guard + err).  A column failing that test (a near-tie, ~3 % of
        // them) is redone below against ALL the blocks.
        const bool approx = a.q_err != nullptr;
        __shared__ float guard[SC_WAVES][4];
        __shared__ int redo_n[SC_WAVES];
        __shared__ int redo_k[SC_WAVES][4];
        __shared__ float redo_v[SC_WAVES][4][NC];
        __shared__ int redo_i[SC_WAVES][4][NC];
        if (lane == 0) redo_n[wave] = 0;
        int ntasks = 0;
        // With the coarse level: the NL best blocks of a column lie in its NL best GROUPS (a block's maximum is at most its
        // group's; order by (value desc, index asc) on both levels: a group ahead of another holds a block ahead of every
        // block of the other).  The group maxima of ALL of this wave's columns are requested together (128 bytes apart, ~2 us
        // away), then the NL x grp_blocks block maxima of every column's winners together: two round trips per wave, not per column.
        constexpr int KW = 32 / SC_WAVES;   // columns per wave
        flmr_toplist<NL> cbt[KW];
        if (coarse) {
            const int G = a.grp_blocks, ngrp = a.nblk / G;   // (G == 8: eight groups' blocks per sweep of the wave)
            const float* const grp = reinterpret_cast<const float*>(a.part_idx) + (size_t)b * ngrp * a.ncol;
            flmr_toplist<NL> gt[KW];
#pragma unroll
            for (int k = 0; k < KW; k++) gt[k].init();
            for (int g0 = 0; g0 < ngrp; g0 += 64 * 4) {
                float gv4[KW][4];
#pragma unroll
                for (int k = 0; k < KW; k++)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int g = g0 + 64 * u + lane, col = wave + SC_WAVES * k;
                        gv4[k][u] = (g < ngrp && col < nqc) ? grp[(size_t)g * a.ncol + col] : FLMR_NEG_INF;
                    }
#pragma unroll
                for (int k = 0; k < KW; k++)
#pragma unroll
                    for (int u = 0; u < 4; u++) gt[k].insert_ascending(gv4[k][u], (g0 + 64 * u + lane < ngrp) ? g0 + 64 * u + lane : 0x7fffffff);
            }
#pragma unroll
            for (int k = 0; k < KW; k++) gt[k].merge_wave();
            constexpr int SW = (NL + 7) / 8;   // sweeps of eight groups
            float bv[KW][SW];
            int bi[KW][SW];
#pragma unroll
            for (int k = 0; k < KW; k++)
#pragma unroll
                for (int w = 0; w < SW; w++) {
                    int gsel = 0x7fffffff;
#pragma unroll
                    for (int t = 0; t < NL; t++) gsel = (t == 8 * w + (lane >> 3)) ? gt[k].id[t] : gsel;
                    const int col = wave + SC_WAVES * k;
                    const bool ok = gsel < ngrp && col < nqc;
                    bi[k][w] = ok ? gsel * G + (lane & 7) : 0x7fffffff;
                    bv[k][w] = ok ? a.part_val[((size_t)b * a.nblk + bi[k][w]) * a.ncol + col] : FLMR_NEG_INF;
                }
#pragma unroll
            for (int k = 0; k < KW; k++) {
                cbt[k].init();
#pragma unroll
                for (int w = 0; w < SW; w++) cbt[k].insert(bv[k][w], bi[k][w]);
            }
        }
#pragma unroll
        for (int k = 0; k < KW; k++) {
            if (wave + SC_WAVES * k >= nqc) break;
            const int col = wave + SC_WAVES * k;
            flmr_toplist<NL> bt;
            bt.init();
            if (coarse) {
                bt = cbt[k];
            } else {
                for (int e = lane; e < SC_WAVES * NL; e += 64) bt.insert(pre_v[e / NL][col][e % NL], pre_i[e / NL][col][e % NL]);
            }
            bt.merge_wave();
            float gv = FLMR_NEG_INF;
#pragma unroll
            for (int t = 0; t < NL; t++) {
                const int id = __builtin_amdgcn_readfirstlane(bt.id[t]);
                if (t < a.ncells && id < a.nblk) {   // wave-uniform
                    if (lane == 0) sel[wave][ntasks] = (k << 24) | id;
                    ntasks++;
                } else if (t == a.ncells && id < a.nblk) {
                    gv = bt.v[t];
                }
            }
            if (lane == 0) guard[wave][k] = gv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        SC_STAMP(5);
        // (Measured, -DSC_PROFILE, 256 queries: block choice 50 k clocks, tasks 72 k, barrier skew 23 k, sort 10 k of 160 k.  The
        // tasks move 64 blocks x 16 KB = 1 MB per query, 268 MB per launch in ~38 us: ~7 TB/s from beyond the L2 -- that phase is
        // at the memory system's rate.  Fetching the rows in memory order (eight whole cache lines per instruction, transposed
        // into fragments through LDS) and hand-counted waits that keep the next block in flight changed nothing: 0.367 vs 0.362 ms.)
        // ---- the tasks, software-pipelined: the next block's 64 rows (16 KB, 64 VGPRs per lane) are requested before the
        // current block is multiplied, so that after the first one no memory round trip (~4 us here: one workgroup per
        // CU, nothing else to switch to) is exposed.  Same MFMA sequence as s0_centroid_scores_f16 / _qs: bitwise the S0
        // values (an output element depends only on its A row and B column). ----
        const int i = lane & 31, h = lane >> 5;
        auto issue = [&](f16x8 (&av)[S0_RT][8], int m) __attribute__((always_inline)) {
            const int r0 = (sel[wave][m] & 0xffffff) * a.part_rows;
#pragma unroll
            for (int rt = 0; rt < S0_RT; rt++) {
                const f16x8* p16 = reinterpret_cast<const f16x8*>(a.centroids_f16 + (size_t)(r0 + rt * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
                for (int st = 0; st < 8; st++) av[rt][st] = p16[st];
            }
        };
        flmr_toplist<NC> tl;
        tl.init();
        int cur_k = -1;
        auto write_cells = [&](int col) __attribute__((always_inline)) {
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < NC; t++)
                    if (t < a.ncells && tl.id[t] < a.K) raw[col * a.ncells + t] = tl.id[t];
            }
        };
        auto finish = [&]() __attribute__((always_inline)) {   // column done: exact top-NC over the wave, ids to the cell list
            if (cur_k < 0) return;
            tl.merge_wave();
            const int col = wave + SC_WAVES * cur_k;
            bool safe = true;
            if (approx) {   // (wave-uniform: the merged list is the same in every lane)
                float vn = FLMR_NEG_INF;
#pragma unroll
                for (int t = 0; t < NC; t++) vn = (t == a.ncells - 1) ? tl.v[t] : vn;
                safe = guard[wave][cur_k] + a.q_err[(size_t)b * a.ncol + col] < vn;
            }
            if (safe) {
                write_cells(col);
            } else if (lane == 0) {   // park the column: its list so far, to be completed after the pipelined loop
                const int n = redo_n[wave];
                redo_k[wave][n] = cur_k;
#pragma unroll
                for (int t = 0; t < NC; t++) { redo_v[wave][n][t] = tl.v[t]; redo_i[wave][n][t] = tl.id[t]; }
                redo_n[wave] = n + 1;
            }
            tl.init();
        };
        auto compute = [&](const f16x8 (&av)[S0_RT][8], int m) __attribute__((always_inline)) {
            const int ent = sel[wave][m];
            const int k = __builtin_amdgcn_readfirstlane(ent >> 24);
            if (k != cur_k) { finish(); cur_k = k; }
            const int col = wave + SC_WAVES * k;
            const int r0 = (ent & 0xffffff) * a.part_rows;
            const bool mine = i == (col & 31);
#pragma unroll
            for (int rt = 0; rt < S0_RT; rt++) {
                f32x16 ah, al;
#pragma unroll
                for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
                // B fragments one k-step ahead of their MFMAs and no further (the fence keeps the scheduler from hoisting all 16
                // reads -- 64 VGPRs -- above the first MFMA, which would spill the second block's rows)
                f16x8 nbh = bfrag[0][0][lane], nbl = bfrag[1][0][lane];
#pragma unroll
                for (int st = 0; st < 8; st++) {
                    const f16x8 bh = nbh, bl = nbl;
                    if (st + 1 < 8) { nbh = bfrag[0][st + 1][lane]; nbl = bfrag[1][st + 1][lane]; }
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][st], bh, ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][st], bl, al, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                    const int row = r0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    tl.insert(mine ? v : FLMR_NEG_INF, mine ? row : 0x7fffffff);
                }
            }
        };
#if SC_WAVES > 8   // 16 waves (128 VGPRs each): ONE block's rows in flight per wave, the other three waves of the SIMD cover its latency
        f16x8 avA[S0_RT][8];
        for (int m = 0; m < ntasks; m++) {
            issue(avA, m);
            compute(avA, m);
        }
#else
        f16x8 avA[S0_RT][8], avB[S0_RT][8];
        if (ntasks > 0) issue(avA, 0);
        for (int m = 0; m < ntasks; m += 2) {
            if (m + 1 < ntasks) issue(avB, m + 1);
            compute(avA, m);
            if (m + 2 < ntasks) issue(avA, m + 2);
            if (m + 1 < ntasks) compute(avB, m + 1);
        }
#endif
        finish();
        SC_STAMP(6);
        // ---- parked columns: every block whose hi maximum + err reaches the current ncells-th best row is recomputed too (the
        // list can only improve, which only tightens the test: a block skipped earlier stays skippable) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nredo = approx ? redo_n[wave] : 0;
        for (int rd = 0; rd < nredo; rd++) {
            const int k = redo_k[wave][rd];
            const int col = wave + SC_WAVES * k;
            const float err = a.q_err[(size_t)b * a.ncol + col];
            // the list so far lives in lane 0 only while rows are inserted (every lane holding it would duplicate its entries in
            // the merge); merged, it is the same in all lanes for the wave-uniform tests
            tl.init();
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < NC; t++) { tl.v[t] = redo_v[wave][rd][t]; tl.id[t] = redo_i[wave][rd][t]; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) tl.merge_xor(m);
            cur_k = -1;
            // the column's block maxima, sixteen 64-block groups in flight at a time (one dependent round trip per group would
            // cost ~2 us each here: one workgroup per CU, nothing else to switch to)
            for (int jb = 0; jb < a.nblk; jb += 64 * 16) {
                float Av[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int j = jb + 64 * u + lane;
                    Av[u] = j < a.nblk ? a.part_val[((size_t)b * a.nblk + j) * a.ncol + col] : FLMR_NEG_INF;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int j0 = jb + 64 * u;
                    if (j0 >= a.nblk) break;   // wave-uniform
                    const int j = j0 + lane;
                    const float A = Av[u];
                    float vn = FLMR_NEG_INF;
#pragma unroll
                    for (int t = 0; t < NC; t++) vn = (t == a.ncells - 1) ? tl.v[t] : vn;
                    unsigned long long need = __ballot(j < a.nblk && !(A + err < vn));
                    while (need) {
                        const int src = __ffsll((long long)need) - 1;
                        need &= need - 1;
                        const int jj = j0 + src;
                        bool chosen = false;   // (wave-uniform) one of the blocks the pipelined pass already recomputed
                        for (int m = 0; m < ntasks; m++) chosen |= (sel[wave][m] >> 24) == k && (sel[wave][m] & 0xffffff) == jj;
                        if (chosen) continue;
                        // (the list may have improved since the ballot: re-test this block against it)
                        const float Aj = __shfl(A, src, 64);
                        vn = FLMR_NEG_INF;
#pragma unroll
                        for (int t = 0; t < NC; t++) vn = (t == a.ncells - 1) ? tl.v[t] : vn;
                        if (Aj + err < vn) continue;
                        if (lane == 0) sel[wave][4 * NC] = (k << 24) | jj;   // the scratch task slot
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (lane != 0) tl.init();
                        issue(avA, 4 * NC);
                        cur_k = k;              // compute() must not take this for a column change
                        compute(avA, 4 * NC);
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) tl.merge_xor(m);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            write_cells(col);
        }
    } else
    for (int col = wave; col < nqc; col += SC_WAVES) {
        flmr_toplist<NC> tl;
        tl.init();
        if (a.part_rows == 0) {
            // partials are per-block top lists: merge them
            const int nent = a.nblk * NC;
            for (int e = lane; e < nent; e += 64) {
                const size_t off = (((size_t)b * a.nblk + (e / NC)) * a.ncol + col) * NC + (e % NC);
                tl.insert(a.part_val[off], a.part_idx[off]);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) tl.merge_xor(m);
        } else {
            // partials are per-block column maxima: the global top-NC scores lie in the NC blocks with the largest
            // maxima (a block holding a top-NC score can be beaten by at most NC-1 other blocks), so re-read just
            // those blocks' rows of this column from the table and take the exact top-NC (value desc, index asc)
            flmr_toplist<NC> bt;
            bt.init();
            if (pre) {
                for (int e = lane; e < SC_WAVES * NC; e += 64) bt.insert(pre_v[e / NC][col][e % NC], pre_i[e / NC][col][e % NC]);
            } else {
                for (int e = lane; e < a.nblk; e += 64) bt.insert(a.part_val[((size_t)b * a.nblk + e) * a.ncol + col], e);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) bt.merge_xor(m);
            const float* cs_b = a.cs + (size_t)b * a.K * a.ncol;
            const bool have_table = a.full_table || (a.ncol >> 5) != 1;
#pragma unroll
            for (int t = 0; t < NC; t++) {
                if (t >= a.ncells || bt.id[t] >= a.nblk) continue;  // wave-uniform (bt is identical in all lanes)
                const int r0 = bt.id[t] * a.part_rows;
                if (have_table) {
                    for (int r = lane; r < a.part_rows; r += 64)
                        if (r0 + r < a.K) tl.insert(cs_b[(size_t)(r0 + r) * a.ncol + col], r0 + r);
                } else {
                    // sparse table: recompute the block's scores with the SAME MFMA sequence as s0_centroid_scores_f16
                    // (bitwise identical values: an output element depends only on its A row and B column)
                    const int i = lane & 31, h = lane >> 5;
                    f16x8 bh[8], bl[8];
                    {
                        const int bcol = (col & ~31) + i;
                        const f16x8* ph = reinterpret_cast<const f16x8*>(a.q_hi + ((size_t)b * a.ncol + bcol) * FLMR_DIM + 64 * h);
                        const f16x8* pl = reinterpret_cast<const f16x8*>(a.q_lo + ((size_t)b * a.ncol + bcol) * FLMR_DIM + 64 * h);
#pragma unroll
                        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
                    }
                    // the A fragments of both row tiles are requested before the first MFMA (fp16 copy of the centroids when
                    // the index holds one: half the bytes, no conversion)
                    f16x8 av[S0_RT][8];
#pragma unroll
                    for (int rt = 0; rt < S0_RT; rt++) {
                        if (a.centroids_f16) {
                            const f16x8* p16 = reinterpret_cast<const f16x8*>(a.centroids_f16 + (size_t)(r0 + rt * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
                            for (int s = 0; s < 8; s++) av[rt][s] = p16[s];
                        } else {
                            const float4* p = reinterpret_cast<const float4*>(a.centroids + (size_t)(r0 + rt * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
                            for (int s = 0; s < 8; s++) {
                                const float4 x = p[2 * s], y = p[2 * s + 1];
                                av[rt][s][0] = (_Float16)x.x; av[rt][s][1] = (_Float16)x.y; av[rt][s][2] = (_Float16)x.z; av[rt][s][3] = (_Float16)x.w;
                                av[rt][s][4] = (_Float16)y.x; av[rt][s][5] = (_Float16)y.y; av[rt][s][6] = (_Float16)y.z; av[rt][s][7] = (_Float16)y.w;
                            }
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < S0_RT; rt++) {
                        f32x16 ah, al;
#pragma unroll
                        for (int r = 0; r < 16; r++) { ah[r] = 0.0f; al[r] = 0.0f; }
#pragma unroll
                        for (int s = 0; s < 8; s++) {
                            ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][s], bh[s], ah, 0, 0, 0);
                            al = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rt][s], bl[s], al, 0, 0, 0);
                        }
                        const bool mine = i == (col & 31);
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float v = fmaf(al[r], 1.0f / 2048.0f, ah[r]);
                            const int row = r0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            tl.insert(mine ? v : FLMR_NEG_INF, mine ? row : 0x7fffffff);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) tl.merge_xor(m);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < NC; t++)
                if (t < a.ncells && tl.id[t] < a.K) raw[col * a.ncells + t] = tl.id[t];
        }
    }
    SC_STAMP(1);
    __syncthreads();
    SC_STAMP(2);
    // ascending sort of <= 1024 ids (INT_MAX padded) + unique; two elements per thread
    // (only the first nqc * ncells slots can hold an id: the network is sized for those, not for all 1024)
    constexpr int NT = 64 * SC_WAVES;
    int nsort = 64;
    while (nsort < nqc * a.ncells) nsort <<= 1;
    if (nsort == 64) {   // (block-uniform) one value per lane: the first wave sorts in registers, no barriers (21 shuffle steps)
        if (wave == 0) {
            int x = raw[lane];
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const int y = __shfl_xor(x, j, 64);
                    const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                    x = keep_min ? min(x, y) : max(x, y);
                }
            }
            const int before = __shfl_up(x, 1, 64);
            const bool first = x != 0x7fffffff && (lane == 0 || before != x);
            const unsigned long long m = __ballot(first);
            if (first) a.cells[(size_t)b * a.max_cells + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = x;
            if (lane == 0) a.ncell[b] = __popcll(m);
        }
        SC_STAMP(3);
        SC_STAMP(4);
    } else {
    for (int k = 2; k <= nsort; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 1024 / NT; u++) {
                const int t = tid + u * NT;
                const int p = t ^ j;
                if (p > t && p < nsort) {
                    const int x = raw[t], y = raw[p];
                    const bool asc = ((t & k) == 0);
                    if (asc ? (x > y) : (x < y)) { raw[t] = y; raw[p] = x; }
                }
            }
            __syncthreads();
        }
    }
    SC_STAMP(3);
    int vv[1024 / NT], flag[1024 / NT], cnt = 0;
#pragma unroll
    for (int u = 0; u < 1024 / NT; u++) {  // thread t owns the consecutive elements t*2, t*2+1 so that ranks stay ordered
        const int t = tid * (1024 / NT) + u;
        vv[u] = raw[t];
        flag[u] = (vv[u] != 0x7fffffff) && (t == 0 || raw[t - 1] != vv[u]);
        cnt += flag[u];
    }
    int total;
    int pos = flmr_block_exclusive_scan(cnt, scan_lds, &total);
#pragma unroll
    for (int u = 0; u < 1024 / NT; u++)
        if (flag[u]) a.cells[(size_t)b * a.max_cells + pos++] = vv[u];
    if (tid == 0) a.ncell[b] = total;
    SC_STAMP(4);
    }
#ifdef SC_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&sc_prof[i], (unsigned long long)pt[i]);
        atomicAdd(&sc_prof[7], 1ull);
    }
#endif
}

int flmr_launch_select_cells(const flmr_s0_args& a, hipStream_t st) {
    if ((int64_t)a.nq_cand * a.ncells > 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq_cand*ncells > 1024");
    switch (nc_bucket(a.ncells)) {
        case 1: hipLaunchKernelGGL(s0_select_cells<1>, dim3(a.nqueries), dim3(64 * SC_WAVES), 0, st, a); break;
        case 2: hipLaunchKernelGGL(s0_select_cells<2>, dim3(a.nqueries), dim3(64 * SC_WAVES), 0, st, a); break;
        case 4: hipLaunchKernelGGL(s0_select_cells<4>, dim3(a.nqueries), dim3(64 * SC_WAVES), 0, st, a); break;
        default: hipLaunchKernelGGL(s0_select_cells<8>, dim3(a.nqueries), dim3(64 * SC_WAVES), 0, st, a); break;
    }
#ifdef SC_PROFILE
    {
        unsigned long long h[8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(sc_prof), sizeof(h));
        fprintf(stderr, "[sc] blocks %llu; clocks per block: scan %.0f columns %.0f (block choice %.0f, tasks %.0f, the rest = parked columns) barrier %.0f sort %.0f unique+write %.0f\n", h[7],
                (double)(h[0]) / h[7], (double)(h[1] + h[5] + h[6]) / h[7], (double)h[5] / h[7], (double)h[6] / h[7], (double)h[2] / h[7], (double)h[3] / h[7], (double)h[4] / h[7]);
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sc_prof), z, sizeof(z));
    }
#endif
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// S0c: OR the IVF lists of the probed cells into the per-query passage bitmap.
// grid = (nqueries, max_cells): blockIdx.x = query keeps a query's blocks on one XCD (block id % 8).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s0_ivf_mark(const int32_t* cells, const int32_t* ncell, int32_t max_cells,
                                                   const int32_t* ivf_pids, const int64_t* ivf_offsets,
                                                   uint32_t* bitmap, int64_t bitmap_words) {
    const int b = blockIdx.x, ci = blockIdx.y;
    if (ci >= ncell[b]) return;
    const int c = cells[(size_t)b * max_cells + ci];
    const int64_t beg = ivf_offsets[c], end = ivf_offsets[c + 1];
    uint32_t* bm = bitmap + (size_t)b * bitmap_words;
    for (int64_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
        const int pid = ivf_pids[e];
        atomicOr(&bm[pid >> 5], 1u << (pid & 31));
    }
}

int flmr_launch_ivf_mark(const int32_t* cells, const int32_t* ncell, int32_t max_cells, int32_t nqueries,
                         const int32_t* ivf_pids, const int64_t* ivf_offsets, uint32_t* bitmap, int64_t bitmap_words,
                         hipStream_t st) {
    FLMR_HIP(hipMemsetAsync(bitmap, 0, (size_t)nqueries * bitmap_words * sizeof(uint32_t), st));
    hipLaunchKernelGGL(s0_ivf_mark, dim3(nqueries, max_cells), dim3(256), 0, st, cells, ncell, max_cells, ivf_pids,
                       ivf_offsets, bitmap, bitmap_words);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// S0d: bitmap -> ascending candidate pid list (popcount + block scan per 1024-word tile).
// grid = nqueries, block = 1024.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void s0_compact(const uint32_t* bitmap, int64_t bitmap_words, int64_t num_passages,
                                                   int32_t* cand, int64_t cand_cap, int32_t* cand_count,
                                                   int32_t* overflow) {
    __shared__ int scan_lds[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t* bm = bitmap + (size_t)b * bitmap_words;
    int32_t* out = cand + (size_t)b * cand_cap;
    int64_t base = 0;
    for (int64_t w0 = 0; w0 < bitmap_words; w0 += 1024) {
        const int64_t w = w0 + tid;
        uint32_t bits = (w < bitmap_words) ? bm[w] : 0u;
        int total;
        int pos = flmr_block_exclusive_scan(__popc(bits), scan_lds, &total);
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            const int64_t o = base + pos++;
            if (o < cand_cap) out[o] = (int32_t)(w * 32 + bit);
        }
        base += total;
    }
    if (tid == 0) {
        if (base > cand_cap) { atomicExch(overflow, 1); base = cand_cap; }
        cand_count[b] = (int32_t)base;
    }
}

int flmr_launch_compact(const uint32_t* bitmap, int64_t bitmap_words, int64_t num_passages, int32_t nqueries,
                        int32_t* cand, int64_t cand_cap, int32_t* cand_count, int32_t* overflow, hipStream_t st) {
    hipLaunchKernelGGL(s0_compact, dim3(nqueries), dim3(1024), 0, st, bitmap, bitmap_words, num_passages, cand,
                       cand_cap, cand_count, overflow);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
