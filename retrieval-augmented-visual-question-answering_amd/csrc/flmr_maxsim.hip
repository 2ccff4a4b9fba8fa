// Stage 3: exact late-interaction score of the finalists, fused end to end.
//
// Reference (CPU path): TPC/search/index_storage.py:160-177 --
//   decompress_residuals_cpp (TPC/search/decompress_residuals.cpp:27-78)  D[t,:] = weight + centroid
//   F.normalize(D, p=2, dim=-1)                                            D[t,:] / max(||D[t,:]||, 1e-12)
//   colbert_score_packed (TPC/modeling/colbert.py:289-311)                 S = D @ Q^T (all nq rows of Q)
//   segmented_maxsim_cpp (TPC/modeling/segmented_maxsim.cpp:22-93)         per doc: max over tokens, INIT 0, sum over nq
//
// MI355X design: one workgroup per (query, finalist document); nothing intermediate touches HBM.
//   * each lane decompresses HALF a token row (64 dims) straight into registers: 8*nbits residual bytes,
//     a 256-byte run of the centroid row, and the fused 256-entry byte->weights table in LDS (4 KB at nbits=2);
//     that register image IS the A operand of v_mfma_f32_32x32x2_f32 (lane = (token, k-half)), so the
//     32 tokens x 32 query-tokens x 128 contraction needs no LDS staging of D at all;
//   * Q rows are the B operand (lane = (query token, k-half)), read from L2 (16 KB per 32 query tokens,
//     shared by all of a query's workgroups, which sit on one XCD because blockIdx.x = query);
//   * L2 normalisation is applied to the 32x32 score tile (one multiply per score by 1/max(||d||,eps))
//     instead of to the 128-wide row; the per-token max starts at 0 exactly as the CPU extension does, so
//     column maxima are non-negative and an integer LDS atomicMax on the float bits combines the 4 waves;
//   * the final sum over query tokens is k-ascending fp32 (the oracle's order; torch's own reduction order
//     is not defined, hence the 1e-4 score tolerance of the parity tests).
#include "flmr_device.h"

template <int NBITS>
__device__ __forceinline__ void decompress_half_row(const uint8_t* __restrict__ res /* 8*NBITS bytes */,
                                                    const float* __restrict__ cen /* 64 floats */,
                                                    const float* wlut /* LDS [256][8/NBITS] */, float* a /* 64 */) {
    constexpr int VPB = 8 / NBITS;
    constexpr int NB = 8 * NBITS;  // bytes covering 64 dims
    const uint2* r2 = reinterpret_cast<const uint2*>(res);
    const float4* c4 = reinterpret_cast<const float4*>(cen);
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const float4 v = c4[t];
        a[4 * t + 0] = v.x; a[4 * t + 1] = v.y; a[4 * t + 2] = v.z; a[4 * t + 3] = v.w;
    }
#pragma unroll
    for (int w = 0; w < NB / 8; w++) {
        const uint2 pk = r2[w];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t word = e < 4 ? pk.x : pk.y;
            const uint32_t byte = (word >> (8 * (e & 3))) & 255u;
            const int kb = w * 8 + e;
#pragma unroll
            for (int l = 0; l < VPB; l++) a[kb * VPB + l] = wlut[byte * VPB + l] + a[kb * VPB + l];
        }
    }
}

template <int NBITS>
__global__ __launch_bounds__(256) void maxsim_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                     const uint8_t* __restrict__ residuals,
                                                     const int64_t* __restrict__ doc_offsets,
                                                     const float* __restrict__ centroids,
                                                     const float* __restrict__ wlut_g, int y_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    constexpr int PACKED = FLMR_DIM / VPB;
    float* wlut = reinterpret_cast<float*>(smem);                       // [256 * VPB]
    int* colmax = reinterpret_cast<int*>(smem + 256 * VPB * sizeof(float));  // [nq]
    const int b = blockIdx.x, d = y_base + blockIdx.y;
    if (d >= m.counts[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    const int pid = m.pids[(size_t)b * m.pid_stride + d];
    const int64_t off = doc_offsets[pid];
    const int len = (int)(doc_offsets[pid + 1] - off);
    for (int t = tid; t < 256 * VPB; t += 256) wlut[t] = wlut_g[t];
    for (int t = tid; t < qlen; t += 256) colmax[t] = 0;  // +0.0f
    __syncthreads();
    const float* Qb = m.Q + (size_t)b * m.nq * FLMR_DIM;

    for (int t0 = wave * 32; t0 < len; t0 += 128) {
        const int tok = t0 + i;
        const bool valid = tok < len;
        float a[64];
        if (valid) {
            const int code = codes[off + tok];
            decompress_half_row<NBITS>(residuals + (size_t)(off + tok) * PACKED + h * (PACKED / 2),
                                       centroids + (size_t)code * FLMR_DIM + 64 * h, wlut, a);
        } else {
#pragma unroll
            for (int t = 0; t < 64; t++) a[t] = 0.0f;
        }
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 64; t++) ss = fmaf(a[t], a[t], ss);
        ss += __shfl_xor(ss, 32, 64);
        float nrm = sqrtf(ss);
        nrm = nrm < 1e-12f ? 1e-12f : nrm;
        const float inv = 1.0f / nrm;

        for (int q0 = 0; q0 < qlen; q0 += 32) {
            const int col = q0 + i;
            float bv[64];
            if (col < qlen) {
                const float4* p = reinterpret_cast<const float4*>(Qb + (size_t)col * FLMR_DIM + 64 * h);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float4 v = p[t];
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
            for (int s = 0; s < 64; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[s], acc, 0, 0, 0);
            float mx = 0.0f;  // segmented_maxsim.cpp:58-59: running max starts at zero
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // token row of this accumulator register
                const float inv_r = __shfl(inv, row, 64);
                mx = fmaxf(mx, acc[r] * inv_r);
            }
            mx = flmr_xhalf_max(mx);
            if (h == 0 && col < qlen) atomicMax(&colmax[col], __float_as_int(mx));
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int k = 0; k < qlen; k++) s += __int_as_float(colmax[k]);
        if (m.keys) m.keys[(size_t)b * m.key_stride + d] = flmr_make_key(s, pid);
        if (m.scores) m.scores[(size_t)b * m.key_stride + d] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 3 in the reference's CUDA-path numerics (FLMR_NUMERICS_GPU_FP16, include/flmr_hip.h): what IndexScorer.score_pids
// computes when use_gpu is set (TPC/search/index_storage.py:157-158,176-177):
//   lookup_pids -> ResidualCodec.decompress (residual.py:242-278): decompress_residuals.cu writes half(weight) then adds the
//     half centroid value in half arithmetic -> D = half(c + half(w)); F.normalize on the half tensor: the norm is
//     accumulated in fp32 and stored as half, the quotient is rounded to half;
//   colbert_score_packed (colbert.py:289-311, use_gpu branch): scores = D_half @ Q_half^T (fp32 accumulation, half result),
//     padded per passage with -9999 (half: -10000), max over the passage's tokens -- NO zero clamp on this path --, then
//     `.sum(-1)` of the half maxima: fp32 accumulation, half result.
// Same structure as maxsim_kernel (one workgroup per (query, finalist), the fp32 MFMA on operands that are exactly fp16, so
// every product is exact and the accumulation is fp32 like the GEMM's); it is an optional compatibility mode, not the tuned
// path.  A row of norm 0 (never produced by a real index) scores 0 instead of the reference's NaN.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int s3g_enc(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float s3g_dec(int e) { return __int_as_float(e ^ ((e >> 31) & 0x7fffffff)); }

template <int NBITS>
__global__ __launch_bounds__(256) void maxsim_gpu_fp16_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                              const uint8_t* __restrict__ residuals,
                                                              const int64_t* __restrict__ doc_offsets,
                                                              const float* __restrict__ centroids,
                                                              const float* __restrict__ wlut_g, int y_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    constexpr int PACKED = FLMR_DIM / VPB;
    float* wlut = reinterpret_cast<float*>(smem);                            // [256 * VPB], rounded to fp16 (bucket_weights.half())
    int* colmax = reinterpret_cast<int*>(smem + 256 * VPB * sizeof(float));  // [nq] order-preserving images
    const int b = blockIdx.x, d = y_base + blockIdx.y;
    if (d >= m.counts[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    const int pid = m.pids[(size_t)b * m.pid_stride + d];
    const int64_t off = doc_offsets[pid];
    const int len = (int)(doc_offsets[pid + 1] - off);
    const float pad = -10000.0f;   // half(-9999)
    for (int t = tid; t < 256 * VPB; t += 256) wlut[t] = flmr_round_f16(wlut_g[t]);
    for (int t = tid; t < qlen; t += 256) colmax[t] = s3g_enc(pad);
    __syncthreads();
    const float* Qb = m.Q + (size_t)b * m.nq * FLMR_DIM;

    for (int t0 = wave * 32; t0 < len; t0 += 128) {
        const int tok = t0 + i;
        const bool valid = tok < len;
        float a[64];
        if (valid) {
            const int code = codes[off + tok];
            decompress_half_row<NBITS>(residuals + (size_t)(off + tok) * PACKED + h * (PACKED / 2),
                                       centroids + (size_t)code * FLMR_DIM + 64 * h, wlut, a);
        } else {
#pragma unroll
            for (int t = 0; t < 64; t++) a[t] = 0.0f;
        }
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 64; t++) {
            a[t] = flmr_round_f16(a[t]);          // half(c + half(w)): the fp32 sum of two halves rounds like the half add
            ss = fmaf(a[t], a[t], ss);
        }
        ss += __shfl_xor(ss, 32, 64);
        const float nrm = flmr_round_f16(sqrtf(ss));
        const float inv_ok = nrm > 0.0f ? 1.0f : 0.0f;
#pragma unroll
        for (int t = 0; t < 64; t++) a[t] = inv_ok != 0.0f ? flmr_round_f16(a[t] / nrm) : 0.0f;

        for (int q0 = 0; q0 < qlen; q0 += 32) {
            const int col = q0 + i;
            float bv[64];
            if (col < qlen) {
                const float4* p = reinterpret_cast<const float4*>(Qb + (size_t)col * FLMR_DIM + 64 * h);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float4 v = p[t];
                    bv[4 * t + 0] = flmr_round_f16(v.x); bv[4 * t + 1] = flmr_round_f16(v.y);
                    bv[4 * t + 2] = flmr_round_f16(v.z); bv[4 * t + 3] = flmr_round_f16(v.w);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 64; t++) bv[t] = 0.0f;
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 64; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[s], acc, 0, 0, 0);
            float mx = pad;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // token row of this accumulator register
                mx = fmaxf(mx, t0 + row < len ? flmr_round_f16(acc[r]) : pad);
            }
            mx = flmr_xhalf_max(mx);
            if (h == 0 && col < qlen) atomicMax(&colmax[col], s3g_enc(mx));
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int k = 0; k < qlen; k++) s += s3g_dec(colmax[k]);
        s = flmr_round_f16(s);
        if (m.keys) m.keys[(size_t)b * m.key_stride + d] = flmr_make_key(s, pid);
        if (m.scores) m.scores[(size_t)b * m.key_stride + d] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// S3, fp16-split variant (default when the centroids are fp16-exact): one WAVE per finalist document.
//   * a wave owns a strided list of documents of one query; lanes fetch all their (pid, offset, length) triples in
//     parallel up front, the NEXT document's codes are loaded while the current one is scored, and the NEXT token
//     tile's centroid half-rows (fp16 image, 128 B per lane) + residual bytes are issued right after the current tile
//     has been decompressed -- so the dependent gather chain pid -> offset -> code -> row is off the critical path;
//   * decompress -> fp32 row, L2 norm (fp32), scale by 1/max(||d||,eps), THEN split d = d_hi + 2^-11 d_lo in fp16;
//     Q is split the same way once per batch (s3_split_q);  d.q ~= hi.hi + 2^-11 (hi.lo + lo.hi): three
//     v_mfma_f32_32x32x16_f16 per 16 dims instead of eight v_mfma_f32_32x32x2_f32, ~2^-21 relative error per term;
//   * the per-query-token running max (zero-initialised) lives in a per-wave LDS row; no block barrier, no atomics.
// ------------------------------------------------------------------------------------------------
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void s3_split_q(const float* Q, const int32_t* q_lens, int nq, int nqp, _Float16* q_hi,
                                                  _Float16* q_lo) {
    const int b = blockIdx.y;
    const int qlen = q_lens ? q_lens[b] : nq;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nqp * FLMR_DIM; e += gridDim.x * blockDim.x) {
        const int row = e / FLMR_DIM;
        float v = 0.0f;
        if (row < qlen && row < nq) v = Q[((size_t)b * nq + row) * FLMR_DIM + (e % FLMR_DIM)];
        const _Float16 hi = (_Float16)v;
        q_hi[(size_t)b * nqp * FLMR_DIM + e] = hi;
        q_lo[(size_t)b * nqp * FLMR_DIM + e] = (_Float16)((v - (float)hi) * 2048.0f);
    }
}

// the VPB bucket weights of one residual byte with ONE LDS read (b128 / b64): the table is read with random per-lane
// addresses, and VPB separate 4-byte reads cost VPB times the bank-conflict cycles
typedef float s3f4 __attribute__((ext_vector_type(4)));
typedef float s3f2 __attribute__((ext_vector_type(2)));
template <int VPB>
__device__ __forceinline__ void s3_lut(const float* wlut_, uint32_t byte, float* w) {
    const float* wlut = static_cast<const float*>(__builtin_assume_aligned(wlut_, 16));  // the LDS carve starts with the table
    if constexpr (VPB == 8) {
        const s3f4 x = reinterpret_cast<const s3f4*>(wlut)[byte * 2], y = reinterpret_cast<const s3f4*>(wlut)[byte * 2 + 1];
#pragma unroll
        for (int l = 0; l < 4; l++) { w[l] = x[l]; w[4 + l] = y[l]; }
    } else if constexpr (VPB == 4) {
        const s3f4 x = reinterpret_cast<const s3f4*>(wlut)[byte];
#pragma unroll
        for (int l = 0; l < 4; l++) w[l] = x[l];
    } else if constexpr (VPB == 2) {
        const s3f2 x = reinterpret_cast<const s3f2*>(wlut)[byte];
        w[0] = x[0]; w[1] = x[1];
    } else {
        w[0] = wlut[byte];
    }
}

// ------------------------------------------------------------------------------------------------
// One lane's half row (64 dims): decode (centroid fp16 + bucket weight), normalise, split into the fp16 hi / lo MFMA A operand.
//   x   = w + (float)c                       one v_fma_mix_f32 (c * 1.0 + w: the fp16 source is read in place -- same single rounding)
//   ss += x * x                              k-ascending, one fp32 chain (decompress + normalise order of the oracle)
//   v   = x * inv, vs = v * 2048             packed fp32 multiplies
//   hi  = fp16(v)                            v_cvt_pk_f16_f32 (two per instruction, round to nearest even)
//   lo  = fp16(fma(hi, -2048, vs))           v_fma_mixlo/mixhi_f16: hi is read from the packed register, the result lands in its
//                                            half of the packed lo register -- (v - hi) * 2048 exactly (v - hi is representable,
//                                            the scalings are powers of two), so lo is the fp16 rounding of the exact remainder
// The mixed-precision forms do what the plain C expressions say (fp32 fma, then one conversion), bit for bit; written as
// instructions because the compiler otherwise converts hi back to fp32 and packs the halves separately (520 -> ~340 VALU
// operations per tile in a kernel that is bound by VALU issue, PMC: profiles/r02_pmc_summary.csv).
// `word(wi)`: wi-th 32-bit word of the lane's 8 * NBITS residual bytes.
// ------------------------------------------------------------------------------------------------
typedef uint32_t s3u4 __attribute__((ext_vector_type(4)));
typedef float s3v2 __attribute__((ext_vector_type(2)));
typedef _Float16 s3h2 __attribute__((ext_vector_type(2)));

template <int HALF>
__device__ __forceinline__ float s3_add_f16(uint32_t cpk, float w) {  // w + (float)half HALF of cpk
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(cpk), "v"(w));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(cpk), "v"(w));
    return r;
}

template <int NBITS, typename WordFn>
__device__ __forceinline__ void s3_decode_split(const float* wlut, WordFn word, const hf8 (&c)[8], bool valid, hf8 (&ah)[8], hf8 (&al)[8]) {
    constexpr int VPB = 8 / NBITS;
    float d[64];
    float ss = 0.0f;
#pragma unroll
    for (int wq = 0; wq < NBITS; wq++) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t byte = (word(wq * 2 + (e >> 2)) >> (8 * (e & 3))) & 255u;
            const int kb = wq * 8 + e;
            float wv[VPB];
            s3_lut<VPB>(wlut, byte, wv);
#pragma unroll
            for (int l = 0; l < VPB; l++) {
                const int dd = kb * VPB + l;
                const uint32_t cpk = __builtin_bit_cast(s3u4, c[dd >> 3])[(dd & 7) >> 1];
                const float v = (dd & 1) ? s3_add_f16<1>(cpk, wv[l]) : s3_add_f16<0>(cpk, wv[l]);
                d[dd] = v;
                ss = fmaf(v, v, ss);
            }
        }
    }
    ss += __shfl_xor(ss, 32, 64);
    float nrm = sqrtf(ss);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;
    const float inv = valid ? 1.0f / nrm : 0.0f;  // padding rows become exact zeros
    const float m2048 = -2048.0f;
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++) {
        s3u4 hpk, lpk;
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            const int dd = s8 * 8 + 2 * pr;
            s3v2 v = {d[dd], d[dd + 1]};
            v *= inv;
            const s3v2 vs = v * 2048.0f;
            const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, s3h2));
            uint32_t lo;
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(m2048), "v"(vs[0]));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(m2048), "v"(vs[1]));
            hpk[pr] = hi;
            lpk[pr] = lo;
        }
        ah[s8] = __builtin_bit_cast(hf8, hpk);
        al[s8] = __builtin_bit_cast(hf8, lpk);
    }
}

// FLMR_NUMERICS_GPU_FP16 (maxsim_gpu_fp16_kernel's arithmetic on this kernel's pipeline): the lane's half row is
// half(c + half(w)) (`wlut` holds the weights already rounded to fp16), the norm is accumulated in fp32 and rounded to fp16, the
// quotient is rounded to fp16 -- the A operand is that fp16 row, there is no lo part.  x / norm is evaluated as x * (1 / norm):
// before the fp16 rounding the two differ by an fp32 ulp, so an element can differ from the true quotient's rounding only when
// it sits within 2^-13 relative of a rounding boundary.
template <int NBITS, typename WordFn>
__device__ __forceinline__ void s3_decode_f16(const float* wlut, WordFn word, const hf8 (&c)[8], hf8 (&ah)[8]) {
    constexpr int VPB = 8 / NBITS;
    float d[64];
    float ss = 0.0f;
#pragma unroll
    for (int wq = 0; wq < NBITS; wq++) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t byte = (word(wq * 2 + (e >> 2)) >> (8 * (e & 3))) & 255u;
            const int kb = wq * 8 + e;
            float wv[VPB];
            s3_lut<VPB>(wlut, byte, wv);
#pragma unroll
            for (int l = 0; l < VPB; l++) {
                const int dd = kb * VPB + l;
                const uint32_t cpk = __builtin_bit_cast(s3u4, c[dd >> 3])[(dd & 7) >> 1];
                const float x = flmr_round_f16((dd & 1) ? s3_add_f16<1>(cpk, wv[l]) : s3_add_f16<0>(cpk, wv[l]));
                d[dd] = x;
                ss = fmaf(x, x, ss);
            }
        }
    }
    ss += __shfl_xor(ss, 32, 64);
    const float nrm = flmr_round_f16(sqrtf(ss));
    const float inv = nrm > 0.0f ? 1.0f / nrm : 0.0f;
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++) {
        s3u4 hpk;
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            const int dd = s8 * 8 + 2 * pr;
            const s3v2 v = {d[dd] * inv, d[dd + 1] * inv};
            hpk[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, s3h2));
        }
        ah[s8] = __builtin_bit_cast(hf8, hpk);
    }
}

// maximum over a lane's 16 accumulator rows of hi + lo / 2048, zero floor (segmented_maxsim.cpp:58-59), as a tree
__device__ __forceinline__ float s3_tile_max(const f32x16& acch, const f32x16& accl) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = fmaf(accl[r], 1.0f / 2048.0f, acch[r]);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r + 8]);
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], v[r + 4]);
    return fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), 0.0f);
}

template <int NBITS>
struct s3_raw {          // one lane's share of one token: half a centroid row (fp16) + its residual bytes
    hf8 c[8];
    uint2 r[NBITS];
    bool valid;
};

template <int NBITS>
__device__ __forceinline__ void s3_issue_rows(s3_raw<NBITS>& raw, const int* cdreg, int t, int64_t off, int len, int i, int h,
                                              const int32_t* __restrict__ codes, const uint8_t* __restrict__ residuals,
                                              const _Float16* __restrict__ cen16) {
    constexpr int PACKED = FLMR_DIM * NBITS / 8, NB = 8 * NBITS;
    const int tok = t * 32 + i;
    raw.valid = tok < len;
    // lanes past the end of the document fetch the document's LAST token instead (finite data, no divergence); their row is
    // zeroed later by a zero normalisation factor, not by 64 per-element selects
    const int tokc = raw.valid ? tok : len - 1;
    int code = 0;
    if (t < 8) {  // tokens < 256 were preloaded: register t>>1 of lane (t&1)*32 + i (wave-uniform register choice)
        const int sel = t >> 1;
        const int reg = sel == 0 ? cdreg[0] : sel == 1 ? cdreg[1] : sel == 2 ? cdreg[2] : cdreg[3];
        code = __shfl(reg, (t & 1) * 32 + i, 64);  // (padding lanes read a preloaded 0: centroid row 0)
    } else {
        code = codes[off + tokc];
    }
    {
        const hf8* pc = reinterpret_cast<const hf8*>(cen16 + (size_t)code * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) raw.c[s] = pc[s];
        const uint2* pr = reinterpret_cast<const uint2*>(residuals + (size_t)(off + tokc) * PACKED + h * NB);
#pragma unroll
        for (int w = 0; w < NBITS; w++) raw.r[w] = pr[w];
    }
}

// GPUF16 = true: the reference's CUDA-branch arithmetic (maxsim_gpu_fp16_kernel's, see there) on this kernel's wave-per-
// document pipeline: fp16 embeddings by s3_decode_f16, ONE product per k-step against the fp16-rounded query, scores rounded
// to fp16, rows past the passage's end excluded (-10000, no zero clamp), fp32 column sum rounded to fp16.  Nq <= 32.
template <int NBITS, bool GPUF16 = false>
__global__ __launch_bounds__(256, 2) void maxsim_f16_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                            const uint8_t* __restrict__ residuals,
                                                            const int64_t* __restrict__ doc_offsets,
                                                            const _Float16* __restrict__ cen16,
                                                            const float* __restrict__ wlut_g, int nqp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    float* wlut = reinterpret_cast<float*>(smem);  // [256 * VPB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    float* colmax = wlut + 256 * VPB + (size_t)wave * nqp;  // this wave's running maxima [nqp]
    const int b = blockIdx.x;
    const int cnt = m.counts[b];
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    constexpr float kFloor = GPUF16 ? -10000.0f : 0.0f;   // a column's running maximum starts at half(-9999) / at zero
    for (int t = tid; t < 256 * VPB; t += 256) wlut[t] = GPUF16 ? flmr_round_f16(wlut_g[t]) : wlut_g[t];
    for (int t = lane; t < nqp; t += 64) colmax[t] = kFloor;
    __syncthreads();

    const int W = gridDim.y * 4, w = blockIdx.y * 4 + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // documents of this wave (<= 64, guaranteed by the launcher)
    if (ndw == 0) return;
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = m.pids[(size_t)b * m.pid_stride + w + lane * W];
        my_off = doc_offsets[my_pid];
        my_len = (int)(doc_offsets[my_pid + 1] - my_off);
    }
    const _Float16* qh_b = m.q_hi + (size_t)b * nqp * FLMR_DIM;
    const _Float16* ql_b = m.q_lo + (size_t)b * nqp * FLMR_DIM;
    hf8 bh[8], bl[8];
    auto load_b = [&](int q0) {
        const hf8* ph = reinterpret_cast<const hf8*>(qh_b + (size_t)(q0 + i) * FLMR_DIM + 64 * h);
        const hf8* pl = reinterpret_cast<const hf8*>(ql_b + (size_t)(q0 + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    };
    const bool single_qt = nqp == 32;
    if (single_qt) load_b(0);

    auto load_codes = [&](int64_t off, int len, int* cd) {
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = (lane + 64 * r < len) ? codes[off + lane + 64 * r] : 0;
    };
    int cd[4], ncd[4];
    {
        const int64_t off0 = ((int64_t)__shfl((int)(uint32_t)((uint64_t)my_off >> 32), 0, 64) << 32) |
                             (uint32_t)__shfl((int)(uint32_t)my_off, 0, 64);
        load_codes(off0, __shfl(my_len, 0, 64), cd);
    }
    s3_raw<NBITS> raw;
    bool have_raw = false;
    float cmx = kFloor;  // single q-tile: this lane's column maximum over its own half's rows of the current document
    for (int j = 0; j < ndw; j++) {
        const int pid = __shfl(my_pid, j, 64);
        const int len = __shfl(my_len, j, 64);
        const int64_t off = ((int64_t)__shfl((int)(uint32_t)((uint64_t)my_off >> 32), j, 64) << 32) |
                            (uint32_t)__shfl((int)(uint32_t)my_off, j, 64);
        int nlen = 0;
        int64_t noff = 0;
        if (j + 1 < ndw) {
            nlen = __shfl(my_len, j + 1, 64);
            noff = ((int64_t)__shfl((int)(uint32_t)((uint64_t)my_off >> 32), j + 1, 64) << 32) |
                   (uint32_t)__shfl((int)(uint32_t)my_off, j + 1, 64);
            load_codes(noff, nlen, ncd);
        }
        const int ntiles = (len + 31) >> 5;
        if (ntiles > 0 && !have_raw) s3_issue_rows<NBITS>(raw, cd, 0, off, len, i, h, codes, residuals, cen16);
        have_raw = false;
        for (int t = 0; t < ntiles; t++) {
            // ---- decompress this lane's half row, normalise, split into fp16 hi/lo (the MFMA A operand) ----
            hf8 ah[8], al[8];
            if constexpr (GPUF16) s3_decode_f16<NBITS>(wlut, [&](int wi) { return (wi & 1) ? raw.r[wi >> 1].y : raw.r[wi >> 1].x; }, raw.c, ah);
            else s3_decode_split<NBITS>(wlut, [&](int wi) { return (wi & 1) ? raw.r[wi >> 1].y : raw.r[wi >> 1].x; }, raw.c, raw.valid, ah, al);
            const int rem = len - t * 32;   // rows of this tile that belong to the passage (GPUF16: the others are excluded)
            // ---- prefetch the next tile's rows (this document's next tile, or the next document's first tile) ----
            if (t + 1 < ntiles) {
                s3_issue_rows<NBITS>(raw, cd, t + 1, off, len, i, h, codes, residuals, cen16);
            } else if (j + 1 < ndw && nlen > 0) {
                s3_issue_rows<NBITS>(raw, ncd, 0, noff, nlen, i, h, codes, residuals, cen16);
                have_raw = true;
            }
            // ---- 32 tokens x 32 query tokens per q-tile ----
            for (int q0 = 0; q0 < qlen; q0 += 32) {
                if (!single_qt) load_b(q0);
                f32x16 acch, accl;
#pragma unroll
                for (int r = 0; r < 16; r++) { acch[r] = 0.0f; accl[r] = 0.0f; }
                float mx;
                if constexpr (GPUF16) {
#pragma unroll
                    for (int s = 0; s < 8; s++) acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acch, 0, 0, 0);
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = flmr_round_f16(acch[r]);   // the half score tensor
                    if (rem < 32) {   // wave-uniform, last tile of a passage: rows past its end are padding (-9999 -> half -10000)
#pragma unroll
                        for (int r = 0; r < 16; r++) v[r] = ((r & 3) + 8 * (r >> 2) + 4 * h) < rem ? v[r] : kFloor;
                    }
#pragma unroll
                    for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r + 8]);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], v[r + 4]);
                    mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                } else {
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acch, 0, 0, 0);
                        accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], accl, 0, 0, 0);
                        accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], accl, 0, 0, 0);
                    }
                    mx = s3_tile_max(acch, accl);  // over this lane's 16 rows
                }
                if (single_qt) {
                    cmx = fmaxf(cmx, mx);  // one column per lane pair: kept in a register until the document ends
                } else {
                    const float mxx = flmr_xhalf_max(mx);
                    const int col = q0 + i;
                    if (h == 0 && col < qlen) colmax[col] = fmaxf(colmax[col], mxx);
                }
            }
        }
        // ---- document done: k-ascending sum of the column maxima, reset for the next document ----
        if (single_qt) {
            const float v = flmr_xhalf_max(cmx);  // the two half-waves hold the maxima over their own rows
            if (h == 0) colmax[i] = v;
            cmx = kFloor;  // segmented_maxsim.cpp:58-59: the running max starts at zero (GPUF16: at the padding value)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const float sc = flmr_seq_sum(colmax, qlen, GPUF16 ? 1 : 0);
            const int dslot = w + j * W;
            if (m.keys) m.keys[(size_t)b * m.key_stride + dslot] = flmr_make_key(sc, pid);
            if (m.scores) m.scores[(size_t)b * m.key_stride + dslot] = sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < nqp; t += 64) colmax[t] = kFloor;
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = ncd[r];
    }
}

// ------------------------------------------------------------------------------------------------
// S3, "centroid + weight" form (default for Nq <= 32 when the centroids are fp16-exact; FLMR_S3_IMPL=cw):
//     d_t . q_k  =  (c_t . q_k  +  w_t . q_k) * inv_t ,     inv_t = 1 / max(||c_t + w_t||, 1e-12)
// The decompressed row d = c + w is never formed.  The centroid row c is fp16-exact, so its fp16 image IS an MFMA A
// operand (c . q = c . q_hi + 2^-11 c . q_lo: two products, the very sequence of stage 0 / stage 2); the bucket weights w take
// only 2^nbits values, so a 256-entry table indexed by the residual BYTE returns the fp16 hi / lo halves of the byte's
// 8 / nbits weights already packed as A-operand fragments (w = w_hi + 2^-11 w_lo; w . q = w_hi . q_hi + 2^-11 (w_hi . q_lo +
// w_lo . q_hi): three products); inv_t is computed once per token when the index is opened (flmr_index.inv_norm, 4 bytes per
// token) and applied to the 32 x 32 score tile.  maxsim_f16_kernel spent 340 VALU operations per tile and lane on
// add / square / scale / convert / split and 24 MFMAs; this form spends ~80 (byte extraction, the hi + lo / 2048 combine, the
// row scaling, the maxima) and 40 MFMAs: the kernel moves from VALU issue to the matrix pipe.
// Rounding differs from the (c + w) / ||c + w|| . q order of the other S3 kernels by fp32-roundoff-class terms (products
// exact, fp32 accumulation; tests/test_split_arithmetic.py bounds it against fp64 like the others); it is NOT bit-identical
// to them.  Document iteration, code preload and the one-tile-ahead row prefetch are maxsim_f16_kernel's.
// LDS: hi table | lo table (256 x 2 * (8 / NBITS) bytes each; NBITS = 8: one table of (hi | lo << 16) words), then per wave
// the column maxima [32] and the tile's 32 row scales.
// ------------------------------------------------------------------------------------------------
template <int NBITS>
struct s3cw_raw {        // one lane's share of one token: half a centroid row (fp16), its residual bytes, the token's 1 / norm
    hf8 c[8];
    uint2 r[NBITS];
    float inv;
    bool valid;
};

template <int NBITS>
__device__ __forceinline__ void s3cw_issue_rows(s3cw_raw<NBITS>& raw, const int* cdreg, int t, int64_t off, int len, int i, int h,
                                                const int32_t* __restrict__ codes, const uint8_t* __restrict__ residuals,
                                                const _Float16* __restrict__ cen16, const float* __restrict__ inv_norm) {
    constexpr int PACKED = FLMR_DIM * NBITS / 8, NB = 8 * NBITS;
    const int tok = t * 32 + i;
    raw.valid = tok < len;
    const int tokc = raw.valid ? tok : len - 1;   // lanes past the end repeat the last token; their row scale is set to zero
    int code = 0;
    if (t < 8) {
        const int sel = t >> 1;
        const int reg = sel == 0 ? cdreg[0] : sel == 1 ? cdreg[1] : sel == 2 ? cdreg[2] : cdreg[3];
        code = __shfl(reg, (t & 1) * 32 + i, 64);
    } else {
        code = codes[off + tokc];
    }
    const hf8* pc = reinterpret_cast<const hf8*>(cen16 + (size_t)code * FLMR_DIM + 64 * h);
#pragma unroll
    for (int s = 0; s < 8; s++) raw.c[s] = pc[s];
    const uint2* pr = reinterpret_cast<const uint2*>(residuals + (size_t)(off + tokc) * PACKED + h * NB);
#pragma unroll
    for (int w = 0; w < NBITS; w++) raw.r[w] = pr[w];
    raw.inv = inv_norm[off + tokc];
}

// residual bytes -> the w_hi / w_lo A-operand fragments of the lane's half row, by table look-up only
// `word(wi)`: wi-th 32-bit word of the lane's 8 * NBITS residual bytes
template <int NBITS, typename WordFn>
__device__ __forceinline__ void s3cw_decode(const char* tab, WordFn word, hf8 (&wh)[8], hf8 (&wl)[8]) {
    constexpr int VPB = 8 / NBITS;
    auto byte_at = [&](int kb) -> uint32_t {   // kb-th of the lane's 8 * NBITS residual bytes
        return (word(kb >> 2) >> (8 * (kb & 3))) & 255u;
    };
#pragma unroll
    for (int s = 0; s < 8; s++) {
        s3u4 hpk, lpk;
        if constexpr (VPB == 8) {            // one byte per k-step: 16-byte entries
            const uint32_t b0 = byte_at(s);
            hpk = reinterpret_cast<const s3u4*>(tab)[b0];
            lpk = reinterpret_cast<const s3u4*>(tab + 4096)[b0];
        } else if constexpr (VPB == 4) {     // two bytes per k-step: 8-byte entries
            const uint32_t b0 = byte_at(2 * s), b1 = byte_at(2 * s + 1);
            const uint2 h0 = reinterpret_cast<const uint2*>(tab)[b0], h1 = reinterpret_cast<const uint2*>(tab)[b1];
            const uint2 l0 = reinterpret_cast<const uint2*>(tab + 2048)[b0], l1 = reinterpret_cast<const uint2*>(tab + 2048)[b1];
            hpk = s3u4{h0.x, h0.y, h1.x, h1.y};
            lpk = s3u4{l0.x, l0.y, l1.x, l1.y};
        } else if constexpr (VPB == 2) {     // four bytes per k-step: 4-byte entries
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t bj = byte_at(4 * s + j);
                hpk[j] = reinterpret_cast<const uint32_t*>(tab)[bj];
                lpk[j] = reinterpret_cast<const uint32_t*>(tab + 1024)[bj];
            }
        } else {                             // eight bytes per k-step: one (hi | lo << 16) word per byte
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e0 = reinterpret_cast<const uint32_t*>(tab)[byte_at(8 * s + 2 * j)];
                const uint32_t e1 = reinterpret_cast<const uint32_t*>(tab)[byte_at(8 * s + 2 * j + 1)];
                hpk[j] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);   // {e0.lo16, e1.lo16}
                lpk[j] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);   // {e0.hi16, e1.hi16}
            }
        }
        wh[s] = __builtin_bit_cast(hf8, hpk);
        wl[s] = __builtin_bit_cast(hf8, lpk);
    }
}

// (hi + lo / 2048) * inv_row, maximum over this lane's 16 rows, zero floor (segmented_maxsim.cpp:58-59); `invs`: the tile's 32
// row scales in LDS (0 for rows past the document's end)
__device__ __forceinline__ float s3cw_tile_max(const f32x16& acch, const f32x16& accl, const float* invs, int h) {
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {   // accumulator registers 4g..4g+3 hold rows 8g + 4h + {0..3}
        const f32x4 iv = *reinterpret_cast<const f32x4*>(invs + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * g + e] = fmaf(accl[4 * g + e], 1.0f / 2048.0f, acch[4 * g + e]) * iv[e];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r + 8]);
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], v[r + 4]);
    return fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), 0.0f);
}

template <int NBITS>
__global__ __launch_bounds__(256, 2) void maxsim_cw_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                           const uint8_t* __restrict__ residuals,
                                                           const int64_t* __restrict__ doc_offsets,
                                                           const _Float16* __restrict__ cen16,
                                                           const uint32_t* __restrict__ wtab_g, const float* __restrict__ inv_norm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    constexpr int TAB_BYTES = NBITS == 8 ? 1024 : 2 * 512 * VPB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    float* colmax = reinterpret_cast<float*>(smem + TAB_BYTES) + (size_t)wave * 64;   // this wave's column maxima [32]
    float* invs = colmax + 32;                                                           // the current tile's row scales [32]
    const int b = blockIdx.x;
    const int cnt = m.counts[b];
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    for (int t = tid; t < TAB_BYTES / 4; t += 256) reinterpret_cast<uint32_t*>(smem)[t] = wtab_g[t];
    if (lane < 32) colmax[lane] = 0.0f;
    __syncthreads();

    const int W = gridDim.y * 4, w = blockIdx.y * 4 + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // documents of this wave (<= 64, guaranteed by the launcher)
    if (ndw == 0) return;
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = m.pids[(size_t)b * m.pid_stride + w + lane * W];
        my_off = doc_offsets[my_pid];
        my_len = (int)(doc_offsets[my_pid + 1] - my_off);
    }
    hf8 bh[8], bl[8];
    {
        const hf8* ph = reinterpret_cast<const hf8*>(m.q_hi + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
        const hf8* pl = reinterpret_cast<const hf8*>(m.q_lo + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    auto load_codes = [&](int64_t off, int len, int* cd) {
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = (lane + 64 * r < len) ? codes[off + lane + 64 * r] : 0;
    };
    auto bcast64 = [&](int64_t v, int src) -> int64_t {
        return ((int64_t)__shfl((int)(uint32_t)((uint64_t)v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    };
    int cd[4], ncd[4];
    load_codes(bcast64(my_off, 0), __shfl(my_len, 0, 64), cd);
    s3cw_raw<NBITS> raw;
    bool have_raw = false;
    float cmx = 0.0f;  // this lane's column maximum over its own half's rows of the current document
    for (int j = 0; j < ndw; j++) {
        const int pid = __shfl(my_pid, j, 64);
        const int len = __shfl(my_len, j, 64);
        const int64_t off = bcast64(my_off, j);
        int nlen = 0;
        int64_t noff = 0;
        if (j + 1 < ndw) {
            nlen = __shfl(my_len, j + 1, 64);
            noff = bcast64(my_off, j + 1);
            load_codes(noff, nlen, ncd);
        }
        const int ntiles = (len + 31) >> 5;
        if (ntiles > 0 && !have_raw) s3cw_issue_rows<NBITS>(raw, cd, 0, off, len, i, h, codes, residuals, cen16, inv_norm);
        have_raw = false;
        for (int t = 0; t < ntiles; t++) {
            // ---- the weights' fragments by table look-up; the row scales to LDS (0 for rows past the document's end) ----
            hf8 wh[8], wl[8];
            s3cw_decode<NBITS>(smem, [&](int wi) { return (wi & 1) ? raw.r[wi >> 1].y : raw.r[wi >> 1].x; }, wh, wl);
            if (h == 0) invs[i] = raw.valid ? raw.inv : 0.0f;
            // ---- centroid part: c . q_hi, c . q_lo (the registers of c are free for the next tile afterwards) ----
            f32x16 acch, accl;
#pragma unroll
            for (int r = 0; r < 16; r++) { acch[r] = 0.0f; accl[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(raw.c[s], bh[s], acch, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(raw.c[s], bl[s], accl, 0, 0, 0);
            }
            // ---- prefetch the next tile's rows (this document's next tile, or the next document's first tile) ----
            if (t + 1 < ntiles) {
                s3cw_issue_rows<NBITS>(raw, cd, t + 1, off, len, i, h, codes, residuals, cen16, inv_norm);
            } else if (j + 1 < ndw && nlen > 0) {
                s3cw_issue_rows<NBITS>(raw, ncd, 0, noff, nlen, i, h, codes, residuals, cen16, inv_norm);
                have_raw = true;
            }
            // ---- weight part: w_hi . q_hi, w_hi . q_lo, w_lo . q_hi ----
#pragma unroll
            for (int s = 0; s < 8; s++) {
                acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], bh[s], acch, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], bl[s], accl, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], bh[s], accl, 0, 0, 0);
            }
            // ---- (hi + lo / 2048) * inv_row, maximum over this lane's 16 rows, zero floor (segmented_maxsim.cpp:58-59) ----
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cmx = fmaxf(cmx, s3cw_tile_max(acch, accl, invs, h));
            __builtin_amdgcn_wave_barrier();   // (the scales are read before the next tile overwrites them)
        }
        // ---- document done: k-ascending sum of the column maxima, reset for the next document ----
        {
            const float vx = flmr_xhalf_max(cmx);  // the two half-waves hold the maxima over their own rows
            if (h == 0) colmax[i] = vx;
            cmx = 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const float sc = flmr_seq_sum(colmax, qlen);
            const int dslot = w + j * W;
            if (m.keys) m.keys[(size_t)b * m.key_stride + dslot] = flmr_make_key(sc, pid);
            if (m.scores) m.scores[(size_t)b * m.key_stride + dslot] = sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = ncd[r];
    }
}

// ------------------------------------------------------------------------------------------------
// S3, fp16-split, centroid rows by LDS-DMA with TWO token tiles in flight (Nq <= 32; default for nbits = 8, FLMR_S3_IMPL=dma).
// maxsim_f16_kernel above keeps one tile's rows in flight in registers (32 VGPRs per lane, requested after the current
// tile has been decompressed): a row gather from the Infinity Cache takes ~2 us, the MFMA phase it overlaps with ~0.3 us,
// and at 250 VGPRs there is no room for a second register buffer -- per tile the wave sits out most of the latency
// (4.5 TB/s of row gathers against the 9.3-9.6 TB/s this chip delivers for 256-byte rows).  Here the rows go straight to LDS
// (global_load_lds_dwordx4: four whole rows per instruction, 16-byte pieces XOR-swizzled by row like stage 2's), two 8 KB
// buffers per wave, requested TWO tiles ahead; the lane's half row is read back (8 x ds_read_b128) when its tile comes up.
// The tiles of all of a wave's documents form one flat sequence, so the pipeline runs across document boundaries.
//
// vmcnt retires in order and the compiler's own wait insertion assumes it sees every outstanding operation; a wait it
// places for one register load would also drain the DMA issued before it.  So every VMEM load inside the loop is issued
// from inline assembly (invisible to that pass) and waited for by hand:
//   * a step issues, in this order: codes of tile g+6 (1 operation: global -> LDS ring, ALWAYS issued -- past the end the
//     last position is repeated), residual bytes of tile g+2 (RL), row DMA of tile g+2 (8);
//   * top of step g: everything older than step g-1's issues must have landed (rows + residual bytes of tile g):
//     vmcnt(9 + RL) when tile g+1 exists (8 + RL for g = 0, where the prologue's second tile comes last), else vmcnt(0);
//   * before the DMA of tile g+2: its codes -- the first operation of step g-4 -- must have landed: younger than them are
//     the rest of that step (RL + 8), three full steps and this step's code + residual requests: vmcnt(36 + 5 RL).
// (The first form requested the codes one step before their rows; that left the codes' HBM latency on every step's path.
// Neither form changes S3's time -- see DESIGN.md section 4.)
// A wait count is safe whenever it does not exceed the number of operations issued after the awaited one; the compiler's
// own stores (keys) only add to that number.  Registers alternate by tile parity, so no value is copied while its load is
// in flight, and each awaited register passes through an empty asm after the wait so its uses cannot be scheduled above it.
// Same arithmetic as maxsim_f16_kernel in the same order: bit-identical scores.
// grid = (nqueries, G), block = 256; dynamic LDS = wlut + 4 * nqp floats + 4 waves x 16 KB.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void s3_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// residual bytes of one lane's half row: 8 * NBITS bytes
template <int NBITS>
struct s3_res {
    u32x4 q[(NBITS + 1) / 2];  // NBITS == 1: only .xy of q[0]
    float inv;                 // "cw" form: the token's 1 / norm (one more load of the tile's residual group)
    __device__ __forceinline__ void issue_inv(const float* p) {
        asm volatile("global_load_dword %0, %1, off" : "=v"(inv) : "v"(p) : "memory");
    }
    __device__ __forceinline__ void issue(const uint8_t* p) {  // asynchronous: wait, then touch(), before any use
        if constexpr (NBITS == 1) {
            // straight into the member's low half (a temporary + copy would read the destination while the load is in flight
            // unless the register allocator happened to coalesce them)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(*reinterpret_cast<u32x2*>(&q[0])) : "v"(p) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[0]) : "v"(p) : "memory");
            if constexpr (NBITS >= 4) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(q[1]) : "v"(p) : "memory");
            if constexpr (NBITS >= 8) {
                asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(q[2]) : "v"(p) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(q[3]) : "v"(p) : "memory");
            }
        }
    }
    __device__ __forceinline__ void touch() {
#pragma unroll
        for (int k = 0; k < (NBITS + 1) / 2; k++) asm volatile("" : "+v"(q[k])::"memory");
        asm volatile("" : "+v"(inv)::"memory");
    }
    __device__ __forceinline__ uint32_t word(int wi) const { return q[wi >> 2][wi & 3]; }
};

// CW = true: the same pipeline with the "centroid + weight" arithmetic of maxsim_cw_kernel (table = wtab16, one more load per
// tile for the token's 1 / norm, 40 MFMAs and no decode arithmetic) -- the default S3 kernel: with the VALU work gone the
// kernel is bound by how many row tiles it keeps in flight, and this pipeline keeps two per wave where the register form
// keeps one.
template <int NBITS, bool CW>
__global__ __launch_bounds__(256, 2) void maxsim_f16_dma_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                                const uint8_t* __restrict__ residuals,
                                                                const int64_t* __restrict__ doc_offsets,
                                                                const _Float16* __restrict__ cen16,
                                                                const float* __restrict__ wlut_g, int nqp,
                                                                const float* __restrict__ inv_norm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS, PACKED = FLMR_DIM * NBITS / 8, NB = 8 * NBITS;
    constexpr int RL = (NBITS == 1 ? 1 : NBITS / 2) + (CW ? 1 : 0);  // residual (+ 1 / norm) load instructions per tile
    constexpr int TABF = CW ? (NBITS == 8 ? 256 : 256 * VPB) : 256 * VPB;   // table size in 32-bit words
    float* wlut = reinterpret_cast<float*>(smem);  // [TABF]: wlut, or (CW) the fp16 hi / lo fragment tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    float* colmax = wlut + TABF + (size_t)wave * (nqp + 32);  // this wave's running maxima [nqp] (nqp == 32 here) + row scales [32]
    float* invs = colmax + nqp;
    char* const rowbuf = smem + ((TABF + 4 * (size_t)(nqp + 32)) * sizeof(float) + 15) / 16 * 16 + (size_t)wave * 16384;
    const uint32_t rowbuf_lds =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)rowbuf);
    char* const ring = smem + ((TABF + 4 * (size_t)(nqp + 32)) * sizeof(float) + 15) / 16 * 16 + 4 * 16384 + (size_t)wave * 2048;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring);
    const int b = blockIdx.x;
    const int cnt = m.counts[b];
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    for (int t = tid; t < TABF; t += 256) wlut[t] = wlut_g[t];
    for (int t = lane; t < nqp; t += 64) colmax[t] = 0.0f;
    __syncthreads();

    const int W = gridDim.y * 4, w = blockIdx.y * 4 + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // documents of this wave (<= 64, guaranteed by the launcher)
    if (ndw == 0) return;
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = m.pids[(size_t)b * m.pid_stride + w + lane * W];
        my_off = doc_offsets[my_pid];
        my_len = (int)(doc_offsets[my_pid + 1] - my_off);
        if (my_len <= 0) {  // an empty passage has no tile: its score is the empty sum (segmented_maxsim.cpp: all maxima 0)
            if (m.keys) m.keys[(size_t)b * m.key_stride + w + lane * W] = flmr_make_key(0.0f, my_pid);
            if (m.scores) m.scores[(size_t)b * m.key_stride + w + lane * W] = 0.0f;
        }
    }
    hf8 bh[8], bl[8];
    {
        const hf8* ph = reinterpret_cast<const hf8*>(m.q_hi + ((size_t)b * nqp + i) * FLMR_DIM + 64 * h);
        const hf8* pl = reinterpret_cast<const hf8*>(m.q_lo + ((size_t)b * nqp + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    const int my_off_lo = (int)(uint32_t)my_off, my_off_hi = (int)(uint32_t)((uint64_t)my_off >> 32);
    // lane j's value, j wave-uniform: v_readlane into an SGPR -- no LDS-crossbar round trip on the step's critical path
    auto bcast = [&](int v, int j) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(j)); };
    // flat tile sequence over (document j, tile t); j == ndw: past the end
    auto normalize = [&](int& j, int& t) {
        while (j < ndw && t >= ((bcast(my_len, j) + 31) >> 5)) { j++; t = 0; }
    };
    auto next_of = [&](int j, int t, int& nj, int& nt) {
        nj = j; nt = t + 1;
        if (j < ndw) normalize(nj, nt); else nj = ndw;
    };
    // array position of this lane's token in tile (j, t); padding lanes take the document's last token
    auto tokpos = [&](int j, int t, bool& valid) {
        const int len = bcast(my_len, j);
        const int64_t off = ((int64_t)bcast(my_off_hi, j) << 32) | (uint32_t)bcast(my_off_lo, j);
        const int tok = t * 32 + i;
        valid = tok < len;
        return off + (valid ? tok : len - 1);
    };
    uint32_t piece_off[8];  // byte offset, inside its row, of the 16-byte piece this lane moves in DMA instruction gq
#pragma unroll
    for (int gq = 0; gq < 8; gq++) piece_off[gq] = (uint32_t)(((lane & 15) ^ ((4 * gq + (lane >> 4)) & 15)) << 4);
    // codes of the tile at array positions `pos` (lane i: row i; lanes 32..63 repeat) -> ring slot g % 8, straight to LDS
    auto issue_codes = [&](int64_t pos, int g) {
        const uint32_t dst = ring_lds + (g & 7) * 256;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(codes + pos), "s"(dst) : "memory", "m0");
    };
    // the codes of tile g's rows that this lane's DMA instructions move (from the ring)
    auto ring_codes = [&](int g, int (&c)[8]) {
        const int* cr = reinterpret_cast<const int*>(ring + (g & 7) * 256) + (lane >> 4);
#pragma unroll
        for (int gq = 0; gq < 8; gq++) c[gq] = cr[4 * gq];
    };
    // 32 rows of tile g -> rowbuf[g & 1]: piece p of row r at position p ^ (r & 15)
    auto dma_rows = [&](int g, const int (&c)[8]) {
#pragma unroll
        for (int gq = 0; gq < 8; gq++) {
            const uint32_t dst = rowbuf_lds + (g & 1) * 8192 + gq * 1024;
            uint32_t voff;
            asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                         : "=&v"(voff)
                         : "s"(cen16), "s"(dst), "v"(c[gq]), "v"(piece_off[gq])
                         : "memory", "m0");
        }
    };
    // tiles g (being consumed), g+1, g+2 of the sequence, and the cursor of the code requests (tile g+6)
    int j0 = 0, t0 = 0, j1, t1, j2, t2, jc, tc;
    normalize(j0, t0);
    next_of(j0, t0, j1, t1);
    next_of(j1, t1, j2, t2);
    jc = j0; tc = t0;
    s3_res<NBITS> rE, rO;              // residual bytes of the even / odd tile in flight
    bool vE = false, vO = false, vtmp;
    int64_t cpos = 0;                  // last valid position of the code cursor (requests past the end repeat it)
    auto next_code_pos = [&]() -> int64_t {
        if (jc < ndw) { cpos = tokpos(jc, tc, vtmp); next_of(jc, tc, jc, tc); }
        return cpos;
    };
    // VMEM issue order:  prologue  C0 .. C5, res0, R0 (8), res1, R1 (8)      Ct = codes of tile t (1 operation, ALWAYS issued:
    //                    step g    C(g+6), res(g+2) (RL), R(g+2) (8)          past the end the last position is repeated)
    // top of step g: R(g) and res(g) must have landed; younger are the operations of step g-1 (the prologue's second tile
    // for g = 0): 1 + RL + 8, or just the code request when tile g+1 does not exist.
    // R(g+2) needs C(g+2), the first operation of step g-4 (or of the prologue): at least 3 + 2 (RL + 8) operations older
    // than the youngest, so the wait at the top of the step has already covered it.
    {
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(bh[s]), "+v"(bl[s])::"memory");
        int64_t p0 = 0, p1 = 0;
        for (int g = 0; g < 6; g++) {
            const int64_t p = next_code_pos();
            if (g == 0) p0 = p;
            if (g == 1) p1 = p;
            issue_codes(p, g);
        }
        s3_wait_vm<4>();  // C0, C1
        asm volatile("" ::: "memory");
        (void)tokpos(j0, t0, vE);
        rE.issue(residuals + (size_t)p0 * PACKED + h * NB);
        if constexpr (CW) rE.issue_inv(inv_norm + p0);
        { int c0[8]; ring_codes(0, c0); dma_rows(0, c0); }
        if (j1 < ndw) {
            (void)tokpos(j1, t1, vO);
            rO.issue(residuals + (size_t)p1 * PACKED + h * NB);
            if constexpr (CW) rO.issue_inv(inv_norm + p1);
            { int c1[8]; ring_codes(1, c1); dma_rows(1, c1); }
        }
    }

    float cmx = 0.0f;
    // one step: consume tile g with residual bytes r / validity v; afterwards r / v belong to tile g+2
    auto step = [&](int g, s3_res<NBITS>& r, bool& v) {
        const int buf = g & 1;
        // ---- tile g's rows and residual bytes: wait, read this lane's half row out of LDS, release the buffer ----
        if (j1 >= ndw) s3_wait_vm<(0)>();
        else if (g == 0) s3_wait_vm<8 + RL>();
        else s3_wait_vm<9 + RL>();
        r.touch();
        // ONE LDS round trip: this tile's rows and the codes of tile g+2 (their request, four steps old, is older than anything
        // the wait above lets through); the decode's table reads only depend on registers and join them
        hf8 c[8];
#pragma unroll
        for (int s = 0; s < 8; s++)
            c[s] = *reinterpret_cast<const hf8*>(rowbuf + buf * 8192 + i * 256 + (((8 * h + s) ^ (i & 15)) << 4));
        int cc[8];
        ring_codes(g + 2, cc);
        // ---- the A operands: decompress + normalise + split, or (CW) the weights' fragments by table look-up ----
        hf8 ah[8], al[8];
        if constexpr (CW) {
            s3cw_decode<NBITS>(reinterpret_cast<const char*>(wlut), [&](int wi) { return r.word(wi); }, ah, al);
            if (h == 0) invs[i] = v ? r.inv : 0.0f;   // the tile's row scales (0 past the document's end), read in the epilogue
        } else {
            s3_decode_split<NBITS>(wlut, [&](int wi) { return r.word(wi); }, c, v, ah, al);
        }
        const bool last_of_doc = (j1 != j0);
        const int pid = bcast(my_pid, j0), dslot = w + j0 * W;
        // ---- keep the pipeline full, in this order: codes of tile g+6, residual bytes of tile g+2, rows of tile g+2 ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row buffer has been read: it may be refilled
        issue_codes(next_code_pos(), g + 6);
        if (j2 < ndw) {
            const int64_t p2 = tokpos(j2, t2, v);
            r.issue(residuals + (size_t)p2 * PACKED + h * NB);
            if constexpr (CW) r.issue_inv(inv_norm + p2);
            dma_rows(g + 2, cc);
        }
        // ---- 32 tokens x 32 query tokens ----
        {
            f32x16 acch, accl;
#pragma unroll
            for (int q = 0; q < 16; q++) { acch[q] = 0.0f; accl[q] = 0.0f; }
            if constexpr (CW) {
#pragma unroll
                for (int s = 0; s < 8; s++) {   // centroid part, then weight part (hi . hi | hi . lo + lo . hi)
                    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[s], bh[s], acch, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[s], bl[s], accl, 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acch, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], accl, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], accl, 0, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                cmx = fmaxf(cmx, s3cw_tile_max(acch, accl, invs, h));
                __builtin_amdgcn_wave_barrier();   // (the scales are read before the next step overwrites them)
            } else {
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acch, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], accl, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], accl, 0, 0, 0);
                }
                cmx = fmaxf(cmx, s3_tile_max(acch, accl));  // this lane's column, over its own half's rows, kept in a register
            }
        }
        if (last_of_doc) {  // k-ascending sum of the column maxima, reset for the next document
            {
                const float v = flmr_xhalf_max(cmx);
                if (h == 0) colmax[i] = v;
                cmx = 0.0f;  // segmented_maxsim.cpp:58-59: the running max starts at zero
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const float sc = flmr_seq_sum(colmax, qlen);
                if (m.keys) m.keys[(size_t)b * m.key_stride + dslot] = flmr_make_key(sc, pid);
                if (m.scores) m.scores[(size_t)b * m.key_stride + dslot] = sc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < nqp; t += 64) colmax[t] = 0.0f;
        }
        // ---- advance the window of tile positions ----
        j0 = j1; t0 = t1; j1 = j2; t1 = t2;
        next_of(j1, t1, j2, t2);
    };
    for (int g = 0; j0 < ndw; g += 2) {
        step(g, rE, vE);
        if (j0 >= ndw) break;
        step(g + 1, rO, vO);
    }
    s3_wait_vm<0>();
}

// ------------------------------------------------------------------------------------------------
// S3 for long queries (Nq > 32, e.g. FLMR's 512 text + 32..320 visual rows): same per-wave document pipeline, but the
// query is walked in CHUNKS of S3_QC tiles of 32 rows that the S3_MQW waves of a workgroup stage once in LDS (fp16 hi/lo,
// rows padded to 272 B so ds_read_b128 across rows is conflict-free).  Loop order: chunk (outer, block barrier) ->
// this wave's documents -> token tiles -> the chunk's q-tiles.  The A operand (decompress + normalise + split) is
// recomputed once per chunk, which costs less than re-reading a 16 KB query tile from L2 per (token tile, q-tile) with
// the latency exposed, as the short-query kernel would.  Column maxima live in a per-wave LDS row of the chunk's width;
// at the end of a document within a chunk they are added, k-ascending, to the document's running sum, so the sum is
// accumulated in exactly the global k order.
// grid = (nqueries, G), block = 64 * S3_MQW; dynamic LDS = wlut + S3_QC * 17408 B + S3_MQW * (32*S3_QC + 64) floats.
// ------------------------------------------------------------------------------------------------
#define S3_QC 8        // q-tiles (of 32 query rows) staged per chunk: the A operand is rebuilt once per chunk
#define S3_MQW 8       // waves per workgroup sharing the staged chunk (one 512-thread workgroup per CU)
#define S3_BROW 136

template <int NBITS>
__global__ __launch_bounds__(64 * S3_MQW) void maxsim_f16_multiq_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                                   const uint8_t* __restrict__ residuals,
                                                                   const int64_t* __restrict__ doc_offsets,
                                                                   const _Float16* __restrict__ cen16,
                                                                   const float* __restrict__ wlut_g, int nqp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS;
    float* wlut = reinterpret_cast<float*>(smem);                                  // [256 * VPB]
    _Float16* bq = reinterpret_cast<_Float16*>(smem + 256 * VPB * sizeof(float));  // [S3_QC][hi|lo][32][S3_BROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    float* colmax = reinterpret_cast<float*>(bq + S3_QC * 2 * 32 * S3_BROW) + (size_t)wave * (32 * S3_QC + 64);
    float* docsum = colmax + 32 * S3_QC;                                          // [64] running sums of this wave's docs
    const int b = blockIdx.x;
    const int cnt = m.counts[b];
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    for (int t = tid; t < 256 * VPB; t += 64 * S3_MQW) wlut[t] = wlut_g[t];
    const int W = gridDim.y * S3_MQW, w = blockIdx.y * S3_MQW + wave;
    const int ndw = cnt > w ? (cnt - w + W - 1) / W : 0;  // <= 64 (launcher); idle waves still join the barriers
    int my_pid = 0, my_len = 0;
    int64_t my_off = 0;
    if (lane < ndw) {
        my_pid = m.pids[(size_t)b * m.pid_stride + w + lane * W];
        my_off = doc_offsets[my_pid];
        my_len = (int)(doc_offsets[my_pid + 1] - my_off);
    }
    docsum[lane] = 0.0f;
    const _Float16* qh_b = m.q_hi + (size_t)b * nqp * FLMR_DIM;
    const _Float16* ql_b = m.q_lo + (size_t)b * nqp * FLMR_DIM;
    auto load_codes = [&](int64_t off, int len, int* cd) {
#pragma unroll
        for (int r = 0; r < 4; r++) cd[r] = (lane + 64 * r < len) ? codes[off + lane + 64 * r] : 0;
    };
    auto off_of = [&](int j) {
        return ((int64_t)__shfl((int)(uint32_t)((uint64_t)my_off >> 32), j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)my_off, j, 64);
    };

    for (int qc0 = 0; qc0 < qlen; qc0 += 32 * S3_QC) {
        const int ntq = ((qlen - qc0 < 32 * S3_QC ? qlen - qc0 : 32 * S3_QC) + 31) >> 5;  // q-tiles in this chunk
        __syncthreads();  // previous chunk fully consumed (also orders the wlut / docsum initialisation)
        for (int e = tid; e < ntq * 1024; e += 64 * S3_MQW) {  // 16-byte pieces: [tile][hi|lo][32 rows][16 pieces]
            const int piece = e & 15, row = (e >> 4) & 31, hl = (e >> 9) & 1, qt = e >> 10;
            const _Float16* src = (hl ? ql_b : qh_b) + (size_t)(qc0 + qt * 32 + row) * FLMR_DIM + piece * 8;  // rows < nqp: zero padded
            *reinterpret_cast<hf8*>(bq + ((qt * 2 + hl) * 32 + row) * S3_BROW + piece * 8) = *reinterpret_cast<const hf8*>(src);
        }
        __syncthreads();
        if (ndw == 0) continue;
        for (int t = lane; t < 32 * S3_QC; t += 64) colmax[t] = 0.0f;
        int cd[4], ncd[4];
        load_codes(off_of(0), __shfl(my_len, 0, 64), cd);
        s3_raw<NBITS> raw;
        bool have_raw = false;
        for (int j = 0; j < ndw; j++) {
            const int len = __shfl(my_len, j, 64);
            const int64_t off = off_of(j);
            int nlen = 0;
            int64_t noff = 0;
            if (j + 1 < ndw) {
                nlen = __shfl(my_len, j + 1, 64);
                noff = off_of(j + 1);
                load_codes(noff, nlen, ncd);
            }
            const int ntiles = (len + 31) >> 5;
            if (ntiles > 0 && !have_raw) s3_issue_rows<NBITS>(raw, cd, 0, off, len, i, h, codes, residuals, cen16);
            have_raw = false;
            for (int t = 0; t < ntiles; t++) {
                hf8 ah[8], al[8];
#ifdef S3M_NO_DECODE   // development probes (profiles/microbench/s3_probe.hip)
#pragma unroll
                for (int s = 0; s < 8; s++) { ah[s] = raw.c[s]; al[s] = raw.c[7 - s]; }
#else
                s3_decode_split<NBITS>(wlut, [&](int wi) { return (wi & 1) ? raw.r[wi >> 1].y : raw.r[wi >> 1].x; }, raw.c, raw.valid, ah, al);
#endif
                if (t + 1 < ntiles) {
                    s3_issue_rows<NBITS>(raw, cd, t + 1, off, len, i, h, codes, residuals, cen16);
                } else if (j + 1 < ndw && nlen > 0) {
                    s3_issue_rows<NBITS>(raw, ncd, 0, noff, nlen, i, h, codes, residuals, cen16);
                    have_raw = true;
                }
                for (int qt = 0; qt < ntq; qt++) {
                    f32x16 acch, accl, accm;  // three independent accumulation chains: hi.hi, hi.lo, lo.hi
#pragma unroll
                    for (int r = 0; r < 16; r++) { acch[r] = 0.0f; accl[r] = 0.0f; accm[r] = 0.0f; }
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        const hf8 bh = *reinterpret_cast<const hf8*>(bq + ((qt * 2 + 0) * 32 + i) * S3_BROW + 64 * h + 8 * s);
                        const hf8 bl = *reinterpret_cast<const hf8*>(bq + ((qt * 2 + 1) * 32 + i) * S3_BROW + 64 * h + 8 * s);
#ifdef S3M_NO_MFMA
                        if (s) { asm volatile("" :: "v"(bh), "v"(bl), "v"(ah[s]), "v"(al[s])); continue; }
#endif
                        acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acch, 0, 0, 0);
                        accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, accl, 0, 0, 0);
                        accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, accm, 0, 0, 0);
                    }
                    float mx = 0.0f;  // segmented_maxsim.cpp:58-59: the running max starts at zero
#ifdef S3M_NO_EPI
                    mx = acch[0] + accl[3] + accm[7];
                    if (mx == 12345.0f) colmax[qt * 32 + i] = mx;
#else
#pragma unroll
                    for (int r = 0; r < 16; r++) mx = fmaxf(mx, fmaf(accl[r] + accm[r], 1.0f / 2048.0f, acch[r]));
                    mx = flmr_xhalf_max(mx);
                    if (h == 0) colmax[qt * 32 + i] = fmaxf(colmax[qt * 32 + i], mx);
#endif
                }
            }
            // document done for this chunk: continue its k-ascending running sum with this chunk's columns
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                float sc = docsum[j];
                const int ncols = (qlen - qc0) < 32 * S3_QC ? (qlen - qc0) : 32 * S3_QC;
                for (int k = 0; k < ncols; k++) sc += colmax[k];
                docsum[j] = sc;
            }
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < 32 * S3_QC; t += 64) colmax[t] = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; r++) cd[r] = ncd[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ndw) {
        const int dslot = w + lane * W;
        const float sc = docsum[lane];
        if (m.keys) m.keys[(size_t)b * m.key_stride + dslot] = flmr_make_key(sc, my_pid);
        if (m.scores) m.scores[(size_t)b * m.key_stride + dslot] = sc;
    }
}

// ------------------------------------------------------------------------------------------------
// S3 on PLANNED tiles (round 5; the default for Nq <= 32, FLMR_S3_IMPL=lean; `cw` selects the kernel above).
// maxsim_f16_dma_kernel<CW> spends ~500 instructions per 32-token tile, of which 40 are MFMAs: the walk over (passage, tile)
// with its v_readlane chains, 64-bit position arithmetic, branches per window slot, the ring of codes in LDS, 32 v_mov to clear
// the accumulators -- a wave issues one instruction every ~5 cycles, so the tile costs what its instruction count says
// (6100 cycles per wave and tile measured, 1280 of them matrix pipe).  Here everything that is wave-uniform is decided
// BEFORE the scoring kernel runs:
//   * s3_plan_kernel (one workgroup per query) writes one 8-byte descriptor per tile of the query's finalists --
//       .x  array position of the tile's first token (32-bit: the launcher requires N < 2^32)
//       .y  first valid row | (end of valid rows) << 6 | (last tile of its passage) << 12 | passage slot << 13
//     -- and the tile ranges of the query's waves (equal shares, cut at passage boundaries).  The LAST tile of a passage with
//     >= 32 tokens is moved back to end at the passage's end: it re-scores up to 31 tokens of the tile before it, which a
//     maximum does not see, and no tile of such a passage has padding rows.  A passage shorter than 32 tokens has one tile
//     whose valid rows are a window [lo, hi) (the tile starts at the passage unless that would read past the arrays' end).
//     Empty passages get their score (0) there; the pid half of every key is written there too.
//   * the scoring wave reads its descriptors with scalar loads two tiles ahead; every VMEM address is an SGPR base (position
//     arithmetic on the scalar unit) plus a per-lane constant; the tile's codes come straight into the registers of the lanes
//     that move the rows (DMA instruction q moves rows {q, 8+q, 16+q, 24+q}: lane group j needs codes 8j..8j+7 = two
//     global_load_dwordx4), no ring in LDS; the accumulators start from the MFMA's zero C operand; the per-passage sums wait
//     in LDS ([column][passage], 16 passages) and are formed k-ascending by 16 lanes at once.
// Arithmetic, accumulation order and the k-ascending sum are maxsim_f16_dma_kernel<CW>'s: bit-identical scores.
// VMEM order per step g: codes(g+3) [2], residual bytes + 1/norm of tile g+2 [RL], rows of tile g+2 [8].  Top of step g needs
// rows(g), residuals(g), codes(g+2): everything but the youngest 8 + RL operations (rows / residuals of tile g+1).  Requests
// past the wave's last tile repeat the last tile (static counts, no branch); their data is never used.
// grid = (nqueries, G), block = 256; LDS = table + 4 x (32 + 32 * 17 + 16) words + 4 x 16 KB.
// ------------------------------------------------------------------------------------------------
#define S3L_FLUSH 16                           // passages whose column maxima wait in LDS for their sums
#define S3L_CBS 17                             // row stride (floats) of that [column][passage] matrix
#define S3L_WAVE_WORDS (32 + 32 * S3L_CBS + S3L_FLUSH)
#define S3L_MAX_DOCS 8192                      // finalists per query the plan kernel handles (its LDS prefix table)

__global__ __launch_bounds__(256) void s3_plan_kernel(flmr_maxsim_args m, const int64_t* __restrict__ doc_offsets, int64_t N, int W) {
    __shared__ int base_lds[S3L_MAX_DOCS + 1];
    __shared__ int scan_lds[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cnt = m.counts[b] < m.max_count ? m.counts[b] : m.max_count;
    uint2* desc = m.plan_desc + (size_t)b * m.plan_stride;
    int running = 0;
    for (int d0 = 0; d0 < cnt; d0 += 256) {
        const int d = d0 + tid;
        int len = 0, pid = 0;
        int64_t off = 0;
        if (d < cnt) {
            pid = m.pids[(size_t)b * m.pid_stride + d];
            off = doc_offsets[pid];
            len = (int)(doc_offsets[pid + 1] - off);
            if (len < 0) len = 0;
        }
        const int nt = (len + 31) >> 5;
        int total;
        const int base = running + flmr_block_exclusive_scan(nt, scan_lds, &total);
        running += total;
        if (d < cnt) {
            base_lds[d] = base;
            const size_t ko = (size_t)b * m.key_stride + d;
            if (len == 0) {  // the empty sum (segmented_maxsim.cpp: every column maximum stays 0)
                if (m.keys) m.keys[ko] = flmr_make_key(0.0f, pid);
                if (m.scores) m.scores[ko] = 0.0f;
            } else if (m.keys) {
                reinterpret_cast<uint32_t*>(m.keys)[2 * ko] = (uint32_t)pid;   // the score half comes from the scoring kernel
            }
            for (int t = 0; t < nt; t++) {
                int64_t pos = off + 32 * (int64_t)t;
                int lo = 0, hi = 32;
                if (t == nt - 1) {
                    if (len >= 32) pos = off + len - 32;
                    else {
                        if (pos + 32 > N) pos = N - 32;
                        lo = (int)(off - pos); hi = lo + len;
                    }
                }
                desc[base + t] = make_uint2((uint32_t)pos, (uint32_t)(lo | (hi << 6) | ((t == nt - 1) << 12) | (d << 13)));
            }
        }
    }
    if (tid == 0) base_lds[cnt] = running;
    __syncthreads();
    int* wb = m.plan_wbeg + (size_t)b * m.plan_wcap;
    for (int w = tid; w <= W; w += 256) {
        const int target = (int)(((int64_t)w * running) / W);
        int lo = 0, hi = cnt;          // smallest d in [0, cnt] with base_lds[d] >= target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (base_lds[mid] >= target) hi = mid; else lo = mid + 1;
        }
        wb[w] = w == W ? running : base_lds[lo];
    }
}

// (residual byte kb of the lane's words) << SH in ONE instruction: the SDWA form of v_lshlrev selects the byte of its operand
template <int BYTE>
__device__ __forceinline__ uint32_t s3l_byte_shl(uint32_t word, uint32_t sh) {
    uint32_t a;
    if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(sh), "v"(word));
    else if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(sh), "v"(word));
    else if constexpr (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(sh), "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(sh), "v"(word));
    return a;
}

// s3cw_decode with one VALU instruction per residual byte (its LDS byte address) and the lo table S3L_LO_PAD bytes past the
// hi table's end: at that distance the compiler cannot fuse a byte's hi and lo look-ups into one ds_read2st64_b64, whose four
// result registers (hi pair | lo pair) then have to be moved apart -- 48 v_mov per tile -- because an MFMA operand is four
// CONSECUTIVE registers of one kind.  `sh`: a register holding log2 (bytes per table entry).
#define S3L_LO_PAD 64
template <int NBITS, typename WordFn>
__device__ __forceinline__ void s3l_decode(const char* tab, WordFn word, uint32_t sh, hf8 (&wh)[8], hf8 (&wl)[8]) {
    constexpr int VPB = 8 / NBITS;
    constexpr int LO = (NBITS == 8 ? 0 : 512 * VPB + S3L_LO_PAD);   // byte offset of the lo table
    auto addr = [&](auto kbc) -> uint32_t {   // LDS byte offset of the entry of the lane's kb-th residual byte
        constexpr int kb = decltype(kbc)::value;
        return s3l_byte_shl<kb & 3>(word(kb >> 2), sh);
    };
#define S3L_KB(x) std::integral_constant<int, (x)>{}
    auto step_s = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        s3u4 hpk, lpk;
        if constexpr (VPB == 8) {
            const uint32_t a0 = addr(S3L_KB(s));
            hpk = *reinterpret_cast<const s3u4*>(tab + a0);
            lpk = *reinterpret_cast<const s3u4*>(tab + LO + a0);
        } else if constexpr (VPB == 4) {
            const uint32_t a0 = addr(S3L_KB(2 * s)), a1 = addr(S3L_KB(2 * s + 1));
            const uint2 h0 = *reinterpret_cast<const uint2*>(tab + a0), h1 = *reinterpret_cast<const uint2*>(tab + a1);
            const uint2 l0 = *reinterpret_cast<const uint2*>(tab + LO + a0), l1 = *reinterpret_cast<const uint2*>(tab + LO + a1);
            hpk = s3u4{h0.x, h0.y, h1.x, h1.y};
            lpk = s3u4{l0.x, l0.y, l1.x, l1.y};
        } else if constexpr (VPB == 2) {
            const uint32_t a0 = addr(S3L_KB(4 * s)), a1 = addr(S3L_KB(4 * s + 1)), a2 = addr(S3L_KB(4 * s + 2)), a3 = addr(S3L_KB(4 * s + 3));
            hpk = s3u4{*reinterpret_cast<const uint32_t*>(tab + a0), *reinterpret_cast<const uint32_t*>(tab + a1),
                       *reinterpret_cast<const uint32_t*>(tab + a2), *reinterpret_cast<const uint32_t*>(tab + a3)};
            lpk = s3u4{*reinterpret_cast<const uint32_t*>(tab + LO + a0), *reinterpret_cast<const uint32_t*>(tab + LO + a1),
                       *reinterpret_cast<const uint32_t*>(tab + LO + a2), *reinterpret_cast<const uint32_t*>(tab + LO + a3)};
        } else {   // one (hi | lo << 16) word per byte
            uint32_t e[8];
            e[0] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s)));     e[1] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 1)));
            e[2] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 2))); e[3] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 3)));
            e[4] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 4))); e[5] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 5)));
            e[6] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 6))); e[7] = *reinterpret_cast<const uint32_t*>(tab + addr(S3L_KB(8 * s + 7)));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                hpk[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);   // {e0.lo16, e1.lo16}
                lpk[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);   // {e0.hi16, e1.hi16}
            }
        }
        wh[s] = __builtin_bit_cast(hf8, hpk);
        wl[s] = __builtin_bit_cast(hf8, lpk);
    };
    step_s(S3L_KB(0)); step_s(S3L_KB(1)); step_s(S3L_KB(2)); step_s(S3L_KB(3));
    step_s(S3L_KB(4)); step_s(S3L_KB(5)); step_s(S3L_KB(6)); step_s(S3L_KB(7));
#undef S3L_KB
}

template <int NBITS>
struct s3l_res {      // one lane's residual bytes (8 * NBITS) and 1 / norm of one tile, loaded from inline assembly
    u32x4 q[(NBITS + 1) / 2];
    float inv;
    __device__ __forceinline__ void issue(uint32_t voff, const uint8_t* sbase, uint32_t ivoff, const float* ibase) {
        if constexpr (NBITS == 1) {
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(*reinterpret_cast<u32x2*>(&q[0])) : "v"(voff), "s"(sbase) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[0]) : "v"(voff), "s"(sbase) : "memory");
            if constexpr (NBITS >= 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(q[1]) : "v"(voff), "s"(sbase) : "memory");
            if constexpr (NBITS >= 8) {
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(q[2]) : "v"(voff), "s"(sbase) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(q[3]) : "v"(voff), "s"(sbase) : "memory");
            }
        }
        asm volatile("global_load_dword %0, %1, %2" : "=v"(inv) : "v"(ivoff), "s"(ibase) : "memory");
    }
    __device__ __forceinline__ void touch() {
#pragma unroll
        for (int k = 0; k < (NBITS + 1) / 2; k++) asm volatile("" : "+v"(q[k])::"memory");
        asm volatile("" : "+v"(inv)::"memory");
    }
    __device__ __forceinline__ uint32_t word(int wi) const { return q[wi >> 2][wi & 3]; }
};

struct s3l_codes {    // the eight codes of the rows this lane's DMA instructions move
    u32x4 lo, hi;
    __device__ __forceinline__ void issue(uint32_t voff, const int32_t* sbase) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(lo) : "v"(voff), "s"(sbase) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(hi) : "v"(voff), "s"(sbase) : "memory");
    }
    __device__ __forceinline__ void touch() { asm volatile("" : "+v"(lo), "+v"(hi)::"memory"); }
    __device__ __forceinline__ uint32_t at(int gq) const { return gq < 4 ? lo[gq] : hi[gq - 4]; }
};

template <int NBITS>
__global__ __launch_bounds__(256, 2) void maxsim_lean_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                             const uint8_t* __restrict__ residuals,
                                                             const _Float16* __restrict__ cen16,
                                                             const uint32_t* __restrict__ wtab_g,
                                                             const float* __restrict__ inv_norm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS, PACKED = FLMR_DIM * NBITS / 8, NB = 8 * NBITS;
    constexpr int RL = (NBITS == 1 ? 1 : NBITS / 2) + 1;                  // residual + 1 / norm load instructions per tile
    constexpr int TABW = NBITS == 8 ? 256 : 256 * VPB;                    // fragment tables (wtab16), 32-bit words
    constexpr int HIW = NBITS == 8 ? 256 : 128 * VPB;                     // words of the hi table (NBITS = 8: the one combined table)
    constexpr int TABL = NBITS == 8 ? 256 : TABW + S3L_LO_PAD / 4;        // words the tables take in LDS (lo table S3L_LO_PAD bytes further)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    // the fragment tables are a STATIC array: its LDS address is a constant that folds into the look-ups' offset fields (with the
    // dynamic carve-out's base the compiler adds a register holding "0" to each of a tile's 16 addresses)
    __shared__ __attribute__((aligned(16))) uint32_t tab_s[TABL];
    float* const invs = reinterpret_cast<float*>(smem) + (size_t)wave * S3L_WAVE_WORDS;          // the tile's 32 row scales
    float* const colbuf = invs + 32;                                                             // [32 columns][S3L_CBS]
    int* const slots = reinterpret_cast<int*>(colbuf + 32 * S3L_CBS);                            // passage slot of each waiting sum
    char* const rowbuf = smem + (4 * (size_t)S3L_WAVE_WORDS * 4 + 15) / 16 * 16 + (size_t)wave * 16384;
    const uint32_t rowbuf_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)rowbuf);
    for (int t = tid; t < TABW; t += 256) tab_s[t < HIW ? t : t + (TABL - TABW)] = wtab_g[t];
    __syncthreads();

    const int b = blockIdx.x;
    const int W = gridDim.y * 4, w = blockIdx.y * 4 + wave;
    const int32_t* wb = m.plan_wbeg + (size_t)b * m.plan_wcap;
    const int tb = __builtin_amdgcn_readfirstlane(wb[w]), te = __builtin_amdgcn_readfirstlane(wb[w + 1]);
    (void)W;
    if (tb >= te) return;
    const uint2* dq = m.plan_desc + (size_t)b * m.plan_stride;
    // descriptors by SCALAR loads written as instructions (the compiler, unable to prove the buffer unwritten behind the asm
    // statements' memory clobbers, makes them vector loads -- whose wait would drain the row pipeline); a load's result may be
    // used after the next `s_waitcnt lgkmcnt(0)` and its touch()
    auto ld_desc = [&](int t, u32x2& d) {
        const int tc = t < te ? t : te - 1;
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(d) : "s"(dq + tc) : "memory");
    };
    auto touch_desc = [&](u32x2& d) { asm volatile("" : "+s"(d)::"memory"); };
    hf8 bh[8], bl[8];
    {
        const hf8* ph = reinterpret_cast<const hf8*>(m.q_hi + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
        const hf8* pl = reinterpret_cast<const hf8*>(m.q_lo + ((size_t)b * 32 + i) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) { bh[s] = ph[s]; bl[s] = pl[s]; }
    }
    // per-lane constants
    const uint32_t res_voff = (uint32_t)(i * PACKED + h * NB), inv_voff = (uint32_t)(i * 4), code_voff = (uint32_t)((lane >> 4) * 32);
    const uint32_t piece_base = (uint32_t)(((lane & 15) ^ (8 * ((lane >> 4) & 1))) << 4);   // row 8j + q: swizzle (8j + q) & 15
    const uint32_t rd_base = (uint32_t)((4 * (i & 7) + (i >> 3)) * 256 + (((8 * h) ^ (i & 15)) << 4));   // row i sits in slot 4 (i & 7) + (i >> 3)

    auto dma_rows = [&](int par, const s3l_codes& c) {
#ifdef S3L_NO_DMA   // development probe (profiles/microbench/s3_probe.hip): the step without its row gather
        return;
#endif
#pragma unroll
        for (int gq = 0; gq < 8; gq++) {
            const uint32_t dst = rowbuf_lds + par * 8192 + gq * 1024;
            const uint32_t po = piece_base ^ (uint32_t)(gq << 4);
            uint32_t voff;
            asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                         : "=&v"(voff)
                         : "s"(cen16), "s"(dst), "v"(c.at(gq)), "v"(po)
                         : "memory", "m0");
        }
    };
    auto issue_tile = [&](s3l_res<NBITS>& r, uint32_t pos) {
        r.issue(res_voff, residuals + (size_t)pos * PACKED, inv_voff, inv_norm + pos);
    };

    s3l_res<NBITS> rE, rO;
    s3l_codes cE, cO;
    u32x2 d0, d1, d2;
    ld_desc(tb, d0); ld_desc(tb + 1, d1); ld_desc(tb + 2, d2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    touch_desc(d0); touch_desc(d1); touch_desc(d2);
    const uint32_t sh_entry = NBITS == 1 ? 4 : NBITS == 2 ? 3 : 2;   // log2 (bytes per table entry), in a register for the SDWA shifts
    {
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(bh[s]), "+v"(bl[s])::"memory");
        cE.issue(code_voff, codes + d0.x);
        cO.issue(code_voff, codes + d1.x);
        s3_wait_vm<0>();
        cE.touch(); cO.touch();
        issue_tile(rE, d0.x);
        dma_rows(0, cE);
        cE.issue(code_voff, codes + d2.x);
        issue_tile(rO, d1.x);
        dma_rows(1, cO);
    }
    float cmx = 0.0f;
    int nd = 0;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nd) {
            float x[32];
#pragma unroll
            for (int k = 0; k < 32; k++) x[k] = colbuf[k * S3L_CBS + lane];
            float sc = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; k++) sc += x[k];   // k-ascending (columns >= q_len hold +0)
            const size_t ko = (size_t)b * m.key_stride + slots[lane];
            if (m.keys) reinterpret_cast<uint32_t*>(m.keys)[2 * ko + 1] = flmr_f2ord(sc);
            if (m.scores) m.scores[ko] = sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // one step: consume tile g (descriptor d0; rows in buffer `par`, residual bytes in r, whose registers take tile g+2's;
    // cn holds the codes of tile g+2, cf takes those of tile g+3)
    auto step = [&](int g, int par, s3l_res<NBITS>& r, s3l_codes& cn, s3l_codes& cf) {
        u32x2 d3;
        ld_desc(g + 3, d3);
#ifdef S3L_NO_DMA
        s3_wait_vm<RL>();
#else
        s3_wait_vm<8 + RL>();
#endif
        r.touch(); cn.touch();
        hf8 c[8];
#pragma unroll
        for (int s = 0; s < 8; s++) c[s] = *reinterpret_cast<const hf8*>(rowbuf + par * 8192 + (rd_base ^ (uint32_t)(s << 4)));
        hf8 ah[8], al[8];
#ifdef S3L_NO_DECODE
#pragma unroll
        for (int s = 0; s < 8; s++) { ah[s] = c[s]; al[s] = c[7 - s]; }
        asm volatile("" :: "v"(r.word(0)));
#else
        s3l_decode<NBITS>(reinterpret_cast<const char*>(tab_s), [&](int wi) { return r.word(wi); }, sh_entry, ah, al);
#endif
        {
            const int lo = (int)(d0.y & 63u), hi = (int)((d0.y >> 6) & 63u);
            if (h == 0) invs[i] = (i >= lo && i < hi) ? r.inv : 0.0f;
        }
        // the row buffer has been read (and the table look-ups are in): nothing may be scheduled across this point -- an LDS read
        // that slipped below it could see the next tile's rows
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        touch_desc(d3);
        cf.issue(code_voff, codes + d3.x);
        issue_tile(r, d2.x);
        dma_rows(par, cn);
        f32x16 acch, accl;
        {
            const f32x16 z = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], bh[0], z, 0, 0, 0);
            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[0], bl[0], z, 0, 0, 0);
        }
#ifndef S3L_NO_MFMA
#pragma unroll
        for (int s = 1; s < 8; s++) {
            acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[s], bh[s], acch, 0, 0, 0);
            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[s], bl[s], accl, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 8; s++) {
            acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acch, 0, 0, 0);
            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], accl, 0, 0, 0);
            accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], accl, 0, 0, 0);
        }
#else
#pragma unroll
        for (int s = 1; s < 8; s++) asm volatile("" :: "v"(c[s]), "v"(ah[s]), "v"(al[s]));
#endif
#ifdef S3L_NO_EPI
        cmx = fmaxf(cmx, acch[0] + accl[5]);
#else
        cmx = fmaxf(cmx, s3cw_tile_max(acch, accl, invs, h));
#endif
        __builtin_amdgcn_wave_barrier();   // (the scales are read before the next step overwrites them)
        if ((d0.y >> 12) & 1u) {   // last tile of its passage
            const float vx = flmr_xhalf_max(cmx);
            cmx = 0.0f;  // segmented_maxsim.cpp:58-59: the running max starts at zero
            if (h == 0) colbuf[i * S3L_CBS + nd] = vx;
            if (lane == 0) slots[nd] = (int)(d0.y >> 13);
            nd++;
            if (nd == S3L_FLUSH) { flush(); nd = 0; }
        }
        d0 = d1; d1 = d2; d2 = d3;
    };
    for (int g = tb; g < te; g += 2) {
        step(g, 0, rE, cE, cO);
        if (g + 1 >= te) break;
        step(g + 1, 1, rO, cO, cE);
    }
    if (nd > 0) flush();
    s3_wait_vm<0>();
}

template <int NBITS>
static int launch_maxsim_lean_t(const flmr_maxsim_args& a, hipStream_t st, int G) {
    const flmr_index* ix = a.ix;
    const size_t lds = (4 * (size_t)S3L_WAVE_WORDS * 4 + 15) / 16 * 16 + 4 * 16384;   // (+ the static fragment tables)
    hipLaunchKernelGGL(s3_plan_kernel, dim3(a.nqueries), dim3(256), 0, st, a, ix->doc_offsets, ix->N, 4 * G);
    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_lean_kernel<NBITS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(maxsim_lean_kernel<NBITS>, dim3(a.nqueries, G), dim3(256), lds, st, a, ix->codes, ix->residuals, ix->centroids_f16,
                       ix->wtab16, ix->inv_norm);
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// S3 for long queries, QUERY-STATIONARY (round 5; the default for Nq >= S3Q_MIN_NQP, FLMR_S3_IMPL=qs; `mq` selects the kernel above).
// FLMR concatenates 32 text rows with 9 x 32 visual rows (Nq = 320, src/models/retriever/FLMR.py:73-99), PreFLMR goes to 832.
// maxsim_f16_multiq_kernel keeps a token tile's A operand in registers and walks the query through LDS in chunks of 256
// columns: every chunk re-gathers (register gathers: 512 tag look-ups per tile) and re-decodes every tile, each (tile, q-tile)
// pays an LDS read-modify-write of the column maxima and a cross-half exchange, and its k-steps wait for their fragment
// reads -- 39 % of the matrix pipe at Nq = 832 (profiles/r05/s3_multiq_ablation.txt).  Here the roles are swapped:
//   * the QUERY sits in registers: each of a workgroup's 8 waves holds the hi / lo B fragments of up to two 32-row q-tiles
//     (128 VGPRs), 16 q-tiles = 512 columns per workgroup "pass" (Nq = 320: one pass, 832: passes of 16 + 10 q-tiles);
//   * the TOKEN TILES stream through LDS, gathered and decoded ONCE per pass: wave w issues DMA instruction w of every tile (rows
//     {w, 8+w, 16+w, 24+w} -> a raw-row buffer, the planned-tile kernel's layout); waves 0-3 -- one per SIMD -- decode k-steps w
//     and w + 4 of every tile (16 of the 128 dims of all 32 rows per k-step: one ds_read_b128 of the raw rows, NBITS table look-ups
//     (all of a stage's in flight together), (w + c) * 1/norm with the index's inv_norm, split into fp16 hi / lo, two
//     ds_write_b128 into the tile's ring slot in fragment order) BEFORE their MFMAs, while their SIMD partners (waves 4-7) only
//     gather and consume: a SIMD's matrix pipe runs the partner's MFMAs during the decode and the decoder's afterwards; all 8
//     waves read every slot (16 ds_read_b128 per tile and wave feed 48 MFMAs);
//   * rounds of 3 tiles, ring and raw buffers double-buffered (96 + 48 KB).  Round r, every wave: decode its k-step of the
//     tiles of round r+1 -> ring[(r + 1) & 1]; request rows / residual bytes / scales of the tiles of round r+2 (codes asked for
//     a round earlier) and the codes of round r+3; consume ring[r & 1]; vmcnt(0) (everything it waits for is a round old); ONE
//     block barrier.  Tile descriptors (s3_plan_kernel) ride in SGPRs, four rounds deep;
//   * ONE accumulator per q-tile: both operands are scaled by L = 64 before the fp16 split, x = hi + lo with lo = fp16(64 x -
//     hi) -- not the 2048-fold residual the other kernels keep -- so hi.hi, hi.lo and lo.hi are products at ONE scale and add
//     into the same registers (the lo products first, over all k-steps, then hi.hi: the small terms are summed before they meet
//     the large ones).  No hi + lo / 2048 combine, half the accumulator registers, an epilogue of 8 v_max3 per q-tile; the
//     column maximum is rescaled once per passage.  The query's scale is per query (s3q_scale_kernel: 64 unless an entry exceeds 256
//     in magnitude), so its images never overflow; rows of an index are unit vectors.  lo is a normal fp16 number for |x| >= 2^-8 and loses nothing that matters below (absolute 2^-31 of x);
//   * column maxima stay in registers until their passage ends (all waves see the same boundaries), then go to a
//     [query][finalist][column] array; s3_colsum_kernel forms the k-ascending sums.
// Scores differ from maxsim_f16_multiq_kernel's by fp32 roundoff (same decode up to the scaling, other accumulation order).
// grid = (nqueries, passes * Y), block = 512; LDS = 144 KB + the static weight table.
// ------------------------------------------------------------------------------------------------
#define S3Q_R 3
#define S3Q_SLOT 16384
#define S3Q_RAW 8192
#define S3Q_SCALE 64.0f
#ifndef S3Q_MIN_NQP
#define S3Q_MIN_NQP 288
#endif

// Per-query operand scale of this kernel's query images: L_b = 64 for queries whose entries stay below 256 in magnitude (every
// normalised query), else the largest power of two with L_b * max|q| <= 16384 -- the fp16 images never overflow whatever the
// caller hands in (FLMR's un-normalised visual rows included).  qscale[b] = 1 / (64 L_b): what a column maximum is multiplied by.
__global__ __launch_bounds__(256) void s3q_scale_kernel(const float* Q, const int32_t* q_lens, int nq, float* qscale) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int qlen = q_lens ? (q_lens[b] < nq ? q_lens[b] : nq) : nq;
    float mx = 0.0f;
    for (int e = threadIdx.x; e < qlen * FLMR_DIM; e += 256) mx = fmaxf(mx, fabsf(Q[(size_t)b * nq * FLMR_DIM + e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float L = S3Q_SCALE;
        if (!(mx * L <= 16384.0f)) {   // (also taken for inf / NaN input: the smallest scale)
            L = 1.0f;
            if (mx < 3.0e38f && mx > 16384.0f) { int ex; (void)frexpf(16384.0f / mx, &ex); L = ldexpf(1.0f, ex - 1); }
            else if (mx <= 16384.0f) { int ex; (void)frexpf(16384.0f / mx, &ex); L = ldexpf(1.0f, ex - 1); }
            else L = ldexpf(1.0f, -100);
        }
        qscale[b] = 1.0f / (S3Q_SCALE * L);
    }
}

// the query images of this kernel: hi = fp16(L q), lo = fp16(L q - hi); rows >= q_len (and the padding to a multiple of 32) zero
__global__ __launch_bounds__(256) void s3q_split_q(const float* Q, const int32_t* q_lens, int nq, int nqp, _Float16* q_hi, _Float16* q_lo,
                                                   const float* qscale) {
    const int b = blockIdx.y;
    const int qlen = q_lens ? q_lens[b] : nq;
    const float L = 1.0f / (qscale[b] * S3Q_SCALE);   // (powers of two: exact)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nqp * FLMR_DIM; e += gridDim.x * blockDim.x) {
        const int row = e / FLMR_DIM;
        float v = 0.0f;
        if (row < qlen && row < nq) v = Q[((size_t)b * nq + row) * FLMR_DIM + (e % FLMR_DIM)] * L;
        const _Float16 hi = (_Float16)v;
        q_hi[(size_t)b * nqp * FLMR_DIM + e] = hi;
        q_lo[(size_t)b * nqp * FLMR_DIM + e] = (_Float16)(v - (float)hi);
    }
}

// 8 dims of one row (the lane's share of a k-step): weights by table, + centroid, * (64 / norm), split -> the hi / lo fragments
// `bytes`: the NBITS residual bytes of those dims, low byte first.  In two parts so that a stage can issue the table look-ups of
// ALL its units before the arithmetic of the first (one LDS round trip per stage instead of one per unit).
template <int NBITS>
__device__ __forceinline__ void s3q_unit_weights(const float* wlut, const uint32_t (&bytes)[2], float (&w)[8]) {
    constexpr int VPB = 8 / NBITS;
#pragma unroll
    for (int bb = 0; bb < NBITS; bb++) {
        const uint32_t byte = (bytes[bb >> 2] >> (8 * (bb & 3))) & 255u;
        float wv[VPB];
        s3_lut<VPB>(wlut, byte, wv);
#pragma unroll
        for (int l = 0; l < VPB; l++) w[bb * VPB + l] = wv[l];
    }
}
__device__ __forceinline__ void s3q_unit_split(const float (&w)[8], const hf8& c, float inv64, hf8& ah, hf8& al) {
    float x[8];
#pragma unroll
    for (int dd = 0; dd < 8; dd++) {
        const uint32_t cpk = __builtin_bit_cast(s3u4, c)[dd >> 1];
        x[dd] = (dd & 1) ? s3_add_f16<1>(cpk, w[dd]) : s3_add_f16<0>(cpk, w[dd]);
    }
    const float m1 = -1.0f;
    s3u4 hpk, lpk;
#pragma unroll
    for (int pr = 0; pr < 4; pr++) {
        s3v2 v = {x[2 * pr], x[2 * pr + 1]};
        v *= inv64;
        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, s3h2));
        uint32_t lo;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(m1), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(m1), "v"(v[1]));
        hpk[pr] = hi;
        lpk[pr] = lo;
    }
    ah = __builtin_bit_cast(hf8, hpk);
    al = __builtin_bit_cast(hf8, lpk);
}

// max(running maximum, a lane's 16 rows, 0) (segmented_maxsim.cpp:58-59: the maximum starts at zero) in nine v_max3 written as
// instructions: fmaxf on MFMA results first canonicalises every operand (ten more VALU operations per tile and q-tile, in a
// kernel whose time is the sum of everything its SIMD issues)
__device__ __forceinline__ float s3q_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float s3q_tile_max(float run, f32x16 acc) {
    // the hazard recogniser does not look inside asm statements: the wait states between the last MFMA writing `acc` and its first
    // VALU read (11 for an 8-pass MFMA) are ours -- one statement that owns the accumulator, so every read below depends on it
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(acc));
    const float a = s3q_max3(acc[0], acc[1], acc[2]), b = s3q_max3(acc[3], acc[4], acc[5]), c = s3q_max3(acc[6], acc[7], acc[8]);
    const float d = s3q_max3(acc[9], acc[10], acc[11]), e = s3q_max3(acc[12], acc[13], acc[14]);
    const float f = s3q_max3(acc[15], a, b), g = s3q_max3(c, d, e);
    float r;
    asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(f), "v"(g));
    return s3q_max3(r, run, r);
}

#ifdef S3Q_PROFILE   // development only: s_memtime ticks per round phase, summed per wave (index wave * 8 + phase), read by s3_probe
__device__ unsigned long long s3q_prof[64];
#define S3Q_STAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pt[k] += now_ - plast; plast = now_; } while (0)
#else
#define S3Q_STAMP(k) do { } while (0)
#endif

// NW waves per workgroup (each owns k-steps {w, w + NW, ..} of every tile and up to two q-tiles: 2 NW q-tiles per pass), R tiles
// per round.  <8, 3> (the default): one workgroup per CU, the two waves of a SIMD staggered by hand.  <4, 1>: TWO workgroups
// per CU (48 KB of LDS each), one wave per SIMD each -- their rounds and barriers are independent, so one workgroup's MFMAs fill
// the matrix pipe while the other decodes, gathers or waits, without any hand-made stagger; the price is a decode per 8 instead
// of 16 q-tiles, and it is the slower form (Nq = 832: 5.59 vs 5.03 ms per 256 queries: what a SIMD issues beside its MFMAs adds to
// their time instead of hiding behind it, and this form issues more per MFMA).
template <int NBITS, int NW, int R>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void maxsim_qs_kernel(flmr_maxsim_args m, const int32_t* __restrict__ codes,
                                                        const uint8_t* __restrict__ residuals,
                                                        const _Float16* __restrict__ cen16, const float* __restrict__ wlut_g,
                                                        const float* __restrict__ inv_norm, int nqp, int npass, int per_pass) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VPB = 8 / NBITS, PACKED = FLMR_DIM * NBITS / 8, NB = 8 * NBITS;
    constexpr int KS = 8 / NW;             // DMA instructions of a tile this wave issues
    constexpr int U = R * KS;
    constexpr int ND = 4;                  // decoding waves: 0 .. 3, one per SIMD -- in the 8-wave form each decodes beside a partner
                                           // (wave + 4) that only consumes: the partner's MFMAs run while it decodes, its own afterwards
    constexpr int KD = 8 / ND;             // k-steps of a tile a decoding wave handles: wave, wave + 4
    constexpr int UD = R * KD;             // decode units per round and decoding wave
    __shared__ __attribute__((aligned(16))) float wlut_s[256 * VPB];
    char* const ring = smem;                                  // [2][R] decoded tiles: [k-step][hi | lo][lane] 16-byte fragments
    char* const raw = smem + 2 * R * S3Q_SLOT;                // [2][R] gathered centroid rows
    const uint32_t raw_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int pass = (int)blockIdx.y % npass, y = (int)blockIdx.y / npass;
    const int32_t* wb = m.plan_wbeg + (size_t)b * m.plan_wcap;
    const int tb = __builtin_amdgcn_readfirstlane(wb[y]), te = __builtin_amdgcn_readfirstlane(wb[y + 1]);
    if (tb >= te) return;   // (the whole workgroup)
    for (int t = tid; t < 256 * VPB; t += 64 * NW) wlut_s[t] = wlut_g[t];
    const int ntiles = te - tb, nrounds = (ntiles + R - 1) / R;
    const int nqt = nqp >> 5, qt0 = pass * per_pass;
    const int npq = (nqt - qt0) < per_pass ? (nqt - qt0) : per_pass;   // q-tiles of this pass (<= 2 NW)
    const int nt = (wave < npq ? 1 : 0) + (wave + NW < npq ? 1 : 0);   // q-tiles this wave owns: qt0 + wave, qt0 + wave + NW
    const uint2* dq = m.plan_desc + (size_t)b * m.plan_stride + tb;
    // descriptors by scalar loads written as instructions (see maxsim_lean_kernel); tile index relative to the workgroup's first
    // tile, clamped into its range: work past either end repeats a real tile and is never used
    auto ld_desc = [&](int t, u32x2& d) {
        const int tc = t < ntiles ? (t < 0 ? 0 : t) : ntiles - 1;
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(d) : "s"(dq + tc) : "memory");
    };
    auto touch_desc = [&](u32x2& d) { asm volatile("" : "+s"(d)::"memory"); };

    hf8 bhA[8], blA[8], bhB[8], blB[8];
    {
        const hf8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const size_t qb = (size_t)b * nqp;
        const int ra = (qt0 + wave) * 32 + i, rb = (qt0 + wave + NW) * 32 + i;
        const hf8* pha = reinterpret_cast<const hf8*>(m.q_hi + (qb + (nt >= 1 ? ra : 0)) * FLMR_DIM + 64 * h);
        const hf8* pla = reinterpret_cast<const hf8*>(m.q_lo + (qb + (nt >= 1 ? ra : 0)) * FLMR_DIM + 64 * h);
        const hf8* phb = reinterpret_cast<const hf8*>(m.q_hi + (qb + (nt >= 2 ? rb : 0)) * FLMR_DIM + 64 * h);
        const hf8* plb = reinterpret_cast<const hf8*>(m.q_lo + (qb + (nt >= 2 ? rb : 0)) * FLMR_DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            bhA[s] = nt >= 1 ? pha[s] : z; blA[s] = nt >= 1 ? pla[s] : z;
            bhB[s] = nt >= 2 ? phb[s] : z; blB[s] = nt >= 2 ? plb[s] : z;
        }
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(bhA[s]), "+v"(blA[s]), "+v"(bhB[s]), "+v"(blB[s])::"memory");
    }
    // per-lane constants.  Gather: DMA instruction q of a tile moves rows {q, 8+q, 16+q, 24+q} (lane group j = lane >> 4: row 8j + q),
    // 16-byte pieces XOR-swizzled by (row & 15), row r in slot 4 (r & 7) + (r >> 3) of the 8 KB buffer (maxsim_lean_kernel's layout);
    // this wave issues instructions q = wave + e NW (e < KS); a decoding wave (< ND) decodes k-steps s = wave + e ND (e < KD): the
    // offsets below are for e = 0, the others follow by XOR / add of compile-time constants.
    const uint32_t code_voff = (uint32_t)(((lane >> 4) * 8 + wave) * 4);
    const uint32_t piece_off = (uint32_t)((((lane & 15) ^ (8 * ((lane >> 4) & 1))) << 4) ^ (wave << 4));
    const uint32_t res_voff = (uint32_t)(i * PACKED + h * NB + wave * NBITS), inv_voff = (uint32_t)(i * 4);
    const uint32_t rd_off = (uint32_t)(((4 * (i & 7) + (i >> 3)) * 256 + (((8 * h) ^ (i & 15)) << 4)) ^ (wave << 4));   // piece 8h + wave of row i
    const uint32_t wr_off = (uint32_t)(lane * 16 + 2 * wave * 1024);   // this lane's hi fragment of k-step `wave` inside a ring slot (lo: + 1024)
    const uint32_t frag_off = (uint32_t)(lane * 16);

    // in-flight state of the pipeline; DMA unit u = k KS + e, decode unit u = k KD + e: tile k of the round, e-th of my instructions / k-steps
    uint32_t cdv[U];              // codes of my DMA rows, tiles of round r+3 (after stage C of round r)
    uint32_t rsv[UD][2];          // residual bytes of my k-steps, tiles of round r+1 (what stage D of round r decodes; decoding waves)
    float ivv[R];                 // 1 / norm of row i, same tiles
    uint32_t rsn[UD][2];          // the same for the tiles of round r+2: loaded by stage G of round r, moved to rsv / ivv at its end
    float ivn[R];
#pragma unroll
    for (int u = 0; u < U; u++) cdv[u] = 0;
#pragma unroll
    for (int u = 0; u < UD; u++) { rsv[u][0] = 0; rsv[u][1] = 0; rsn[u][0] = 0; rsn[u][1] = 0; }
#pragma unroll
    for (int k = 0; k < R; k++) { ivv[k] = 0.0f; ivn[k] = 0.0f; }
    auto touch_state = [&]() {
#pragma unroll
        for (int u = 0; u < U; u++) asm volatile("" : "+v"(cdv[u])::"memory");
#pragma unroll
        for (int u = 0; u < UD; u++) asm volatile("" : "+v"(rsn[u][0]), "+v"(rsn[u][1])::"memory");
#pragma unroll
        for (int k = 0; k < R; k++) asm volatile("" : "+v"(ivn[k])::"memory");
    };
    // descriptors of the tiles of rounds r (consume), r+1 (decode), r+2 (gather), r+3 (codes), and the ones being loaded (r+4)
    u32x2 dM[R], dD[R], dG[R], dC[R], dN[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        ld_desc(-3 * R + k, dM[k]); ld_desc(-2 * R + k, dD[k]); ld_desc(-1 * R + k, dG[k]); ld_desc(k, dC[k]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < R; k++) { touch_desc(dM[k]); touch_desc(dD[k]); touch_desc(dG[k]); touch_desc(dC[k]); }
    float cmxA = 0.0f, cmxB = 0.0f;
    float* const cm_q = m.colmax_ws + (size_t)b * m.key_stride * (size_t)nqp + (size_t)(qt0 + wave) * 32 + i;
    const float out_scale = (m.colmax_ws + (size_t)m.nqueries * m.key_stride * (size_t)nqp)[b];   // 1 / (64 L_b), s3q_scale_kernel
    __syncthreads();
#ifdef S3Q_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif

    for (int r = -3; r < nrounds; r++) {
#pragma unroll
        for (int k = 0; k < R; k++) ld_desc((r + 4) * R + k, dN[k]);
        // ---- stage D: my k-steps of the tiles of round r+1 -> ring[(r + 1) & 1] (rows landed and fenced at the end of round r-1) ----
        auto stage_d = [&]() {
            const int par = (r + 1) & 1;
            const char* rw = raw + par * (R * S3Q_RAW);
            char* wr = ring + par * (R * S3Q_SLOT) + wr_off;
            hf8 c[UD];
#pragma unroll
            for (int u = 0; u < UD; u++)   // all raw-row reads first: one LDS round trip for the stage (k-step w + 4e: XOR, the bits are disjoint)
                c[u] = *reinterpret_cast<const hf8*>(rw + (u / KD) * S3Q_RAW + (rd_off ^ (uint32_t)(((u % KD) * ND) << 4)));
            float wts[UD][8];
#ifndef S3Q_NO_DECODE
#pragma unroll
            for (int u = 0; u < UD; u++) s3q_unit_weights<NBITS>(wlut_s, rsv[u], wts[u]);   // (all look-ups in flight together)
#endif
#pragma unroll
            for (int u = 0; u < UD; u++) {
                const int k = u / KD, e = u % KD;
                const int lo = (int)(dD[k].y & 63u), hi = (int)((dD[k].y >> 6) & 63u);
                const float inv64 = (i >= lo && i < hi) ? ivv[k] * S3Q_SCALE : 0.0f;     // rows outside the passage become exact zeros
                hf8 fa, fl;
#ifdef S3Q_NO_DECODE
                fa = c[u]; fl = c[u]; asm volatile("" :: "v"(inv64), "v"(rsv[u][0]), "v"(wts[u][0]));
#else
                s3q_unit_split(wts[u], c[u], inv64, fa, fl);
#endif
                *reinterpret_cast<hf8*>(wr + k * S3Q_SLOT + e * ND * 2048) = fa;
                *reinterpret_cast<hf8*>(wr + k * S3Q_SLOT + e * ND * 2048 + 1024) = fl;
            }
        };
        // decoding waves: decode, gather, consume; the others: gather, consume
        if (wave < ND) stage_d();
        S3Q_STAMP(0);
        // ---- stage G: my rows, residual bytes and scales of the tiles of round r+2; stage C: my codes of round r+3 ----
        {
            const uint32_t dst0 = raw_lds + (uint32_t)((r + 2) & 1) * (R * S3Q_RAW) + (uint32_t)wave * 1024;
#pragma unroll
            for (int k = 0; k < R; k++) {
                const uint32_t pos = dG[k].x;
                const uint8_t* rbase = residuals + (size_t)pos * PACKED;
                if (wave < ND) {   // (wave-uniform; every wave issues the same number of operations in a round or none of these)
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(ivn[k]) : "v"(inv_voff), "s"(inv_norm + pos) : "memory");
#pragma unroll
                    for (int e = 0; e < KD; e++) {
                        const int u = k * KD + e;
                        const uint8_t* rb2 = rbase + e * ND * NBITS;
                        if constexpr (NBITS == 1) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(rsn[u][0]) : "v"(res_voff), "s"(rb2) : "memory");
                        else if constexpr (NBITS == 2) asm volatile("global_load_ushort %0, %1, %2" : "=v"(rsn[u][0]) : "v"(res_voff), "s"(rb2) : "memory");
                        else if constexpr (NBITS == 4) asm volatile("global_load_dword %0, %1, %2" : "=v"(rsn[u][0]) : "v"(res_voff), "s"(rb2) : "memory");
                        else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(*reinterpret_cast<u32x2*>(&rsn[u][0])) : "v"(res_voff), "s"(rb2) : "memory");
                    }
                }
#ifndef S3Q_NO_DMA
#pragma unroll
                for (int e = 0; e < KS; e++) {
                    const int u = k * KS + e;
                    const uint32_t dst = dst0 + k * S3Q_RAW + e * NW * 1024;
                    const uint32_t po = piece_off ^ (uint32_t)((e * NW) << 4);
                    uint32_t voff;
                    asm volatile("s_mov_b32 m0, %2\n\tv_lshl_add_u32 %0, %3, 8, %4\n\tglobal_load_lds_dwordx4 %0, %1"
                                 : "=&v"(voff)
                                 : "s"(cen16), "s"(dst), "v"(cdv[u]), "v"(po)
                                 : "memory", "m0");
                }
#endif
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                asm volatile("global_load_dword %0, %1, %2" : "=v"(cdv[u]) : "v"(code_voff), "s"(codes + dC[u / KS].x + (u % KS) * NW) : "memory");
        }
        S3Q_STAMP(1);
        // ---- stage M: consume ring[r & 1] ----
#ifdef S3Q_NO_M
        if (r == 12345 && nt > 0) {
#else
        if (r >= 0 && nt > 0) {
#endif
            const int nrt = (ntiles - r * R) < R ? (ntiles - r * R) : R;
            const char* rbase = ring + (r & 1) * (R * S3Q_SLOT) + frag_off;
            // One MFMA stream over the round's tiles: the fragments of k-step s+1 (of the next tile after a tile's last k-step) are
            // requested before the MFMAs of k-step s, and the accumulators alternate by tile.  Per k-step: hi.lo and lo.hi first,
            // then hi.hi.
            const f32x16 z = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            f32x16 accA[2], accB[2];
            hf8 fh[2], fl[2];
            fh[0] = *reinterpret_cast<const hf8*>(rbase);
            fl[0] = *reinterpret_cast<const hf8*>(rbase + 1024);
            auto tile = [&](auto tc, auto ntc) {
                constexpr int t = decltype(tc)::value;
                constexpr int NT = decltype(ntc)::value;   // q-tiles this wave owns (compile-time inside the MFMA stream: no branches)
                constexpr int P = t & 1;
                const char* slot = rbase + t * S3Q_SLOT;
                const uint32_t dy = dM[t].y;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int cur = (t * 8 + s) & 1, nxt = cur ^ 1;
                    if (s < 7) {
                        fh[nxt] = *reinterpret_cast<const hf8*>(slot + (2 * s + 2) * 1024);
                        fl[nxt] = *reinterpret_cast<const hf8*>(slot + (2 * s + 3) * 1024);
                    } else if (t + 1 < R && t + 1 < nrt) {
                        fh[nxt] = *reinterpret_cast<const hf8*>(slot + S3Q_SLOT);
                        fl[nxt] = *reinterpret_cast<const hf8*>(slot + S3Q_SLOT + 1024);
                    }
                    accA[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], blA[s], s ? accA[P] : z, 0, 0, 0);
                    if constexpr (NT == 2) accB[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], blB[s], s ? accB[P] : z, 0, 0, 0);
                    accA[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur], bhA[s], accA[P], 0, 0, 0);
                    if constexpr (NT == 2) accB[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur], bhB[s], accB[P], 0, 0, 0);
                    accA[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], bhA[s], accA[P], 0, 0, 0);
                    if constexpr (NT == 2) accB[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur], bhB[s], accB[P], 0, 0, 0);
                }
                cmxA = s3q_tile_max(cmxA, accA[P]);
                if constexpr (NT == 2) cmxB = s3q_tile_max(cmxB, accB[P]);
                if ((dy >> 12) & 1u) {   // last tile of its passage: the column maxima leave the registers
                    float* row = cm_q + (size_t)(dy >> 13) * (size_t)nqp;
                    const float va = flmr_xhalf_max(cmxA) * out_scale;
                    if (h == 0) row[0] = va;
                    cmxA = 0.0f;  // segmented_maxsim.cpp:58-59: the running max starts at zero
                    if constexpr (NT == 2) {
                        const float vb = flmr_xhalf_max(cmxB) * out_scale;
                        if (h == 0) row[NW * 32] = vb;
                        cmxB = 0.0f;
                    }
                }
            };
            auto run = [&](auto ntc) {
                tile(std::integral_constant<int, 0>{}, ntc);
                if constexpr (R > 1) { if (nrt > 1) tile(std::integral_constant<int, 1>{}, ntc); }
                if constexpr (R > 2) { if (nrt > 2) tile(std::integral_constant<int, 2>{}, ntc); }
            };
            if (nt == 2) run(std::integral_constant<int, 2>{});
            else run(std::integral_constant<int, 1>{});
        }
        S3Q_STAMP(2);
        // everything this wave asked for in this round has landed (a consume phase ago) -- rows in LDS included -- before the
        // barrier lets the other waves read them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        touch_state();
#pragma unroll
        for (int k = 0; k < R; k++) touch_desc(dN[k]);
        S3Q_STAMP(3);
        __syncthreads();
        S3Q_STAMP(4);
#pragma unroll
        for (int k = 0; k < R; k++) { dM[k] = dD[k]; dD[k] = dG[k]; dG[k] = dC[k]; dC[k] = dN[k]; ivv[k] = ivn[k]; }
#pragma unroll
        for (int u = 0; u < UD; u++) { rsv[u][0] = rsn[u][0]; rsv[u][1] = rsn[u][1]; }
    }
#ifdef S3Q_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 5; k++) atomicAdd(&s3q_prof[wave * 8 + k], (unsigned long long)pt[k]);
        atomicAdd(&s3q_prof[wave * 8 + 7], (unsigned long long)(nrounds + 3));
    }
#endif
}

// k-ascending sums of the column maxima (the order of maxsim_f16_multiq_kernel and of the oracle); one thread per finalist
__global__ __launch_bounds__(256) void s3_colsum_kernel(flmr_maxsim_args m, const int64_t* __restrict__ doc_offsets, int nqp) {
    const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    const int cnt = m.counts[b] < m.max_count ? m.counts[b] : m.max_count;
    if (d >= cnt) return;
    const int pid = m.pids[(size_t)b * m.pid_stride + d];
    if (doc_offsets[pid + 1] - doc_offsets[pid] <= 0) return;   // empty passage: s3_plan_kernel wrote its score
    const int qlen = m.q_lens ? m.q_lens[b] : m.nq;
    const f32x4* row = reinterpret_cast<const f32x4*>(m.colmax_ws + ((size_t)b * m.key_stride + d) * nqp);
    float s = 0.0f;
    int k = 0;
    for (; k + 4 <= qlen; k += 4) {
        const f32x4 v = row[k >> 2];
        s += v[0]; s += v[1]; s += v[2]; s += v[3];
    }
    for (; k < qlen; k++) s += reinterpret_cast<const float*>(row)[k];
    const size_t ko = (size_t)b * m.key_stride + d;
    if (m.keys) m.keys[ko] = flmr_make_key(s, pid);
    if (m.scores) m.scores[ko] = s;
}

template <int NBITS, int NW, int R>
static int launch_maxsim_qs_nw(const flmr_maxsim_args& a, hipStream_t st, int nqp) {
    const flmr_index* ix = a.ix;
    const int nqt = nqp / 32, npass = (nqt + 2 * NW - 1) / (2 * NW);
    // q-tiles per pass: equal shares with two workgroups per CU (their waves land on all SIMDs), full passes first with one
    const int per_pass = NW == 4 ? (nqt + npass - 1) / npass : 2 * NW;
    // workgroups per (query, pass): enough to fill the chip a few times over, at least ~24 tiles each
    int Y = (int)flmr_ceil_div(NW == 4 ? 2048 : 1024, (int64_t)a.nqueries * npass);
    const int64_t tiles_bound = (int64_t)a.max_count * ((ix->max_doclen + 31) / 32);
    if (Y > tiles_bound / 24) Y = (int)(tiles_bound / 24);
    if (Y > a.plan_wcap - 1) Y = a.plan_wcap - 1;
    if (Y < 1) Y = 1;
    const size_t lds = (size_t)2 * R * (S3Q_SLOT + S3Q_RAW);
    float* qscale = a.colmax_ws + (size_t)a.nqueries * a.key_stride * (size_t)nqp;   // [nqueries], behind the column maxima
    hipLaunchKernelGGL(s3q_scale_kernel, dim3(a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq, qscale);
    hipLaunchKernelGGL(s3q_split_q, dim3((nqp * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq, nqp, a.q_hi, a.q_lo, qscale);
    hipLaunchKernelGGL(s3_plan_kernel, dim3(a.nqueries), dim3(256), 0, st, a, ix->doc_offsets, ix->N, Y);
    FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_qs_kernel<NBITS, NW, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((maxsim_qs_kernel<NBITS, NW, R>), dim3(a.nqueries, npass * Y), dim3(64 * NW), lds, st, a, ix->codes, ix->residuals,
                       ix->centroids_f16, ix->wlut, ix->inv_norm, nqp, npass, per_pass);
    hipLaunchKernelGGL(s3_colsum_kernel, dim3((a.max_count + 255) / 256, a.nqueries), dim3(256), 0, st, a, ix->doc_offsets, nqp);
    return FLMR_OK;
}

template <int NBITS>
static int launch_maxsim_qs_t(const flmr_maxsim_args& a, hipStream_t st, int nqp) {
#ifdef S3Q_FORM4   // development probe: two 4-wave workgroups per CU, rounds of one tile (Nq = 832: 5.59 vs 5.03 ms per 256 queries)
    return launch_maxsim_qs_nw<NBITS, 4, 1>(a, st, nqp);
#else
    return launch_maxsim_qs_nw<NBITS, 8, 3>(a, st, nqp);
#endif
}

template <int NBITS>
static int launch_maxsim_f16_t(const flmr_maxsim_args& a, hipStream_t st) {
    const flmr_index* ix = a.ix;
    const int nqp = (int)flmr_round_up(a.nq, 32);
    const size_t lds = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)4 * nqp * sizeof(float);
    if (lds > 64 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large for the MaxSim kernel's LDS column maxima", a.nq);
    // long queries: the query-stationary kernel on planned tiles (below ~6 q-tiles most of a workgroup's waves would hold no query
    // rows: the chunked kernel stays); it splits the query itself (other scaling)
    const bool use_qs = (nqp >= S3Q_MIN_NQP || (nqp > 32 && flmr_opts().is(FLMR_OPT_S3_IMPL, "qs"))) && !flmr_opts().has(FLMR_OPT_S3_NO_MULTIQ) &&
        ix->inv_norm && a.plan_desc && a.plan_wbeg && a.colmax_ws && a.colmax_cap >= (int64_t)a.nqueries * a.key_stride * nqp + a.nqueries && ix->N >= 32 &&
        ix->N < ((int64_t)1 << 32) && ((size_t)ix->K * 256 < ((size_t)1 << 32)) && a.max_count <= S3L_MAX_DOCS && a.plan_wcap >= 2 &&
        a.plan_stride >= (int64_t)a.max_count * ((ix->max_doclen + 31) / 32) &&
        (flmr_opts().is(FLMR_OPT_S3_IMPL, "qs") || !flmr_opts().has(FLMR_OPT_S3_IMPL));
    if (use_qs) {
        const int rc = launch_maxsim_qs_t<NBITS>(a, st, nqp);
        if (rc) return rc;
        FLMR_LAUNCH_CHECK();
        return FLMR_OK;
    }
    if (!a.q_split_done)
        hipLaunchKernelGGL(s3_split_q, dim3((nqp * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq, nqp,
                           a.q_hi, a.q_lo);
    // waves per query: enough to fill the chip, at most 64 documents per wave, at least 1
    int G = (int)flmr_ceil_div(4096, 4 * (int64_t)a.nqueries);
    const int gmin = (int)flmr_ceil_div(a.max_count, 4 * 64);
    if (G < gmin) G = gmin;
    if (G > (int)flmr_ceil_div(a.max_count, 4)) G = (int)flmr_ceil_div(a.max_count, 4);
    if (G < 1) G = 1;
#ifdef S3L_G
    G = S3L_G;
#endif
    if (nqp > 32 && !flmr_opts().has(FLMR_OPT_S3_NO_MULTIQ)) {
        const size_t lds2 = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)S3_QC * 2 * 32 * S3_BROW * sizeof(_Float16) +
                            (size_t)S3_MQW * (32 * S3_QC + 64) * sizeof(float);
        int G2 = (int)flmr_ceil_div(4096, S3_MQW * (int64_t)a.nqueries);
        const int g2min = (int)flmr_ceil_div(a.max_count, S3_MQW * 64);
        if (G2 < g2min) G2 = g2min;
        if (G2 > (int)flmr_ceil_div(a.max_count, S3_MQW)) G2 = (int)flmr_ceil_div(a.max_count, S3_MQW);
        if (G2 < 1) G2 = 1;
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_f16_multiq_kernel<NBITS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(maxsim_f16_multiq_kernel<NBITS>, dim3(a.nqueries, G2), dim3(64 * S3_MQW), lds2, st, a, ix->codes,
                           ix->residuals, ix->doc_offsets, ix->centroids_f16, ix->wlut, nqp);
    } else if (nqp == 32 && ix->inv_norm && ix->wtab16 && ((size_t)ix->K * 256 < ((size_t)1 << 32)) && a.plan_desc && a.plan_wbeg &&
               ix->N >= 32 && ix->N < ((int64_t)1 << 32) && a.max_count <= S3L_MAX_DOCS && a.plan_wcap >= 4 * G + 1 &&
               a.plan_stride >= (int64_t)a.max_count * ((ix->max_doclen + 31) / 32) &&
               (flmr_opts().is(FLMR_OPT_S3_IMPL, "lean") || !flmr_opts().has(FLMR_OPT_S3_IMPL))) {
        // planned tiles (the default for one query tile)
        const int rc = launch_maxsim_lean_t<NBITS>(a, st, G);
        if (rc) return rc;
    } else if (nqp == 32 && ix->inv_norm && ix->wtab16 && ((size_t)ix->K * 256 < ((size_t)1 << 32)) &&
               (flmr_opts().is(FLMR_OPT_S3_IMPL, "cw") || flmr_opts().is(FLMR_OPT_S3_IMPL, "lean") || !flmr_opts().has(FLMR_OPT_S3_IMPL))) {
        // centroid + weight form on the LDS-DMA pipeline (the default for one query tile; 32-bit row offsets: table < 4 GB)
        const size_t tabw = NBITS == 8 ? 256 : 256 * (8 / NBITS);
#ifdef S3_LDS_PAD   // development probe: a larger request leaves one workgroup per CU (one wave per SIMD)
        const size_t lds4 = ((tabw + (size_t)4 * (nqp + 32)) * sizeof(float) + 15) / 16 * 16 + 4 * 16384 + 4 * 2048 + S3_LDS_PAD;
#else
        const size_t lds4 = ((tabw + (size_t)4 * (nqp + 32)) * sizeof(float) + 15) / 16 * 16 + 4 * 16384 + 4 * 2048;
#endif
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_f16_dma_kernel<NBITS, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        hipLaunchKernelGGL((maxsim_f16_dma_kernel<NBITS, true>), dim3(a.nqueries, G), dim3(256), lds4, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids_f16, reinterpret_cast<const float*>(ix->wtab16), nqp, ix->inv_norm);
    } else if (nqp == 32 && ix->inv_norm && ix->wtab16 && flmr_opts().is(FLMR_OPT_S3_IMPL, "cwregs")) {
        // the same arithmetic with register row gathers, one tile in flight (A/B runs; bit-identical to the default)
        const size_t lds4 = (size_t)(NBITS == 8 ? 1024 : 2 * 512 * (8 / NBITS)) + (size_t)4 * 64 * sizeof(float);
        hipLaunchKernelGGL(maxsim_cw_kernel<NBITS>, dim3(a.nqueries, G), dim3(256), lds4, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids_f16, ix->wtab16, ix->inv_norm);
    } else if (nqp == 32 && ((size_t)ix->K * 256 < ((size_t)1 << 32)) && flmr_opts().is(FLMR_OPT_S3_IMPL, "dma")) {
        // (measured, 1 M passages: nbits = 2  2.03 vs 2.04 ms for the register form; nbits = 8  2.31 vs 2.43 ms)
        const size_t lds3 = (((size_t)256 * (8 / NBITS) + (size_t)4 * (nqp + 32)) * sizeof(float) + 15) / 16 * 16 + 4 * 16384 + 4 * 2048;
        FLMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_f16_dma_kernel<NBITS, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        hipLaunchKernelGGL((maxsim_f16_dma_kernel<NBITS, false>), dim3(a.nqueries, G), dim3(256), lds3, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids_f16, ix->wlut, nqp, nullptr);
    } else {
        hipLaunchKernelGGL(maxsim_f16_kernel<NBITS>, dim3(a.nqueries, G), dim3(256), lds, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids_f16, ix->wlut, nqp);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

template <int NBITS>
static int launch_maxsim_t(const flmr_maxsim_args& a, hipStream_t st) {
    const flmr_index* ix = a.ix;
    const size_t lds = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)a.nq * sizeof(int);
    if (lds > 64 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large for the MaxSim kernel's LDS column maxima", a.nq);
    for (int y0 = 0; y0 < a.max_count; y0 += 32768) {
        const int ny = (a.max_count - y0) < 32768 ? (a.max_count - y0) : 32768;
        hipLaunchKernelGGL(maxsim_kernel<NBITS>, dim3(a.nqueries, ny), dim3(256), lds, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids, ix->wlut, y0);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

template <int NBITS>
static int launch_maxsim_f16_gpunum_t(const flmr_maxsim_args& a, hipStream_t st) {
    const flmr_index* ix = a.ix;
    const int nqp = 32;
    const size_t lds = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)4 * nqp * sizeof(float);
    hipLaunchKernelGGL(s3_split_q, dim3((nqp * FLMR_DIM + 255) / 256, a.nqueries), dim3(256), 0, st, a.Q, a.q_lens, a.nq, nqp,
                       a.q_hi, a.q_lo);   // q_hi = half(Q): the only image this mode reads
    int G = (int)flmr_ceil_div(4096, 4 * (int64_t)a.nqueries);
    const int gmin = (int)flmr_ceil_div(a.max_count, 4 * 64);
    if (G < gmin) G = gmin;
    if (G > (int)flmr_ceil_div(a.max_count, 4)) G = (int)flmr_ceil_div(a.max_count, 4);
    if (G < 1) G = 1;
    hipLaunchKernelGGL((maxsim_f16_kernel<NBITS, true>), dim3(a.nqueries, G), dim3(256), lds, st, a, ix->codes, ix->residuals,
                       ix->doc_offsets, ix->centroids_f16, ix->wlut, nqp);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

template <int NBITS>
static int launch_maxsim_gpu_fp16_t(const flmr_maxsim_args& a, hipStream_t st) {
    const flmr_index* ix = a.ix;
    const size_t lds = (size_t)256 * (8 / NBITS) * sizeof(float) + (size_t)a.nq * sizeof(int);
    if (lds > 64 * 1024) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nq=%d too large for the MaxSim kernel's LDS column maxima", a.nq);
    for (int y0 = 0; y0 < a.max_count; y0 += 32768) {
        const int ny = (a.max_count - y0) < 32768 ? (a.max_count - y0) : 32768;
        hipLaunchKernelGGL(maxsim_gpu_fp16_kernel<NBITS>, dim3(a.nqueries, ny), dim3(256), lds, st, a, ix->codes, ix->residuals,
                           ix->doc_offsets, ix->centroids, ix->wlut, y0);
    }
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}

// ------------------------------------------------------------------------------------------------
// Index-side tables of the "cw" form, built once at flmr_index_open (optional: a failure leaves them NULL and S3 takes the
// decompress-normalise-split kernel).
//   inv_norm[t] = 1 / max(|| weight + centroid ||_2, 1e-12): the decompressed row exactly as decompress_residuals.cpp:69-71
//                 forms it (fp32 `weight + centroid`), squared and summed k-ascending per 64-dim half, the halves added.
//   wtab16      = per residual byte the fp16 hi / lo halves (w = hi + 2^-11 lo) of its 8 / nbits weights, packed in the
//                 order maxsim_cw_kernel's look-ups expect.
// ------------------------------------------------------------------------------------------------
template <int NBITS>
__global__ __launch_bounds__(256) void s3_inv_norm_kernel(const int32_t* __restrict__ codes, const uint8_t* __restrict__ residuals,
                                                          const float* __restrict__ centroids, const float* __restrict__ wlut_g,
                                                          int64_t N, float* __restrict__ out) {
    constexpr int VPB = 8 / NBITS;
    constexpr int PACKED = FLMR_DIM / VPB;
    __shared__ float wlut[256 * VPB];
    for (int t = threadIdx.x; t < 256 * VPB; t += 256) wlut[t] = wlut_g[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); wv * 32 < N; wv += nwaves) {
        const int64_t tok = wv * 32 + i;
        const int64_t tc = tok < N ? tok : N - 1;
        float a[64];
        decompress_half_row<NBITS>(residuals + (size_t)tc * PACKED + h * (PACKED / 2), centroids + (size_t)codes[tc] * FLMR_DIM + 64 * h, wlut, a);
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 64; t++) ss = fmaf(a[t], a[t], ss);
        ss += __shfl_xor(ss, 32, 64);
        float nrm = sqrtf(ss);
        nrm = nrm < 1e-12f ? 1e-12f : nrm;
        if (h == 0 && tok < N) out[tok] = 1.0f / nrm;
    }
}

int flmr_build_s3_tables(flmr_index* ix) {
    ix->wtab16 = nullptr;
    ix->inv_norm = nullptr;
    if (!ix->centroids_f16_exact || !ix->centroids_f16 || ix->N <= 0) return FLMR_OK;
    const int vpb = 8 / ix->nbits;
    float wl[256 * 8];
    if (hipMemcpy(wl, ix->wlut, sizeof(float) * 256 * vpb, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return FLMR_OK; }
    uint32_t tab[2 * 512 * 8 / 4];
    const size_t tab_bytes = ix->nbits == 8 ? 1024 : (size_t)2 * 512 * vpb;
    auto split = [](float w, uint16_t& hi, uint16_t& lo) {
        const _Float16 h = (_Float16)w;
        const _Float16 l = (_Float16)((w - (float)h) * 2048.0f);
        memcpy(&hi, &h, 2);
        memcpy(&lo, &l, 2);
    };
    uint16_t* t16 = reinterpret_cast<uint16_t*>(tab);
    for (int byte = 0; byte < 256; byte++)
        for (int l = 0; l < vpb; l++) {
            uint16_t hi, lo;
            split(wl[byte * vpb + l], hi, lo);
            if (ix->nbits == 8) { t16[2 * byte] = hi; t16[2 * byte + 1] = lo; }            // one (hi | lo << 16) word per byte
            else { t16[byte * vpb + l] = hi; t16[256 * vpb + byte * vpb + l] = lo; }      // hi table, then lo table
        }
    if (hipMalloc(reinterpret_cast<void**>(&ix->wtab16), tab_bytes) != hipSuccess ||
        hipMemcpy(ix->wtab16, tab, tab_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ix->inv_norm), ((size_t)ix->N + 64) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ix->wtab16); (void)hipFree(ix->inv_norm);
        ix->wtab16 = nullptr; ix->inv_norm = nullptr;
        return FLMR_OK;   // feature unavailable, not an error
    }
    (void)hipMemset(ix->inv_norm + ix->N, 0, 64 * sizeof(float));
    const int64_t want = flmr_ceil_div(ix->N, 128);
    const dim3 grid((unsigned)(want < 65536 ? want : 65536)), block(256);
    switch (ix->nbits) {
        case 1: hipLaunchKernelGGL(s3_inv_norm_kernel<1>, grid, block, 0, 0, ix->codes, ix->residuals, ix->centroids, ix->wlut, ix->N, ix->inv_norm); break;
        case 2: hipLaunchKernelGGL(s3_inv_norm_kernel<2>, grid, block, 0, 0, ix->codes, ix->residuals, ix->centroids, ix->wlut, ix->N, ix->inv_norm); break;
        case 4: hipLaunchKernelGGL(s3_inv_norm_kernel<4>, grid, block, 0, 0, ix->codes, ix->residuals, ix->centroids, ix->wlut, ix->N, ix->inv_norm); break;
        default: hipLaunchKernelGGL(s3_inv_norm_kernel<8>, grid, block, 0, 0, ix->codes, ix->residuals, ix->centroids, ix->wlut, ix->N, ix->inv_norm); break;
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ix->wtab16); (void)hipFree(ix->inv_norm);
        ix->wtab16 = nullptr; ix->inv_norm = nullptr;
    }
    return FLMR_OK;
}

// FLMR_S3_IMPL = f16 (default when the centroids are fp16-exact and split buffers are supplied) | f32
int flmr_launch_maxsim(const flmr_maxsim_args& a, hipStream_t st) {
    if (a.max_count <= 0) return FLMR_OK;
    if (a.gpu_fp16) {
        // one query tile, fp16 centroid copy and split buffers at hand: the wave-per-document pipeline; else (long queries,
        // FLMR_S3_IMPL=f32) the plain one-workgroup-per-passage kernel with the same arithmetic
        const bool tuned = a.nq <= 32 && a.ix->centroids_f16 && a.q_hi && a.q_lo && !flmr_opts().is(FLMR_OPT_S3_IMPL, "f32");
        switch (a.ix->nbits) {
            case 1: return tuned ? launch_maxsim_f16_gpunum_t<1>(a, st) : launch_maxsim_gpu_fp16_t<1>(a, st);
            case 2: return tuned ? launch_maxsim_f16_gpunum_t<2>(a, st) : launch_maxsim_gpu_fp16_t<2>(a, st);
            case 4: return tuned ? launch_maxsim_f16_gpunum_t<4>(a, st) : launch_maxsim_gpu_fp16_t<4>(a, st);
            case 8: return tuned ? launch_maxsim_f16_gpunum_t<8>(a, st) : launch_maxsim_gpu_fp16_t<8>(a, st);
        }
        FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d", a.ix->nbits);
    }
    const bool f16 = a.ix->centroids_f16_exact && a.ix->centroids_f16 && a.q_hi && a.q_lo && !flmr_opts().is(FLMR_OPT_S3_IMPL, "f32");
    if (f16) {
        switch (a.ix->nbits) {
            case 1: return launch_maxsim_f16_t<1>(a, st);
            case 2: return launch_maxsim_f16_t<2>(a, st);
            case 4: return launch_maxsim_f16_t<4>(a, st);
            case 8: return launch_maxsim_f16_t<8>(a, st);
        }
    }
    switch (a.ix->nbits) {
        case 1: return launch_maxsim_t<1>(a, st);
        case 2: return launch_maxsim_t<2>(a, st);
        case 4: return launch_maxsim_t<4>(a, st);
        case 8: return launch_maxsim_t<8>(a, st);
    }
    FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "nbits=%d", a.ix->nbits);
}

// ------------------------------------------------------------------------------------------------
// exclusive prefix of doc lengths for a pid list (used by the op-level decompress / lookup entry points)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_lengths_kernel(const int32_t* pids, const int64_t* doclens,
                                                            const int64_t* offsets, int32_t n, int64_t* out) {
    __shared__ int scan_lds[17];
    int64_t base = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int len = 0;
        if (i < n) {
            const int p = pids ? pids[i] : i;
            len = (int)(doclens ? doclens[p] : (offsets[p + 1] - offsets[p]));
        }
        int total;
        const int ex = flmr_block_exclusive_scan(len, scan_lds, &total);
        if (i < n) out[i] = base + ex;
        base += total;
    }
    if (threadIdx.x == 0) out[n] = base;
}

int flmr_launch_exclusive_scan_lengths(const int32_t* pids, const int64_t* doclens, const int64_t* offsets, int32_t n,
                                       int64_t* out_offsets, hipStream_t st) {
    hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, st, pids, doclens, offsets, n, out_offsets);
    FLMR_LAUNCH_CHECK();
    return FLMR_OK;
}
