// generic_kernels.hpp -- the general-shape search path: any dim % 8 == 0, nbits in {1,2,4,8}, any query length, any
// nprobe <= K, any k.  The reference accepts all of these (src/indexing/codecs/residual.jl:698-721,
// src/searching.jl:93-128, src/utils.jl:327-332); the tuned kernels of search_kernels.hpp / approx_kernels.hpp are
// built for dim 128, nbits <= 4, T <= 128, nprobe <= 32, k <= 4096.  Everything here follows the reference's own
// structure (score centroids -> top-nprobe -> union -> decompress -> maxsim -> sort) with the canonical arithmetic of
// the oracle (dot_canonical, sumsq_canonical), one plain thread per output: correct and on the device, not tuned.
#pragma once
#include "codec_kernels.hpp"
#include "common.hpp"
#include "search_kernels.hpp"

namespace clb {

// cells[t * K + c] = <Q_t, C_c>  (ranking.jl:27), canonical fmaf chain.  grid = (ceil(K / 128), T), block = 128,
// dynamic LDS = dim floats.
static __global__ __launch_bounds__(128) void generic_cells_kernel(const float* __restrict__ C,
                                                                  const float* __restrict__ Q, int dim, int K,
                                                                  float* __restrict__ cells) {
    extern __shared__ float gq[];
    const int t = blockIdx.y;
    for (int d = threadIdx.x; d < dim; d += blockDim.x) gq[d] = Q[(size_t)t * dim + d];
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < K) cells[(size_t)t * K + c] = dot_canonical(gq, C + (size_t)c * dim, dim);
}

// The same products on the fp32 MFMA, for B queries at once and in the layout the tuned selection kernels read
// ([b][centroid][Tpad]: topn_partial / topn_final, mark, compaction then serve the general shapes unchanged): one wave per
// 16 centroids, v_mfma_f32_16x16x4_f32 with k walked in ascending order = the canonical fmaf chain of dot_canonical
// (lane (r, g) supplies dims 4s + g of centroid c0 + r and of token t0 + r).  The centroid values of the wave's tile stay in
// registers (dim <= 4 KSMAX) while it walks the token groups of all queries.  grid = (ceil(K / 64), B), block = 256.
template <int KSMAX>
static __global__ __launch_bounds__(256) void generic_cells_mfma_kernel(const float* __restrict__ C, const float* __restrict__ Q,
                                                                       int dim, int K, int T, int Tpad,
                                                                       float* __restrict__ cells) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (c0 >= K) return;
    const int b = blockIdx.y;
    const int ks = dim >> 2;
    const float* crow = C + (size_t)(c0 + r < K ? c0 + r : K - 1) * dim + g;
    float a[KSMAX];
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) a[s] = s < ks ? crow[4 * s] : 0.f;
    float* out = cells + ((size_t)b * K + c0) * Tpad;
    for (int t0 = 0; t0 < Tpad; t0 += 16) {
        const int t = t0 + r;
        const float* qrow = Q + ((size_t)b * T + (t < T ? t : T - 1)) * dim + g;
        float q[KSMAX];
#pragma unroll
        for (int s = 0; s < KSMAX; ++s) q[s] = (s < ks && t < T) ? qrow[4 * s] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSMAX; ++s)
            if (s < ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], q[s], acc, 0, 0, 0);
        // accumulator: lane (col = token r, g) holds centroids c0 + 4 g + i
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c0 + 4 * g + i < K) out[(size_t)(4 * g + i) * Tpad + t] = acc[i];
    }
}

// Top-nprobe per token by a stable radix sort (utils.jl:327-332: partialsortperm(v, 1:k, rev = true), lower index
// first on ties): key = token << 32 | ~order(score) ascending == token ascending, score descending; the sort is
// stable and the values enter in ascending centroid order.  `cells` is addressed with strides so that both layouts
// ([T][K] of generic_cells_kernel and [K][Tpad] of centroid_scores_kernel) can be sorted.
static __global__ void generic_sel_keys_kernel(const float* __restrict__ cells, size_t stride_t, size_t stride_c, int K,
                                               int T, unsigned long long* __restrict__ keys,
                                               uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * K) return;
    const int t = (int)(i / K), c = (int)(i % K);
    const float v = cells[(size_t)t * stride_t + (size_t)c * stride_c];
    keys[i] = ((unsigned long long)(uint32_t)t << 32) | (unsigned long long)(~f32_order_key(v));
    vals[i] = (uint32_t)c;
}
// sel[t * NP + p] = p-th best centroid of token t (0-based), t < T, p < nprobe
static __global__ void generic_sel_extract_kernel(const uint32_t* __restrict__ vals_sorted, int K, int T, int nprobe,
                                                  int NP, int* __restrict__ sel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * nprobe) return;
    const int t = i / nprobe, p = i % nprobe;
    sel[(size_t)t * NP + p] = (int)vals_sorted[(size_t)t * K + p];
}

// decompress + maxsim of the candidate passages of ONE query (residual.jl:759-784, ranking.jl:69-86).  One work-group
// per passage (grid-stride); thread e decompresses and normalises embedding e into the work-group's scratch rows
// (canonical sum of squares, IEEE sqrt and divide), then thread t takes the maximum over the passage of the
// canonical dot products with query token t, and thread 0 adds the T maxima in token order.
// scratch: gridDim.x * max_len * dim floats.  dynamic LDS = T floats.
static __global__ __launch_bounds__(256) void generic_score_kernel(
    const float* __restrict__ C, const float* __restrict__ weights, const uint32_t* __restrict__ codes0,
    const uint8_t* __restrict__ residuals, const uint2* __restrict__ cand_hdr, const int* __restrict__ ncand,
    const float* __restrict__ Q, int dim, int nbits, int T, float* __restrict__ scratch, size_t max_len,
    float* __restrict__ scores) {
    extern __shared__ float gmax[];
    const int n = *ncand;
    const int rows = dim / 8 * nbits;
    const uint32_t mask = (1u << nbits) - 1u;
    float* my = scratch + (size_t)blockIdx.x * max_len * dim;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const uint2 hd = cand_hdr[j];
        const uint32_t off = hd.x, len = hd.y;
        for (uint32_t e = threadIdx.x; e < len; e += blockDim.x) {
            const uint8_t* r = residuals + (size_t)(off + e) * rows;
            const float* c = C + (size_t)codes0[off + e] * dim;
            float* x = my + (size_t)e * dim;
            for (int d = 0; d < dim; ++d) {
                const int p = d * nbits;
                const uint32_t idx = ((uint32_t)r[p >> 3] >> (p & 7)) & mask;
                x[d] = c[d] + weights[idx];
            }
            const float den = sqrtf(sumsq_canonical(x, dim)) + FLT_EPSILON;
            for (int d = 0; d < dim; ++d) x[d] = x[d] / den;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            float m = 0.f;
            for (uint32_t e = 0; e < len; ++e) {
                const float s = dot_canonical(Q + (size_t)t * dim, my + (size_t)e * dim, dim);
                m = (e == 0 || s > m) ? s : m;
            }
            gmax[t] = m;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc = acc + gmax[t];
            scores[j] = acc;
        }
        __syncthreads();
    }
}

// The same on the fp32 MFMA (round 3), for every shape with T <= 16 * kGenericMaxTokenGroups: one WAVE per passage,
// 16 embeddings x 16 tokens per v_mfma_f32_16x16x4_f32 with k walked in ascending order -- exactly the canonical fmaf
// chain of dot_canonical -- in the lane layout of score_exact_kernel: lane (r = lane & 15, g = lane >> 4) owns the dims
// = g mod 4 of embedding r, i.e. one of the four interleaved partial sums of the canonical sum of squares.  Any
// dim % 4 == 0 (the reference requires dim % 8 == 0) and nbits in {1, 2, 4, 8}: the decompressed value of a dim is
// recomputed from the packed residual and the centroid row for every token group instead of being kept (no register
// array sized by dim).  Rows past the end of a passage duplicate its last row (a duplicate cannot change a maximum).
// grid = any, block = 256 (4 waves); dynamic LDS = (1 << nbits) floats (the bucket weights).
constexpr int kGenericMaxTokenGroups = 32;
static __global__ __launch_bounds__(256) void generic_score_mfma_kernel(
    const float* __restrict__ C, const float* __restrict__ weights, const uint32_t* __restrict__ codes0,
    const uint8_t* __restrict__ residuals, const uint2* __restrict__ cand_hdr, const int* __restrict__ ncand,
    const float* __restrict__ Q, int dim, int nbits, int T, float* __restrict__ scores) {
    extern __shared__ float gw[];
    for (int i = threadIdx.x; i < (1 << nbits); i += blockDim.x) gw[i] = weights[i];
    __syncthreads();
    const int n = *ncand;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int rows = dim / 8 * nbits;
    const uint32_t mask = (1u << nbits) - 1u;
    const int ks = dim >> 2;                             // MFMA k-steps: dims 4s .. 4s+3
    const int ngroups = (T + 15) >> 4;
    for (int j = (int)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += (int)gridDim.x * 4) {
        const uint2 hd = cand_hdr[j];
        const uint32_t off = hd.x, len = hd.y;
        float mt[kGenericMaxTokenGroups];                // running maximum of token 16 tg + (lane & 15), per token group
#pragma unroll
        for (int tg = 0; tg < kGenericMaxTokenGroups; ++tg) mt[tg] = kNegInf;
        for (uint32_t e0 = 0; e0 < len; e0 += 16) {
            const uint32_t e = off + (e0 + r < len ? e0 + r : len - 1);
            const uint8_t* rp = residuals + (size_t)e * rows;
            const float* cent = C + (size_t)codes0[e] * dim;
            // canonical sum of squares: lane (r, g) sums dims = g mod 4 in ascending order, then (p0 + p1) + (p2 + p3)
            float p = 0.f;
            for (int s = 0; s < ks; ++s) {
                const int d = 4 * s + g, bit = d * nbits;
                const uint32_t idx = ((uint32_t)rp[bit >> 3] >> (bit & 7)) & mask;
                const float v = cent[d] + gw[idx];
                const float sq = v * v;
                p = p + sq;
            }
            const float a2 = p + __shfl_xor(p, 16, 64);
            const float n2 = a2 + __shfl_xor(a2, 32, 64);
            const float den = sqrtf(n2) + FLT_EPSILON;
#pragma unroll
            for (int tg = 0; tg < kGenericMaxTokenGroups; ++tg) {
                if (tg >= ngroups) break;
                const int t = 16 * tg + r;               // B operand: lane (col = r, k = g) supplies Q[t][4s + g]
                const float* qrow = Q + (size_t)(t < T ? t : T - 1) * dim + g;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < ks; ++s) {
                    const int d = 4 * s + g, bit = d * nbits;
                    const uint32_t idx = ((uint32_t)rp[bit >> 3] >> (bit & 7)) & mask;
                    const float x = (cent[d] + gw[idx]) / den;
                    const float q = t < T ? qrow[4 * s] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, q, acc, 0, 0, 0);
                }
                // accumulator: lane (col = token r, g) holds embeddings 4 g + i, i = 0..3
                float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                mt[tg] = fmaxf(mt[tg], m);
            }
        }
        float total = 0.f;                               // sequential sum over the tokens (ranking.jl:83)
#pragma unroll
        for (int tg = 0; tg < kGenericMaxTokenGroups; ++tg) {
            if (tg >= ngroups) break;
            for (int c = 0; c < 16; ++c) {
                const float v = __shfl(mt[tg], c, 64);
                if (16 * tg + c < T) total = total + v;
            }
        }
        if (lane == 0) scores[j] = total;
    }
}

// The same kernel with a step's 16 decompressed, normalised embeddings held in REGISTERS (dim <= 4 KSMAX) and for B queries
// per launch (grid = (G, B)).  In the loop form above every k-step issues its byte, centroid and query loads inside the
// MFMA chain -- with `dim` and `nbits` run-time values hipcc keeps them there, one memory round trip per k-step: 58 us per
// 16-row step, 0.64 ms per query on 100 k passages of dim 64 / nbits 8 (69 % of the general path).  Here the unrolled,
// predicated loops put all loads of a step in flight at once, the values are decompressed once per step instead of once
// per token group, and the division is the correctly rounded reciprocal form of the tuned kernels (div_by_reciprocal;
// IEEE division outside its guarded range).  Bit-identical to the loop form and to the oracle.
template <int KSMAX>
static __global__ __launch_bounds__(256) void generic_score_mfma_fast_kernel(
    const float* __restrict__ C, const float* __restrict__ weights, const uint32_t* __restrict__ codes0,
    const uint8_t* __restrict__ residuals, const uint2* __restrict__ cand_hdr, const int* __restrict__ ncand,
    const float* __restrict__ Q, int dim, int nbits, int T, size_t cand_cap, float* __restrict__ scores) {
    constexpr int NG = 8;                                // token groups of 16: T <= 128 on this path (host)
    extern __shared__ float gw[];                        // [1 << nbits] bucket weights, then the query: T rows of dim + 1 floats
    const int b = blockIdx.y;
    const float* Qb = Q + (size_t)b * T * dim;
    float* qs = gw + (1 << nbits);
    const int qld = dim + 1;                             // row stride 1 mod 64 banks: lanes r = 0..15 of a read hit 16 banks
    for (int i = threadIdx.x; i < (1 << nbits); i += blockDim.x) gw[i] = weights[i];
    for (int i = threadIdx.x; i < T * dim; i += blockDim.x) qs[(i / dim) * qld + i % dim] = Qb[i];
    __syncthreads();
    const int n = ncand[b];
    const uint2* hdr = cand_hdr + (size_t)b * cand_cap;
    float* out = scores + (size_t)b * cand_cap;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int rows = dim / 8 * nbits;
    const uint32_t mask = (1u << nbits) - 1u;
    const int ks = dim >> 2;
    const int ngroups = (T + 15) >> 4;
    const int stride = (int)gridDim.x * 4;
    int j = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint2 hd = j < n ? hdr[j] : make_uint2(0u, 0u);
    for (; j < n; j += stride) {
        const uint32_t off = hd.x, len = hd.y;
        if (j + stride < n) hd = hdr[j + stride];        // the next passage's header flies during this one
        float mt[NG];
#pragma unroll
        for (int tg = 0; tg < NG; ++tg) mt[tg] = kNegInf;
        uint32_t code = len ? codes0[off + (r < len ? r : len - 1)] : 0u;
        for (uint32_t e0 = 0; e0 < len; e0 += 16) {
            const uint32_t e = off + (e0 + r < len ? e0 + r : len - 1);
            const uint8_t* rp = residuals + (size_t)e * rows;
            const float* cent = C + (size_t)code * dim + g;
            if (e0 + 16 < len) code = codes0[off + (e0 + 16 + r < len ? e0 + 16 + r : len - 1)];   // one step ahead
            float x[KSMAX];
            uint32_t by[KSMAX];
#pragma unroll
            for (int s = 0; s < KSMAX; ++s) {          // all loads of the step first
                const int bit = (4 * s + g) * nbits;
                by[s] = s < ks ? (uint32_t)rp[bit >> 3] : 0u;
                x[s] = s < ks ? cent[4 * s] : 0.f;
            }
            float p = 0.f;
#pragma unroll
            for (int s = 0; s < KSMAX; ++s)
                if (s < ks) {
                    const int bit = (4 * s + g) * nbits;
                    const float v = x[s] + gw[(by[s] >> (bit & 7)) & mask];
                    x[s] = v;
                    const float sq = v * v;
                    p = p + sq;
                }
            const float a2 = p + __shfl_xor(p, 16, 64);
            const float n2 = a2 + __shfl_xor(a2, 32, 64);
            const float den = sqrtf(n2) + FLT_EPSILON;
            if (__builtin_expect(den > 1e-18f && den < 1e18f, 1)) {
                const float y = 1.0f / den;
#pragma unroll
                for (int s = 0; s < KSMAX; ++s) x[s] = div_by_reciprocal(x[s], den, y);
            } else {
#pragma unroll
                for (int s = 0; s < KSMAX; ++s) x[s] = x[s] / den;
            }
#pragma unroll
            for (int tg = 0; tg < NG; ++tg) {
                if (tg < ngroups) {
                    const int t = 16 * tg + r;
                    const float* qrow = qs + (size_t)(t < T ? t : T - 1) * qld + g;      // staged in LDS once per work-group
                    float q[KSMAX];
#pragma unroll
                    for (int s = 0; s < KSMAX; ++s) q[s] = (s < ks && t < T) ? qrow[4 * s] : 0.f;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KSMAX; ++s)
                        if (s < ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], q[s], acc, 0, 0, 0);
                    float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    mt[tg] = fmaxf(mt[tg], m);
                }
            }
        }
        float total = 0.f;
#pragma unroll
        for (int tg = 0; tg < NG; ++tg)
            if (tg < ngroups)
                for (int c = 0; c < 16; ++c) {
                    const float v = __shfl(mt[tg], c, 64);
                    if (16 * tg + c < T) total = total + v;
                }
        if (lane == 0) out[j] = total;
    }
}

// Top-k by a full sort (searching.jl:125-127: sortperm(scores, rev = true), stable): key = ~order(score) << 32 | position
// ascending == score descending, lower candidate position (= lower pid) first.  `list` (two-pass mode) maps positions
// to candidate slots.
// `n_ptr` (device) = the number of valid entries: positions past it get the largest key and sort to the end, so the host
// launches the sort over the slot's capacity without reading the count back (no synchronisation per query).
static __global__ void generic_topk_keys_kernel(const float* __restrict__ scores, const int* __restrict__ list,
                                                const int* __restrict__ n_ptr, int cap,
                                                unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const int n = *n_ptr;
    if (i >= n) { keys[i] = ~0ull; return; }
    const int slot = list ? list[i] : i;
    keys[i] = ((unsigned long long)(~f32_order_key(scores[slot])) << 32) | (unsigned long long)(uint32_t)i;
}
static __global__ void generic_topk_emit_kernel(const unsigned long long* __restrict__ keys_sorted,
                                                const float* __restrict__ scores, const uint32_t* __restrict__ cand,
                                                const int* __restrict__ list, const int* __restrict__ n_ptr,
                                                const int* __restrict__ ncand_ptr, int k, int64_t pid_offset,
                                                int64_t* __restrict__ out_pids, float* __restrict__ out_scores,
                                                int* __restrict__ short_flag, int64_t* __restrict__ n_cand_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *n_ptr;
    if (i == 0) {
        *short_flag = n < k ? 1 : 0;
        if (n_cand_out) *n_cand_out = *ncand_ptr;
    }
    if (i >= k) return;
    if (i < n) {
        const int pos = (int)(uint32_t)keys_sorted[i];
        const int slot = list ? list[pos] : pos;
        out_pids[i] = pid_offset + (int64_t)cand[slot] + 1;
        out_scores[i] = scores[slot];
    } else {
        out_pids[i] = 0;
        out_scores[i] = kNegInf;
    }
}

}  // namespace clb
