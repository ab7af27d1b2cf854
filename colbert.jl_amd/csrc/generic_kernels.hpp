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

// Top-k by a full sort (searching.jl:125-127: sortperm(scores, rev = true), stable): key = ~order(score) << 32 | position
// ascending == score descending, lower candidate position (= lower pid) first.  `list` (two-pass mode) maps positions
// to candidate slots.
static __global__ void generic_topk_keys_kernel(const float* __restrict__ scores, const int* __restrict__ list, int n,
                                                unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int slot = list ? list[i] : i;
    keys[i] = ((unsigned long long)(~f32_order_key(scores[slot])) << 32) | (unsigned long long)(uint32_t)i;
}
static __global__ void generic_topk_emit_kernel(const unsigned long long* __restrict__ keys_sorted,
                                                const float* __restrict__ scores, const uint32_t* __restrict__ cand,
                                                const int* __restrict__ list, int n, int k, int64_t pid_offset,
                                                int64_t* __restrict__ out_pids, float* __restrict__ out_scores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
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
