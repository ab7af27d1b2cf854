// codec_kernels.hpp -- residual codec, index-build and encoder-epilogue kernels (gfx950).
// General-shape kernels (any dim that the reference accepts); every floating-point reduction follows the
// canonical order of oracle/colbert_oracle.h, so results are bit-identical to the CPU oracle.
//   dot       d-ascending fmaf chain           sumsq   4 interleaved partial sums, (p0+p1)+(p2+p3)
#pragma once
#include "common.hpp"
#include "search_kernels.hpp"
#include "approx_kernels.hpp"

namespace clb {

__device__ __forceinline__ float dot_canonical(const float* __restrict__ a, const float* __restrict__ b, int dim) {
    float acc = 0.f;
    for (int d = 0; d < dim; ++d) acc = fmaf(a[d], b[d], acc);
    return acc;
}
__device__ __forceinline__ float sumsq_canonical(const float* __restrict__ x, int dim, int stride = 1) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int d = 0;
    for (; d + 3 < dim; d += 4) {
        const float a = x[(size_t)d * stride], b = x[(size_t)(d + 1) * stride];
        const float c = x[(size_t)(d + 2) * stride], e = x[(size_t)(d + 3) * stride];
        const float sa = a * a, sb = b * b, sc = c * c, se = e * e;
        p0 = p0 + sa; p1 = p1 + sb; p2 = p2 + sc; p3 = p3 + se;
    }
    if (d < dim) { const float a = x[(size_t)d * stride]; const float s = a * a; p0 = p0 + s; ++d; }
    if (d < dim) { const float a = x[(size_t)d * stride]; const float s = a * a; p1 = p1 + s; ++d; }
    if (d < dim) { const float a = x[(size_t)d * stride]; const float s = a * a; p2 = p2 + s; ++d; }
    return (p0 + p1) + (p2 + p3);
}

// _normalize_array!(X, dims=1)  utils.jl:320-325 ; one thread per column
static __global__ void normalize_columns_kernel(float* __restrict__ X, int dim, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float* x = X + e * dim;
    const float den = sqrtf(sumsq_canonical(x, dim)) + FLT_EPSILON;
    for (int d = 0; d < dim; ++d) x[d] = x[d] / den;
}

// decompress  residual.jl:759-784, any dim % 8 == 0, nbits in {1,2,4,8}; one thread per embedding
static __global__ void decompress_generic_kernel(int dim, int nbits, const float* __restrict__ C,
                                          const float* __restrict__ weights,
                                          const uint32_t* __restrict__ codes /*1-based*/,
                                          const uint8_t* __restrict__ residuals, int64_t n,
                                          float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int rows = dim / 8 * nbits;
    const uint8_t* r = residuals + e * rows;
    const float* c = C + (size_t)(codes[e] - 1) * dim;
    float* x = out + e * dim;
    const uint32_t mask = (1u << nbits) - 1u;
    for (int d = 0; d < dim; ++d) {
        const int p = d * nbits;
        const uint32_t idx = ((uint32_t)r[p >> 3] >> (p & 7)) & mask;
        x[d] = c[d] + weights[idx];
    }
    const float den = sqrtf(sumsq_canonical(x, dim)) + FLT_EPSILON;
    for (int d = 0; d < dim; ++d) x[d] = x[d] / den;
}

// The same decompression the fused search kernel performs (dim 128), written out -- lets the tests
// pin the product path's decompression bit-for-bit.  One 16-embedding group per wave step.
template <int NBITS>
static __global__ __launch_bounds__(256) void decompress_dim128_kernel(const float* __restrict__ C,
                                                               const float* __restrict__ weights,
                                                               const uint32_t* __restrict__ codes /*1-based*/,
                                                               const uint8_t* __restrict__ residuals,
                                                               int64_t n, float* __restrict__ out) {
    constexpr int RD = NBITS * 4;
    __shared__ float tbl[weight_table_floats<NBITS>()];
    fill_weight_table<NBITS>(tbl, weights);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int64_t groups = (n + 15) / 16;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t el = grp * 16 + r;
        const int64_t e = el < n ? el : n - 1;
        uint32_t R[RD];
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(residuals + (size_t)e * (RD * 4));
#pragma unroll
        for (int k4 = 0; k4 < RD; k4 += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(rp + k4);
            R[k4] = v.x; R[k4 + 1] = v.y; R[k4 + 2] = v.z; R[k4 + 3] = v.w;
        }
        float x[32];
        decompress_lane_dims_fast<NBITS>(C + (size_t)(codes[e] - 1) * kDim + g, R, g, tbl, x);
        if (el < n) {
#pragma unroll
            for (int s = 0; s < 32; ++s) out[(size_t)e * kDim + 4 * s + g] = x[s];
        }
    }
}

// maxsim  ranking.jl:69-86 ; one workgroup per pid, thread t owns query token t
static __global__ void maxsim_generic_kernel(const float* __restrict__ Q, int dim, int T, const float* __restrict__ D,
                                      const int64_t* __restrict__ offsets /*n_pids+1*/,
                                      float* __restrict__ scores) {
    extern __shared__ float sm[];
    const int j = blockIdx.x;
    const int64_t lo = offsets[j], hi = offsets[j + 1];
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float m = 0.f;
        for (int64_t e = lo; e < hi; ++e) {
            const float s = dot_canonical(Q + (size_t)t * dim, D + (size_t)e * dim, dim);
            m = (e == lo || s > m) ? s : m;
        }
        sm[t] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = acc + sm[t];
        scores[j] = acc;
    }
}

// compress_into_codes!  residual.jl:67-81 : argmax inner product, first index on ties (generic dim)
// MODE 0: argmax dot (codes, 1-based UInt32) ; MODE 1: argmin ((-2 dot + c2) + x2) (k-means, Int32 1-based)
template <int MODE>
static __global__ void nearest_centroid_generic_kernel(const float* __restrict__ C, const float* __restrict__ c2,
                                                int dim, int K, const float* __restrict__ X, int64_t n,
                                                uint32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* x = X + e * dim;
    float x2 = 0.f;
    if (MODE == 1) x2 = sumsq_canonical(x, dim);
    int best = 0;
    float bestv = 0.f;
    for (int c = 0; c < K; ++c) {
        float v = dot_canonical(MODE == 1 ? C + (size_t)c * dim : x, MODE == 1 ? x : C + (size_t)c * dim, dim);
        if (MODE == 1) {
            v = -2.0f * v;
            v = v + c2[c];
            v = v + x2;
            if (c == 0 || v < bestv) { bestv = v; best = c; }
        } else {
            if (c == 0 || v > bestv) { bestv = v; best = c; }
        }
    }
    out[e] = (uint32_t)(best + 1);
}

// dim = 128 fast path of the above on the f32 MFMA: one wave owns 32 points (B operand, resident in
// registers) and streams every centroid tile through LDS exactly like centroid_scores_kernel.
// grid = ceil(n/32/2) workgroups of 128 threads; LDS = 2*32*132*4.
template <int MODE>
static __global__ __launch_bounds__(128) void nearest_centroid_mfma_kernel(const float* __restrict__ C,
                                                                    const float* __restrict__ c2, int K,
                                                                    const float* __restrict__ X, int64_t n,
                                                                    uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    float* my = lds + wave * (32 * kCentTileStride);
    const int64_t ptile = (int64_t)blockIdx.x * 2 + wave;
    if (ptile * 32 >= n) return;  // whole wave exits together
    const int64_t pt = ptile * 32 + i;
    const float* xrow = X + (size_t)(pt < n ? pt : n - 1) * kDim;
    float qf[64];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        float4 v = *reinterpret_cast<const float4*>(xrow + 4 * m);
        qf[2 * m] = h ? v.y : v.x;
        qf[2 * m + 1] = h ? v.w : v.z;
    }
    float x2 = 0.f;
    if (MODE == 1) x2 = sumsq_canonical(xrow, kDim);
    float bestv = 0.f;
    int best = 0x7fffffff;
    const int n_tiles = (K + 31) / 32;
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int c0 = tile * 32;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int row = 2 * m + h;
            int c = c0 + row;
            c = c < K ? c : K - 1;
            float4 v = *reinterpret_cast<const float4*>(C + (size_t)c * kDim + 4 * i);
            *reinterpret_cast<float4*>(my + row * kCentTileStride + 4 * i) = v;
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 a4 = *reinterpret_cast<const float4*>(my + i * kCentTileStride + 4 * m);
            const float a0 = h ? a4.y : a4.x;
            const float a1 = h ? a4.w : a4.z;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qf[2 * m], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qf[2 * m + 1], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        // rows (centroids) ascend with r for a fixed half h: (r&3) + 8*(r>>2) + 4h
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (c < K) {
                float v = acc[r];
                if (MODE == 1) {
                    v = -2.0f * v;
                    v = v + c2[c];
                    v = v + x2;
                    if (best == 0x7fffffff || v < bestv || (v == bestv && c < best)) { bestv = v; best = c; }
                } else {
                    if (best == 0x7fffffff || v > bestv || (v == bestv && c < best)) { bestv = v; best = c; }
                }
            }
        }
    }
    // merge the two halves (same point, disjoint centroid rows); the lower index wins ties
    const float ov = __shfl_xor(bestv, 32, 64);
    const int oi = __shfl_xor(best, 32, 64);
    const bool take = MODE == 1 ? (ov < bestv || (ov == bestv && oi < best)) : (ov > bestv || (ov == bestv && oi < best));
    if (oi != 0x7fffffff && (best == 0x7fffffff || take)) { bestv = ov; best = oi; }
    if (h == 0 && pt < n) out[pt] = (uint32_t)(best + 1);
}

static __global__ void centroid_sumsq_kernel(const float* __restrict__ C, int dim, int K, float* __restrict__ c2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < K) c2[c] = sumsq_canonical(C + (size_t)c * dim, dim);
}

// compress  residual.jl:586-604 after the codes: residual = x - C[:,code], bucket index =
// searchsortedfirst(cutoffs, r) - 1 (residual.jl:348-351), LSB-first bit packing (residual.jl:400-407).
// One thread per output byte: fully coalesced stores, no read-modify-write.
static __global__ void pack_residuals_kernel(const float* __restrict__ C, const float* __restrict__ cutoffs, int ncut,
                                      int dim, int nbits, const float* __restrict__ X,
                                      const uint32_t* __restrict__ codes /*1-based*/, int64_t n,
                                      uint8_t* __restrict__ out) {
    const int rows = dim / 8 * nbits;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * rows) return;
    const int64_t e = gid / rows;
    const int j = (int)(gid % rows);
    const float* x = X + e * dim;
    const float* c = C + (size_t)(codes[e] - 1) * dim;
    uint32_t byte = 0;
    for (int bit = 0; bit < 8; ++bit) {
        const int p = 8 * j + bit;
        const int d = p / nbits, b = p % nbits;
        const float r = x[d] - c[d];
        int lo = 0, hi = ncut + 1;  // Julia's searchsortedfirst
        while (lo < hi - 1) {
            const int m = lo + ((hi - lo) >> 1);
            if (cutoffs[m - 1] < r) lo = m; else hi = m;
        }
        const int idx = hi - 1;
        byte |= (uint32_t)((idx >> b) & 1) << bit;
    }
    out[gid] = (uint8_t)byte;
}

// residual values of the held-out sample  collection_indexer.jl:185-187 (+ sum |r| for avg_residual)
static __global__ void heldout_residual_kernel(const float* __restrict__ C, int dim, const float* __restrict__ X,
                                        const uint32_t* __restrict__ codes, int64_t n, float* __restrict__ res,
                                        double* __restrict__ abs_sum) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0;
    if (gid < n * dim) {
        const int64_t e = gid / dim;
        const int d = (int)(gid % dim);
        const float r = X[gid] - C[(size_t)(codes[e] - 1) * dim + d];
        res[gid] = r;
        a = fabs((double)r);
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(abs_sum, a);
}

// ---- k-means update  utils.jl:288-303 -------------------------------------------------------------
// Members of each cluster are visited in ascending point order (stable sort by assignment); within a
// batch of `point_bsize` points the contributions are summed first and the batch sum is then added to
// the running centroid sum -- the order `mul!(new_centroids, batch_data, one_hot', 1, 1)` produces.
// One thread per (cluster, dim): coalesced over dims.
static __global__ void kmeans_accumulate_kernel(const float* __restrict__ X, int dim, const uint32_t* __restrict__ order /*point ids, grouped by cluster*/,
                                         const uint32_t* __restrict__ start /*K+1*/, int K, int point_bsize,
                                         float* __restrict__ newc, int* __restrict__ counts) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)K * dim) return;
    const int c = (int)(gid / dim), d = (int)(gid % dim);
    const uint32_t lo = start[c], hi = start[c + 1];
    float total = 0.f, part = 0.f;
    uint32_t cur_batch = 0xffffffffu;
    for (uint32_t m = lo; m < hi; ++m) {
        const uint32_t pt = order[m];
        const uint32_t bt = pt / (uint32_t)point_bsize;
        if (bt != cur_batch) {
            if (cur_batch != 0xffffffffu) total = total + part;
            part = 0.f;
            cur_batch = bt;
        }
        part = part + X[(size_t)pt * dim + d];
    }
    if (cur_batch != 0xffffffffu) total = total + part;
    newc[gid] = total;
    if (d == 0) counts[c] = (int)(hi - lo);
}

// new ./= max.(counts,1) ; delta = maximum(abs.(centroids - new))   utils.jl:302-306
static __global__ void kmeans_finalize_kernel(float* __restrict__ newc, const int* __restrict__ counts,
                                       const float* __restrict__ oldc, int dim, int K,
                                       unsigned int* __restrict__ delta_bits) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float diff = 0.f;
    if (gid < (int64_t)K * dim) {
        const int c = (int)(gid / dim);
        const int cnt = counts[c];
        const float v = newc[gid] / (float)(cnt > 1 ? cnt : 1);
        newc[gid] = v;
        diff = fabsf(oldc[gid] - v);
    }
    for (int o = 32; o > 0; o >>= 1) diff = fmaxf(diff, __shfl_down(diff, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(delta_bits, __float_as_uint(diff));  // diff >= 0: bits order like floats
}

// Sharded k-means (one shard per GPU): total = ((p_0 + p_1) + p_2) + ... over the ranks' partial sums in RANK order
// (deterministic whatever order the exchange delivered them in), counts added, then the centroid update of
// utils.jl:302-306.  gathered_sums: [world][K*dim], gathered_counts: [world][K].
// sums_stride / counts_stride: distance between two ranks' blocks in floats / in int64 words ([world][K*dim] and
// [world][K] for the separate arrays of clb_kmeans_reduce_update; the packed [sums | counts] blocks of the device
// exchange have both inside one record per rank)
static __global__ void kmeans_reduce_update_kernel(const float* __restrict__ gathered_sums,
                                                   const long long* __restrict__ gathered_counts, int world,
                                                   const float* __restrict__ oldc, int dim, int K,
                                                   float* __restrict__ newc, unsigned int* __restrict__ delta_bits,
                                                   size_t sums_stride, size_t counts_stride) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float diff = 0.f;
    if (gid < (int64_t)K * dim) {
        const int c = (int)(gid / dim);
        long long cnt = 0;
        for (int r = 0; r < world; ++r) cnt += gathered_counts[(size_t)r * counts_stride + c];
        float total = gathered_sums[gid];
        for (int r = 1; r < world; ++r) total = total + gathered_sums[(size_t)r * sums_stride + gid];
        const float v = total / (float)(cnt > 1 ? cnt : 1);
        newc[gid] = v;
        diff = fabsf(oldc[gid] - v);
    }
    for (int o = 32; o > 0; o >>= 1) diff = fmaxf(diff, __shfl_down(diff, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(delta_bits, __float_as_uint(diff));
}
static __global__ void widen_counts_kernel(const int* __restrict__ c32, long long* __restrict__ c64, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) c64[i] = c32[i];
}

static __global__ void widen_u32_i64_kernel(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// dst[i, :] = src[rows[i], :] -- the sampled / shuffled columns of the reference's (dim, n) matrices
// (collection_indexer.jl:56-91) cut out on the device; a row is `per_row` pieces of type P (16 or 4 bytes), one thread per
// piece: consecutive threads copy consecutive pieces of a row (coalesced on both sides).  An index outside [0, n_src)
// zeroes the destination row and raises the flag.
template <class P>
static __global__ __launch_bounds__(256) void gather_rows_kernel(const P* __restrict__ src, int64_t n_src, int64_t per_row,
                                                                 const int64_t* __restrict__ rows, int64_t n, P* __restrict__ dst,
                                                                 int* __restrict__ err) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * per_row) return;
    const int64_t i = t / per_row, j = t - i * per_row;
    const int64_t r = rows[i];
    if (r < 0 || r >= n_src) {
        if (j == 0) atomicOr(err, 1);
        dst[t] = P{};
        return;
    }
    dst[t] = src[r * per_row + j];
}

static __global__ void iota_kernel(uint32_t* __restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
// histogram of 1-based codes into counts[K]; flags codes outside 1..K
static __global__ void code_histogram_kernel(const uint32_t* __restrict__ codes, int64_t n, uint32_t K,
                                      unsigned int* __restrict__ counts, int* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = codes[i];
    if (c < 1u || c > K) { atomicOr(err, 1); return; }
    atomicAdd(&counts[c - 1], 1u);
}
static __global__ void ivf_widen_kernel(const uint32_t* __restrict__ order, int64_t n, int64_t* __restrict__ ivf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ivf[i] = (int64_t)order[i] + 1;
}

// ---- encoder epilogue  checkpoint.jl:27-71, embedding_utils.jl:172-205 ----------------------------
// mask[l,n] = ids[l,n] not in skiplist ; doclens[n] = sum(mask[:,n]) ; one thread per document
static __global__ void epilogue_mask_kernel(const int32_t* __restrict__ ids, int L, int N, const int64_t* __restrict__ skip,
                                     int nskip, uint8_t* __restrict__ mask, int64_t* __restrict__ doclens) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int64_t len = 0;
    for (int l = 0; l < L; ++l) {
        const int32_t id = ids[(size_t)n * L + l];
        bool keep = true;
        for (int s = 0; s < nskip; ++s) keep = keep && ((int64_t)id != skip[s]);
        mask[(size_t)n * L + l] = keep ? 1 : 0;
        len += keep ? 1 : 0;
    }
    doclens[n] = len;
}
// one thread per token column: D .* mask, normalise, write to its compacted slot (or in place)
static __global__ void epilogue_normalize_kernel(const float* __restrict__ D, int dim, int L, int N,
                                          const uint8_t* __restrict__ mask, const int64_t* __restrict__ doc_start /*N, exclusive scan of doclens; null = in place*/,
                                          float* __restrict__ out, int* __restrict__ err = nullptr /* bit 1: non-finite input row */) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (int64_t)L * N) return;
    if (err) {
        const float n2 = sumsq_canonical(D + j * dim, dim);
        if (!(n2 <= FLT_MAX)) atomicOr(err, 2);
    }
    const int n = (int)(j / L), l = (int)(j % L);
    const bool keep = mask[j] != 0;
    const float* x = D + j * dim;
    if (doc_start) {
        if (!keep) return;
        int before = 0;
        for (int q = 0; q < l; ++q) before += mask[(size_t)n * L + q];
        float* o = out + (size_t)(doc_start[n] + before) * dim;
        const float den = sqrtf(sumsq_canonical(x, dim)) + FLT_EPSILON;
        for (int d = 0; d < dim; ++d) o[d] = x[d] / den;
    } else {
        float* o = out + j * dim;
        if (!keep) {
            // D .* mask zeroes the column (-0.0 for negative entries); 0/(0+eps) keeps it
            for (int d = 0; d < dim; ++d) o[d] = (x[d] * 0.0f) / FLT_EPSILON;
            return;
        }
        const float den = sqrtf(sumsq_canonical(x, dim)) + FLT_EPSILON;
        for (int d = 0; d < dim; ++d) o[d] = x[d] / den;
    }
}

// _query_embeddings' epilogue (checkpoint.jl:61-69) for the device-resident query path, mask and normalisation in ONE
// kernel with four lanes per token: lane g owns the dims = g mod 4, i.e. exactly one of the four interleaved partial sums
// of the canonical sum of squares (each still summed in ascending order), combined as (p0 + p1) + (p2 + p3) by two
// shuffles -- bit-identical to epilogue_mask_kernel + epilogue_normalize_kernel (one thread per token: 32 us per 32
// queries for 1 024 rows of 128 floats), which the host-buffer entry points keep.  dim % 4 == 0.
static __global__ __launch_bounds__(256) void epilogue_query_fused_kernel(const float* __restrict__ D, int dim, int64_t n_tok,
                                                                         const int32_t* __restrict__ ids,
                                                                         const int64_t* __restrict__ skip, int nskip,
                                                                         float* __restrict__ out,
                                                                         int* __restrict__ err = nullptr /* bit 1: non-finite row */) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = gid >> 2;
    const int g = (int)(gid & 3);
    const bool live = j < n_tok;
    const int64_t jj = live ? j : n_tok - 1;
    const int32_t id = ids[jj];
    bool keep = true;
    for (int s = 0; s < nskip; ++s) keep = keep && ((int64_t)id != skip[s]);
    const float* x = D + jj * dim + g;
    float p = 0.f;
    for (int d = 0; d < dim; d += 4) {
        const float a = x[d];
        const float sq = a * a;
        p = p + sq;
    }
    const float q = p + __shfl_xor(p, 1, 64);          // lanes 0,1: p0 + p1; lanes 2,3: p2 + p3 (commutative: same bits)
    const float n2 = __shfl(q, (threadIdx.x & 60), 64) + __shfl(q, (threadIdx.x & 60) + 2, 64);   // (p0 + p1) + (p2 + p3)
    if (!live) return;
    if (err && g == 0 && !(n2 <= FLT_MAX)) atomicOr(err, 2);      // NaN / Inf somewhere in the encoder (f16 split range)
    float* o = out + jj * dim + g;
    if (!keep) {
        // D .* mask zeroes the column (-0.0 for negative entries); 0/(0+eps) keeps it
        for (int d = 0; d < dim; d += 4) o[d] = (x[d] * 0.0f) / FLT_EPSILON;
        return;
    }
    const float den = sqrtf(n2) + FLT_EPSILON;
    for (int d = 0; d < dim; d += 4) o[d] = x[d] / den;
}

// -------------------------------------------------------------------------------------------------------------
// Index build (round 5): the group lists of nearest_refine_kernel from ONE fp16 product per fp32 product instead of the three
// bf16 products of centroid_top_bf16x3_mq_kernel<false, BIAS> -- the lists only have to bring the exact winner's group
// within a PROVEN margin of the best listed one (the winner itself is re-scored in canonical fp32), and the margin can
// carry the measured conversion errors of both operands (centroid_product_bound in approx_kernels.hpp):
// |x.c - x'.c'| <= ||x - x'|| ||c'|| + ||x|| ||c - c'||.  A third of the MFMAs, half the centroid bytes, and -- the point
// operands needing half the registers -- NQ = 4 groups of 32 points per wave: 512 points share each staged 32-centroid tile
// (256 before), and one A fragment read from LDS feeds four MFMAs.
// Layout and roles as in the mq kernel: A = 32 centroids x 16 dims from LDS (rows of 272 B, dims 64h + 8s + j in k-step s),
// B = the wave's points in registers, lane (i, h) = (point i of the group, dim half h); accumulator lane (i, h) register r =
// centroid c0 + (r & 3) + 8 (r >> 2) + 4 h.  bias[c] (= -||c||^2 / 2 for the k-means distance) is where the fp32 sums start.
// partial: [group of 32 points][32 points][2 halves][kTopPartial].  grid = ceil(groups32 / (4 NQ)), block = 256,
// LDS = 2 buffers x 32 rows x 272 B.
// -------------------------------------------------------------------------------------------------------------
template <bool BIAS, int NQ, int WPE = 2>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void nearest_top_f16_kernel(
    const uint16_t* __restrict__ C16, const float* __restrict__ X, ValIdx* __restrict__ partial, int K, int groups32,
    int n_tiles, const float* __restrict__ bias, int64_t n_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int bq0 = (int)blockIdx.x * (4 * NQ) + wave * NQ;      // this wave's groups: bq0 .. bq0 + NQ - 1
    u32x4 xq[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int b = bq0 + q < groups32 ? bq0 + q : groups32 - 1;   // past the input: a duplicate whose lists are dropped
        int64_t row = (int64_t)b * 32 + i;
        row = row < n_rows ? row : n_rows - 1;
        const float* xrow = X + (size_t)row * kDim + 64 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * s);
            const float4 c = *reinterpret_cast<const float4*>(xrow + 8 * s + 4);
            xq[q][s] = u32x4{pack_f16(a.x, a.y), pack_f16(a.z, a.w), pack_f16(c.x, c.y), pack_f16(c.z, c.w)};
        }
    }
    float bv[NQ][kTopPartial];
    int bi[NQ][kTopPartial];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) { bv[q][p] = kNegInf; bi[q][p] = 0x7fffffff; }
    // loader: thread tid moves the 16-byte chunk (tid & 15) of rows (tid >> 4) and 16 + (tid >> 4)
    const int prow = threadIdx.x >> 4, pchunk = threadIdx.x & 15;
    auto tile_rows = [&](int tl, u32x4& r0, u32x4& r1) {
        int c0 = tl * 32 + prow, c1 = c0 + 16;
        c0 = c0 < K ? c0 : K - 1;                      // rows past K are copies of row K - 1 (group_max16 masks them)
        c1 = c1 < K ? c1 : K - 1;
        r0 = *reinterpret_cast<const u32x4*>(C16 + (size_t)c0 * kDim + 8 * pchunk);
        r1 = *reinterpret_cast<const u32x4*>(C16 + (size_t)c1 * kDim + 8 * pchunk);
    };
    u32x4 p0, p1;
    tile_rows(0, p0, p1);
    // the tile's 32 bias values come through the scalar cache one tile ahead (the address is uniform): as vector loads next to
    // their use they exposed an L2 round trip per tile
    float bcur[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) bcur[r] = BIAS ? bias[r] : 0.f;
    int buf = 0;
    for (int tile = 0; tile < n_tiles; ++tile) {
        unsigned char* my = lds16 + buf * (32 * kRowBytes16);
        *reinterpret_cast<u32x4*>(my + prow * kRowBytes16 + 16 * pchunk) = p0;
        *reinterpret_cast<u32x4*>(my + (16 + prow) * kRowBytes16 + 16 * pchunk) = p1;
        tile_rows(tile + 1 < n_tiles ? tile + 1 : n_tiles - 1, p0, p1);
        // one barrier per tile: the buffer written now was last read two iterations ago, before the previous barrier
        __syncthreads();
        const int c0 = tile * 32;
        // the accumulators START at the bias (this lane's rows c0 + (r & 3) + 8 (r >> 2) + 4 h: four aligned float4 of the
        // padded bias row) -- the first MFMA of every group reads it as its C operand: no add per score afterwards
        f32x16 init;
#pragma unroll
        for (int r = 0; r < 16; ++r) init[r] = h ? bcur[(r & 3) + 8 * (r >> 2) + 4] : bcur[(r & 3) + 8 * (r >> 2)];
        if (BIAS) {
            const float* bn = bias + (tile + 1 < n_tiles ? c0 + 32 : c0);      // (the bias row is padded by 32 entries)
#pragma unroll
            for (int r = 0; r < 32; ++r) bcur[r] = bn[r];
        }
        f32x16 acc[NQ];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f16x8 a = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(my + i * kRowBytes16 + 16 * (8 * h + s)));
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, xq[q][s]), s == 0 ? init : acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) topn_insert_lazy<kTopPartial>(bv[q], bi[q], group_max16(acc[q], c0, h, K), 2 * tile + h);
        buf ^= 1;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (bq0 + q >= groups32) continue;
        ValIdx* out = partial + (((size_t)(bq0 + q) * 32 + i) * 2 + h) * kTopPartial;
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv[q][p], bi[q][p]};
    }
}

// -------------------------------------------------------------------------------------------------------------
// The same lists with the centroid tiles moved by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) from a
// TILED fp16 table -- to_f16_tiled_kernel writes [tile of 32 centroids][16-byte chunk c of the row: dims 8c .. 8c+7][row][8 halfs],
// 8 KB per tile, rows past K = copies of row K - 1 -- so that a tile is one contiguous block in memory AND lands in LDS
// chunk-major: lane (i, h) reads its A fragment of k-step s at (8h + s) * 512 + 16 i, 32 lanes x 16 contiguous bytes per chunk
// (no bank conflicts, no padding, one base address + immediate offsets).  Ring of three tile buffers, one barrier per tile as
// in gemm_planes2_kernel (encoder_kernels.hpp): tile k waits for its own two DMAs (vmcnt(2): only those of tile k + 1 are
// younger), meets the barrier -- every wave has left buffer (k - 1) % 3 -- and refills that buffer with tile k + 2.
// The eight registers the staging took hold two more A fragments: four are in flight, re-filled three k-steps (12 MFMAs) ahead,
// and the first four of tile k + 1 are requested BEFORE the list epilogue of tile k, which then runs under their latency.
// grid = ceil(groups32 / 16), block = 256, LDS = 3 x 8 KB.
// -------------------------------------------------------------------------------------------------------------
static __global__ void to_f16_tiled_kernel(const float* __restrict__ C, int K, uint16_t* __restrict__ out, int64_t n_chunks) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte chunk per thread
    if (g >= n_chunks) return;
    const int row = (int)(g & 31), chunk = (int)((g >> 5) & 15);
    const int64_t tile = g >> 9;
    int64_t c = tile * 32 + row;
    c = c < K ? c : K - 1;
    const float4 a = *reinterpret_cast<const float4*>(C + (size_t)c * kDim + 8 * chunk);
    const float4 b = *reinterpret_cast<const float4*>(C + (size_t)c * kDim + 8 * chunk + 4);
    *reinterpret_cast<u32x4*>(out + g * 8) = u32x4{pack_f16(a.x, a.y), pack_f16(a.z, a.w), pack_f16(b.x, b.y), pack_f16(b.z, b.w)};
}

template <bool BIAS>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void nearest_top_f16_dma_kernel(
    const uint16_t* __restrict__ C16t, const float* __restrict__ X, ValIdx* __restrict__ partial, int K, int groups32,
    int n_tiles, const float* __restrict__ bias, int64_t n_rows) {
    constexpr int NQ = 4, STAGES = 3, TILEB = 8192;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int bq0 = (int)blockIdx.x * (4 * NQ) + wave * NQ;
    // this wave's two DMAs of a tile: bytes [2 wave, 2 wave + 2) KB of the tile, 16 per lane
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds16 + (uint32_t)wave * 2048u;
    const char* tbase = reinterpret_cast<const char*>(C16t) + wave * 2048;
#define CLB_ND_ISSUE(TL, BUF)                                                                                            \
    {                                                                                                                    \
        const char* t_ = tbase + (int64_t)(TL) * TILEB;                                                                  \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"                                                \
                     :: "v"(voff), "s"(t_), "s"(lds0 + (uint32_t)(BUF) * TILEB) : "memory");                             \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"                                                \
                     :: "v"(voff), "s"(t_ + 1024), "s"(lds0 + (uint32_t)(BUF) * TILEB + 1024u) : "memory");              \
    }
    u32x4 xq[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int b = bq0 + q < groups32 ? bq0 + q : groups32 - 1;   // past the input: a duplicate whose lists are dropped
        int64_t row = (int64_t)b * 32 + i;
        row = row < n_rows ? row : n_rows - 1;
        const float* xrow = X + (size_t)row * kDim + 64 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * s);
            const float4 c = *reinterpret_cast<const float4*>(xrow + 8 * s + 4);
            xq[q][s] = u32x4{pack_f16(a.x, a.y), pack_f16(a.z, a.w), pack_f16(c.x, c.y), pack_f16(c.z, c.w)};
        }
    }
    // (every point load above has been consumed by its conversion: nothing of this wave's is pending in front of the DMAs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLB_ND_ISSUE(0, 0)
    if (n_tiles > 1) CLB_ND_ISSUE(1, 1)
    float bv[NQ][kTopPartial];
    int bi[NQ][kTopPartial];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) { bv[q][p] = kNegInf; bi[q][p] = 0x7fffffff; }
    // the tile's 32 bias values come through the scalar cache (uniform address), requested with the tile's first fragments
    float bcur[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) bcur[r] = 0.f;
    const unsigned char* a_lane = lds16 + h * 4096 + i * 16;
    f32x16 acc[NQ];
    u32x4 af[4];
    int buf = 0;
    // tile k has landed once at most the DMAs of tile k + 1 are pending; behind the barrier buffer (k - 1) % 3 is free
#define CLB_ND_ENTER(KT)                                                                                                 \
    {                                                                                                                    \
        if ((KT) + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");                            \
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                               \
        if ((KT) + 2 < n_tiles) CLB_ND_ISSUE((KT) + 2, buf == 0 ? STAGES - 1 : buf - 1)                                   \
        if (BIAS) _Pragma("unroll") for (int r = 0; r < 32; ++r) bcur[r] = bias[(KT) * 32 + r];                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                    \
            af[s] = *reinterpret_cast<const u32x4*>(a_lane + buf * TILEB + s * 512);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    CLB_ND_ENTER(0)
    // the last tile alone can be partial (its group maxima mask the rows past K): it is peeled off the loop, whose tiles take the
    // plain maximum -- inside one loop hipcc computes the mask's sixteen row compares in FRONT of the branch, for every tile
#define CLB_ND_BODY(LAST)                                                                                                \
    {                                                                                                                    \
        const int c0 = tile * 32;                                                                                        \
        f32x16 init;                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) init[r] = h ? bcur[(r & 3) + 8 * (r >> 2) + 4] : bcur[(r & 3) + 8 * (r >> 2)]; \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        const unsigned char* a_tile = a_lane + buf * TILEB;                                                              \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                                  \
            const f16x8 a = __builtin_bit_cast(f16x8, af[s & 3]);                                                        \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                               \
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, xq[q][s]), s == 0 ? init : acc[q], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            if (s + 4 < 8) {                                                                                             \
                af[s & 3] = *reinterpret_cast<const u32x4*>(a_tile + (s + 4) * 512);                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
            }                                                                                                            \
        }                                                                                                                \
        buf = buf + 1 == STAGES ? 0 : buf + 1;                                                                           \
        if (!(LAST)) CLB_ND_ENTER(tile + 1)          /* the next tile's first fragments fly under the epilogue below */   \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                                 \
            float gm_;                                                                                                   \
            if (LAST) gm_ = group_max16(acc[q], c0, h, K);                                                               \
            else {                                                                                                       \
                const f32x16& v_ = acc[q];                                                                               \
                gm_ = fmaxf(fmaxf(v_[0], v_[1]), v_[2]);                                                                 \
                gm_ = fmaxf(fmaxf(gm_, v_[3]), v_[4]); gm_ = fmaxf(fmaxf(gm_, v_[5]), v_[6]);                             \
                gm_ = fmaxf(fmaxf(gm_, v_[7]), v_[8]); gm_ = fmaxf(fmaxf(gm_, v_[9]), v_[10]);                            \
                gm_ = fmaxf(fmaxf(gm_, v_[11]), v_[12]); gm_ = fmaxf(fmaxf(gm_, v_[13]), v_[14]);                         \
                gm_ = fmaxf(gm_, v_[15]);                                                                                \
            }                                                                                                            \
            topn_insert_lazy<kTopPartial>(bv[q], bi[q], gm_, 2 * tile + h);                                              \
        }                                                                                                                \
    }
    int tile = 0;
    for (; tile + 1 < n_tiles; ++tile) CLB_ND_BODY(0)
    CLB_ND_BODY(1)
#undef CLB_ND_BODY
#undef CLB_ND_ENTER
#undef CLB_ND_ISSUE
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (bq0 + q >= groups32) continue;
        ValIdx* out = partial + (((size_t)(bq0 + q) * 32 + i) * 2 + h) * kTopPartial;
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv[q][p], bi[q][p]};
    }
}

// -------------------------------------------------------------------------------------------------------------
// Index build: exact nearest centroid of every point from the group lists centroid_top_bf16x3_mq_kernel<false, BIAS>
// wrote with gx = 1 (partial: [ceil(n/32)][32 points][2 halves][kTopPartial]).  MODE 0: argmax of the canonical dot
// product, first index on ties (compress_into_codes!, residual.jl:67-81).  MODE 1: argmin of
// fl(fl(-2 dot + ||c||^2) + ||x||^2), first index on ties (assign_clusters of kmeans_gpu_onehot!, utils.jl:38-79).
// With w(c) = x.c - ||c||^2/2 the k-means distance is -2 w + ||x||^2 up to its own rounding (<= delta), and the
// listed group maxima approximate w within e_dot, so the exact winner lives in a group whose maximum is
// >= g1 - (2 e_dot + delta), g1 = the largest group maximum.  16 lanes per point re-score those groups (normally
// one) with the canonical arithmetic.  A full list whose last entry qualifies triggers the exhaustive scan.
// grid = ceil(n / 16), block = 256 (4 waves x 4 points).
// -------------------------------------------------------------------------------------------------------------
// nearest_centroid_mfma_kernel over a LIST of points (the overflow list of nearest_refine_kernel; `count` lives on the device):
// the same exact arithmetic and tie rule, 32 listed points per wave.  The list is normally a few dozen points of millions (near
// ties inside the refine margin), and one wave walking all K centroids for them took 19 ms at K = 131 072 -- per chunk, 2 s of a
// 1 M-passage build.  So the centroid tiles are dealt to gridDim.y work-groups per point tile: each scores its slice with the
// canonical chain and merges its best (value, index) into the point's 64-bit key with one atomicMax -- larger key = better
// value, then SMALLER index (first index on ties, as in the serial order); nearest_list_finalize_kernel turns keys into codes.
// -0 and +0 compare equal in the serial rule, so values are canonicalised (v + 0) before they become keys.
// grid = (point-tile pairs in flight, slices), block = 128; keys[slot] = 0 is set by the thread that appended the slot.
template <int MODE>
__device__ __forceinline__ unsigned long long nearest_key(float v, int c) {
    const uint32_t k = f32_order_key(v + 0.0f);
    return ((unsigned long long)(MODE == 1 ? ~k : k) << 32) | (0xffffffffu - (uint32_t)c);
}
template <int MODE>
static __global__ __launch_bounds__(128) void nearest_centroid_mfma_list_kernel(const float* __restrict__ C,
                                                                         const float* __restrict__ c2, int K,
                                                                         const float* __restrict__ X,
                                                                         const uint32_t* __restrict__ list,
                                                                         const unsigned int* __restrict__ count_p,
                                                                         unsigned long long* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    float* my = lds + wave * (32 * kCentTileStride);
    const int64_t count = (int64_t)*count_p;
    const int n_tiles = (K + 31) / 32;
    const int per = (n_tiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_lo = (int)blockIdx.y * per, t_hi = t_lo + per < n_tiles ? t_lo + per : n_tiles;
    if (t_lo >= t_hi) return;
    for (int64_t ptile = (int64_t)blockIdx.x * 2 + wave; ptile * 32 < count; ptile += (int64_t)gridDim.x * 2) {
        const int64_t slot = ptile * 32 + i;
        const int64_t pt = list[slot < count ? slot : count - 1];
        const float* xrow = X + (size_t)pt * kDim;
        float qf[64];
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 v = *reinterpret_cast<const float4*>(xrow + 4 * m);
            qf[2 * m] = h ? v.y : v.x;
            qf[2 * m + 1] = h ? v.w : v.z;
        }
        float x2 = 0.f;
        if (MODE == 1) x2 = sumsq_canonical(xrow, kDim);
        float bestv = 0.f;
        int best = 0x7fffffff;
        // the next tile's 16 float4 per lane are requested before this tile's 64 MFMAs and stored to LDS after them
        typedef float f32x4_t __attribute__((ext_vector_type(4)));      // (an array of HIP's float4 structs stays in scratch here)
        f32x4_t nx[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            int c = t_lo * 32 + 2 * m + h;
            c = c < K ? c : K - 1;
            nx[m] = *reinterpret_cast<const f32x4_t*>(C + (size_t)c * kDim + 4 * i);
        }
        for (int tile = t_lo; tile < t_hi; ++tile) {
            const int c0 = tile * 32;
#pragma unroll
            for (int m = 0; m < 16; ++m) *reinterpret_cast<f32x4_t*>(my + (2 * m + h) * kCentTileStride + 4 * i) = nx[m];
            const int cn0 = tile + 1 < t_hi ? c0 + 32 : c0;             // (the last iteration re-reads its own tile)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                int c = cn0 + 2 * m + h;
                c = c < K ? c : K - 1;
                nx[m] = *reinterpret_cast<const f32x4_t*>(C + (size_t)c * kDim + 4 * i);
            }
            __builtin_amdgcn_wave_barrier();
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                float4 a4 = *reinterpret_cast<const float4*>(my + i * kCentTileStride + 4 * m);
                const float a0 = h ? a4.y : a4.x;
                const float a1 = h ? a4.w : a4.z;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qf[2 * m], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qf[2 * m + 1], acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (c < K) {
                    float v = acc[r];
                    if (MODE == 1) {
                        v = -2.0f * v;
                        v = v + c2[c];
                        v = v + x2;
                        if (best == 0x7fffffff || v < bestv || (v == bestv && c < best)) { bestv = v; best = c; }
                    } else {
                        if (best == 0x7fffffff || v > bestv || (v == bestv && c < best)) { bestv = v; best = c; }
                    }
                }
            }
        }
        if (slot < count && best != 0x7fffffff) atomicMax(keys + slot, nearest_key<MODE>(bestv, best));
    }
}
static __global__ void nearest_list_finalize_kernel(const uint32_t* __restrict__ list, const unsigned int* __restrict__ count_p,
                                                    const unsigned long long* __restrict__ keys, uint32_t* __restrict__ out) {
    const int64_t count = (int64_t)*count_p;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < count; s += (int64_t)gridDim.x * blockDim.x)
        out[list[s]] = (0xffffffffu - (uint32_t)(keys[s] & 0xffffffffull)) + 1u;
}
template <int MODE>
static __global__ __launch_bounds__(256) void nearest_refine_kernel(const ValIdx* __restrict__ partial,
                                                                   const float* __restrict__ C,
                                                                   const float* __restrict__ c2,
                                                                   const float* __restrict__ X, int64_t n, int K,
                                                                   const unsigned int* __restrict__ cn_max_bits,
                                                                   uint32_t* __restrict__ out,
                                                                   uint32_t* __restrict__ ovf_list = nullptr,
                                                                   unsigned int* __restrict__ ovf_count = nullptr,
                                                                   unsigned long long* __restrict__ ovf_keys = nullptr) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t p = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const int64_t pp = p < n ? p : n - 1;                       // idle quarters shadow the last point
    const float* x = X + (size_t)pp * kDim;
    // ||x|| and ||x - fp16(x)||: 8 dims per lane; the canonical ||x||^2 of MODE 1 is computed separately below
    float part = 0.f, dpart = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float v = x[8 * sub + d], dv = fabsf(v) < 6.0e4f ? v - round_f16(v) : __builtin_inff();
        part = fmaf(v, v, part);
        dpart = fmaf(dv, dv, dpart);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { part += __shfl_xor(part, o, 64); dpart += __shfl_xor(dpart, o, 64); }
    // cn_max_bits[1] = max ||c - fp16(c)|| when the lists come from the single-fp16-product kernel, 0 for the bf16 split
    const float xn = sqrtf(part) * 1.001f, cn = __uint_as_float(cn_max_bits[0]), dc = __uint_as_float(cn_max_bits[1]);
    // (single-product lists: their accumulators start at the bias -||c||^2/2, so its magnitude rides through the 128 adds)
    const float margin = 2.f * kEpsSafety * centroid_product_bound(xn, sqrtf(dpart) * 1.001f, cn, dc) + 4e-6f * (1.f + xn * xn + cn * cn) +
                         (dc > 0.f ? 1.6e-5f * cn * cn : 0.f);
    const float x2 = MODE == 1 ? sumsq_canonical(x, kDim) : 0.f;
    const int64_t b = pp >> 5;
    const int i = (int)(pp & 31);
    const ValIdx* l0 = partial + (((size_t)b * 32 + i) * 2 + 0) * kTopPartial;     // [group of 32 points][point][slot][entry]
    const ValIdx* l1 = partial + (((size_t)b * 32 + i) * 2 + 1) * kTopPartial;
    ValIdx ent[2 * kTopPartial];
#pragma unroll
    for (int e = 0; e < kTopPartial; ++e) { ent[e] = l0[e]; ent[kTopPartial + e] = l1[e]; }
    const float g1 = fmaxf(ent[0].v, ent[kTopPartial].v);
    const float thr = g1 - margin;
    // (a point the fp16 operand cannot hold -- a component beyond its range, NaN -- has no finite margin: scored exactly)
    const bool overflow = (ent[kTopPartial - 1].i != 0x7fffffff && ent[kTopPartial - 1].v >= thr) ||
                          (ent[2 * kTopPartial - 1].i != 0x7fffffff && ent[2 * kTopPartial - 1].v >= thr) ||
                          (dc > 0.f && !(margin < 3.0e38f));
    float bestv = 0.f;
    int best = 0x7fffffff;
    auto consider = [&](int c) {
        const float4* c4 = reinterpret_cast<const float4*>(C + (size_t)c * kDim);
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float a = 0.f;
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {                            // the chain the fp32 MFMA kernel performs
            const float4 cv = c4[m], xv = x4[m];
            a = fmaf(cv.x, xv.x, a);
            a = fmaf(cv.y, xv.y, a);
            a = fmaf(cv.z, xv.z, a);
            a = fmaf(cv.w, xv.w, a);
        }
        if (MODE == 1) {
            float v = -2.0f * a;
            v = v + c2[c];
            v = v + x2;
            if (best == 0x7fffffff || v < bestv || (v == bestv && c < best)) { bestv = v; best = c; }
        } else {
            if (best == 0x7fffffff || a > bestv || (a == bestv && c < best)) { bestv = a; best = c; }
        }
    };
    if (!overflow) {
#pragma unroll
        for (int e = 0; e < 2 * kTopPartial; ++e) {
            if (ent[e].i == 0x7fffffff || !(ent[e].v >= thr)) continue;    // uniform over the point's 16 lanes
            const int gid = ent[e].i;
            const int c = (gid >> 1) * 32 + (sub & 3) + 8 * (sub >> 2) + 4 * (gid & 1);
            if (c < K) consider(c);
        }
    } else if (ovf_list) {
        // mass ties (identical or nearly degenerate centroids: more qualifying groups than a list holds): the point is handed
        // to nearest_centroid_mfma_list_kernel, which scores it against ALL centroids on the fp32 MFMA -- 16 lanes walking K
        // centroids here took ~2 ms per point, and a sample whose embeddings are nearly degenerate sends thousands this way
        if (sub == 0 && p < n) {
            const unsigned int at = atomicAdd(ovf_count, 1u);
            ovf_list[at] = (uint32_t)p;
            ovf_keys[at] = 0ull;                                      // below every key nearest_key forms (index field >= 1)
        }
        return;                                                       // uniform over the point's 16 lanes; no shuffle below is shared
    } else {
        for (int c = sub; c < K; c += 16) consider(c);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(best, o, 64);
        const bool take = oi != 0x7fffffff &&
                          (best == 0x7fffffff || (MODE == 1 ? ov < bestv : ov > bestv) || (ov == bestv && oi < best));
        if (take) { bestv = ov; best = oi; }
    }
    if (sub == 0 && p < n) out[p] = (uint32_t)(best + 1);
}

// Second tier of the nearest-centroid search (codec.hip): the points the single-product lists could not decide are copied
// into a dense block, decided there by the three-product lists (a five times tighter margin), and their codes copied back.
static __global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ X, const uint32_t* __restrict__ list,
                                                                   int64_t count, float* __restrict__ Xc) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 per thread, 32 per point
    if (t >= count * 32) return;
    reinterpret_cast<float4*>(Xc)[t] = reinterpret_cast<const float4*>(X)[(size_t)list[t >> 5] * 32 + (t & 31)];
}
static __global__ void scatter_codes_kernel(const uint32_t* __restrict__ list, int64_t count, const uint32_t* __restrict__ codes_c,
                                            uint32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) out[list[t]] = codes_c[t];
}

static __global__ void half_neg_kernel(const float* __restrict__ c2, int K, float* __restrict__ out, int n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) out[i] = i < K ? -0.5f * c2[i] : 0.f;
}

}  // namespace clb
