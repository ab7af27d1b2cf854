/*
 * colbert_oracle.c -- CPU restatement of ColBERT.jl's hot path.  TEST INFRASTRUCTURE ONLY.
 * See colbert_oracle.h for scope, pinning status, conventions and the canonical arithmetic order.
 * Every function cites the reference file:line (relative to the ColBERT.jl checkout) it restates.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off -mfma -mavx2 -fopenmp).  -ffp-contract=off makes
 * every fused multiply-add explicit (fmaf below); -mfma only makes fmaf() a single instruction.
 */
#include "colbert_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* canonical arithmetic                                                                          */
/* ------------------------------------------------------------------------------------------- */
float orc_dot(const float* a, const float* b, int64_t dim) {
    float acc = 0.0f;
    for (int64_t d = 0; d < dim; ++d) acc = fmaf(a[d], b[d], acc);
    return acc;
}

float orc_sumsq(const float* x, int64_t dim) {
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t d = 0; d < dim; ++d) {
        float sq = x[d] * x[d];
        p[d & 3] = p[d & 3] + sq;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

/* strided variant, for dims = 2 */
static float sumsq_strided(const float* x, int64_t count, int64_t stride) {
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t d = 0; d < count; ++d) {
        float v = x[d * stride];
        float sq = v * v;
        p[d & 3] = p[d & 3] + sq;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

/* utils.jl:320-325  X ./= (sqrt.(sum(abs2, X, dims)) .+ eps(T)) */
int orc_normalize_columns(float* X, int64_t dim, int64_t n) {
#pragma omp parallel for schedule(static) if (n > 256)
    for (int64_t e = 0; e < n; ++e) {
        float* x = X + e * dim;
        float den = sqrtf(orc_sumsq(x, dim)) + FLT_EPSILON;
        for (int64_t d = 0; d < dim; ++d) x[d] = x[d] / den;
    }
    return ORC_OK;
}

int orc_normalize_rows(float* X, int64_t rows, int64_t cols) {
    for (int64_t r = 0; r < rows; ++r) {
        float den = sqrtf(sumsq_strided(X + r, cols, rows)) + FLT_EPSILON;
        for (int64_t c = 0; c < cols; ++c) X[r + c * rows] = X[r + c * rows] / den;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* _topk  utils.jl:327-332 : mapslices(v -> partialsortperm(v, 1:k, rev=true)).                 */
/* partialsortperm orders indices by (value descending, index ascending).                       */
/* ------------------------------------------------------------------------------------------- */
static void topk_slice(const float* v, int64_t len, int64_t stride, int64_t k, int64_t* out,
                       int64_t out_stride) {
    /* insertion into a sorted list of k (value desc, index asc); k is small on this path */
    int64_t have = 0;
    float* bv = (float*)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    for (int64_t i = 0; i < len; ++i) {
        float x = v[i * stride];
        if (have == k && !(x > bv[k - 1])) continue; /* ties keep the earlier index */
        int64_t pos = have < k ? have : k - 1;
        while (pos > 0 && x > bv[pos - 1]) {
            bv[pos] = bv[pos - 1];
            bi[pos] = bi[pos - 1];
            --pos;
        }
        bv[pos] = x;
        bi[pos] = i + 1;
        if (have < k) ++have;
    }
    for (int64_t j = 0; j < k; ++j) out[j * out_stride] = bi[j];
    free(bv);
    free(bi);
}

int orc_topk(const float* data, int64_t rows, int64_t cols, int64_t k, int dims, int64_t* out) {
    if (dims != 1 && dims != 2) return ORC_EDOMAIN; /* utils.jl:330 */
    if (dims == 2) {
        if (k > cols || k < 0) return ORC_EBOUNDS;
        for (int64_t r = 0; r < rows; ++r) topk_slice(data + r, cols, rows, k, out + r, rows);
    } else {
        if (k > rows || k < 0) return ORC_EBOUNDS;
        for (int64_t c = 0; c < cols; ++c) topk_slice(data + c * rows, rows, 1, k, out + c * k, 1);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* k-means pieces  utils.jl:38-89, 253-318                                                      */
/* ------------------------------------------------------------------------------------------- */
/* utils.jl:38-59.  dist = -2 * C'X (mul! with alpha=-2, beta=1 on a zeroed matrix), then
 * .+= sum(centroids.^2), then .+= sum(data.^2). */
int orc_kmeans_distances(float* dist, int64_t dist_rows, int64_t dist_cols, const float* data,
                         int64_t data_dim, int64_t b, const float* centroids, int64_t cent_dim,
                         int64_t K) {
    if (dist_rows != K || dist_cols != b) return ORC_EDIMENSION; /* utils.jl:41-43 */
    if (data_dim != cent_dim) return ORC_EDIMENSION;             /* utils.jl:44-46 */
    int64_t dim = data_dim;
    float* c2 = (float*)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    for (int64_t c = 0; c < K; ++c) c2[c] = orc_sumsq(centroids + c * dim, dim);
#pragma omp parallel for schedule(static) if (b * K > 4096)
    for (int64_t i = 0; i < b; ++i) {
        const float* x = data + i * dim;
        float x2 = orc_sumsq(x, dim);
        for (int64_t c = 0; c < K; ++c) {
            float dot = orc_dot(centroids + c * dim, x, dim);
            float d = -2.0f * dot;
            d = d + c2[c];
            d = d + x2;
            dist[c + i * K] = d;
        }
    }
    free(c2);
    return ORC_OK;
}

/* utils.jl:71-79  findmin(dims=1): first index on ties */
int orc_kmeans_assign(int32_t* assign, int64_t n_assign, const float* dist, int64_t K, int64_t b) {
    if (n_assign != b) return ORC_EDIMENSION; /* utils.jl:73-76 */
    for (int64_t i = 0; i < b; ++i) {
        const float* col = dist + i * K;
        int64_t best = 0;
        for (int64_t c = 1; c < K; ++c)
            if (col[c] < col[best]) best = c;
        assign[i] = (int32_t)(best + 1);
    }
    return ORC_OK;
}

/* utils.jl:81-89 */
int orc_onehot_encode(float* one_hot, int64_t oh_rows, int64_t oh_cols, const int32_t* assign,
                      int64_t b, int64_t k) {
    if (oh_rows != k || oh_cols != b) return ORC_EDIMENSION; /* utils.jl:83-85 */
    for (int64_t i = 0; i < b; ++i) {
        if (assign[i] < 1 || assign[i] > k) return ORC_EBOUNDS;
        one_hot[(assign[i] - 1) + i * k] = 1.0f;
    }
    return ORC_OK;
}

/* utils.jl:61-69  mul!(new_centroids, data, one_hot', 1, 1): the batch's contribution is formed
 * first (points in ascending order), then added to new_centroids. */
int orc_kmeans_update(float* new_centroids, int64_t nc_rows, int64_t nc_cols, const float* data,
                      int64_t dim, int64_t b, const float* one_hot, int64_t oh_rows,
                      int64_t oh_cols) {
    if (nc_rows != dim || nc_cols != oh_rows) return ORC_EDIMENSION; /* utils.jl:64-67 */
    if (oh_cols != b) return ORC_EDIMENSION;
    int64_t K = oh_rows;
    float* part = (float*)calloc((size_t)(dim * K > 0 ? dim * K : 1), sizeof(float));
    for (int64_t i = 0; i < b; ++i)
        for (int64_t c = 0; c < K; ++c) {
            float w = one_hot[c + i * K];
            if (w == 0.0f) continue;
            for (int64_t d = 0; d < dim; ++d)
                part[d + c * dim] = fmaf(data[d + i * dim], w, part[d + c * dim]);
        }
    for (int64_t j = 0; j < dim * K; ++j) new_centroids[j] = new_centroids[j] + part[j];
    free(part);
    return ORC_OK;
}

/* utils.jl:253-318 without the RNG draw at :261 (the caller supplies the initial centroids).
 * Per iteration and per batch of `point_bsize` points: distances -> argmin -> per-cluster partial
 * sums of the batch (ascending point order) -> added onto new_centroids; counts in Int32.
 * Then new ./= max.(counts,1); delta = max|old - new|; `delta < tol` leaves the OLD centroids
 * in place (utils.jl:308-314). */
int orc_kmeans(const float* data, int64_t dim, int64_t n, float* centroids, int64_t K,
               int64_t max_iters, float tol, int64_t point_bsize, int32_t* assignments,
               int64_t* iters_done) {
    if (K <= 0 || dim <= 0 || point_bsize <= 0) return ORC_EDIMENSION;
    float* newc = (float*)malloc(sizeof(float) * (size_t)(dim * K));
    float* part = (float*)malloc(sizeof(float) * (size_t)(dim * K));
    float* c2 = (float*)malloc(sizeof(float) * (size_t)K);
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
    int64_t* touched = (int64_t*)malloc(sizeof(int64_t) * (size_t)point_bsize);
    int64_t it = 0;
    for (it = 0; it < max_iters; ++it) {
        memset(newc, 0, sizeof(float) * (size_t)(dim * K));
        memset(counts, 0, sizeof(int32_t) * (size_t)K);
        for (int64_t c = 0; c < K; ++c) c2[c] = orc_sumsq(centroids + c * dim, dim);
        for (int64_t start = 0; start < n; start += point_bsize) {
            int64_t end = start + point_bsize < n ? start + point_bsize : n;
#pragma omp parallel for schedule(static) if ((end - start) * K > 4096)
            for (int64_t i = start; i < end; ++i) {
                const float* x = data + i * dim;
                float x2 = orc_sumsq(x, dim);
                int64_t best = 0;
                float bestd = 0.0f;
                for (int64_t c = 0; c < K; ++c) {
                    float d = -2.0f * orc_dot(centroids + c * dim, x, dim);
                    d = d + c2[c];
                    d = d + x2;
                    if (c == 0 || d < bestd) {
                        bestd = d;
                        best = c;
                    }
                }
                assignments[i] = (int32_t)(best + 1);
            }
            /* batch partial sums, ascending point order; only touched clusters are flushed */
            int64_t nt = 0;
            for (int64_t i = start; i < end; ++i) {
                int64_t c = assignments[i] - 1;
                int seen = 0;
                for (int64_t j = 0; j < nt; ++j)
                    if (touched[j] == c) { seen = 1; break; }
                if (!seen) {
                    touched[nt++] = c;
                    for (int64_t d = 0; d < dim; ++d) part[d + c * dim] = 0.0f;
                }
                for (int64_t d = 0; d < dim; ++d)
                    part[d + c * dim] = part[d + c * dim] + data[d + i * dim];
                counts[c] += 1;
            }
            for (int64_t j = 0; j < nt; ++j) {
                int64_t c = touched[j];
                for (int64_t d = 0; d < dim; ++d)
                    newc[d + c * dim] = newc[d + c * dim] + part[d + c * dim];
            }
        }
        float delta = 0.0f;
        for (int64_t c = 0; c < K; ++c) {
            float cs = (float)(counts[c] > 1 ? counts[c] : 1);
            for (int64_t d = 0; d < dim; ++d) {
                float v = newc[d + c * dim] / cs;
                newc[d + c * dim] = v;
                float diff = fabsf(centroids[d + c * dim] - v);
                if (diff > delta) delta = diff;
            }
        }
        if (delta < tol) { ++it; break; }
        memcpy(centroids, newc, sizeof(float) * (size_t)(dim * K));
    }
    if (iters_done) *iters_done = it;
    free(newc); free(part); free(c2); free(counts); free(touched);
    return ORC_OK;
}

/* Sharded k-means (the reference has no multi-GPU build; this is the canonical order the
 * multi-GPU index build is checked against).  One iteration of utils.jl:271-306 splits into
 *   (1) per shard: the body of the batch loop over the shard's own points (batches counted from the shard's
 *       first point) -> un-normalised per-cluster sums (dim,K) and counts;
 *   (2) reduction in RANK ORDER: total = ((p_0 + p_1) + p_2) + ..., counts added;
 *   (3) utils.jl:302-314: new = total ./ max.(counts,1); delta = max|old - new|; delta < tol keeps the old centroids.
 * With one shard this is orc_kmeans' iteration bit for bit. */
int orc_kmeans_shard_pass(const float* data, int64_t dim, int64_t n, const float* centroids, int64_t K,
                          int64_t point_bsize, float* sums, int64_t* counts, int32_t* assignments) {
    if (K <= 0 || dim <= 0 || point_bsize <= 0) return ORC_EDIMENSION;
    float* part = (float*)malloc(sizeof(float) * (size_t)(dim * K));
    float* c2 = (float*)malloc(sizeof(float) * (size_t)K);
    int64_t* touched = (int64_t*)malloc(sizeof(int64_t) * (size_t)point_bsize);
    int32_t* assign = assignments ? assignments : (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    memset(sums, 0, sizeof(float) * (size_t)(dim * K));
    memset(counts, 0, sizeof(int64_t) * (size_t)K);
    for (int64_t c = 0; c < K; ++c) c2[c] = orc_sumsq(centroids + c * dim, dim);
    for (int64_t start = 0; start < n; start += point_bsize) {
        int64_t end = start + point_bsize < n ? start + point_bsize : n;
#pragma omp parallel for schedule(static) if ((end - start) * K > 4096)
        for (int64_t i = start; i < end; ++i) {
            const float* x = data + i * dim;
            float x2 = orc_sumsq(x, dim);
            int64_t best = 0;
            float bestd = 0.0f;
            for (int64_t c = 0; c < K; ++c) {
                float d = -2.0f * orc_dot(centroids + c * dim, x, dim);
                d = d + c2[c];
                d = d + x2;
                if (c == 0 || d < bestd) { bestd = d; best = c; }
            }
            assign[i] = (int32_t)(best + 1);
        }
        int64_t nt = 0;
        for (int64_t i = start; i < end; ++i) {
            int64_t c = assign[i] - 1;
            int seen = 0;
            for (int64_t j = 0; j < nt; ++j)
                if (touched[j] == c) { seen = 1; break; }
            if (!seen) {
                touched[nt++] = c;
                for (int64_t d = 0; d < dim; ++d) part[d + c * dim] = 0.0f;
            }
            for (int64_t d = 0; d < dim; ++d) part[d + c * dim] = part[d + c * dim] + data[d + i * dim];
            counts[c] += 1;
        }
        for (int64_t j = 0; j < nt; ++j) {
            int64_t c = touched[j];
            for (int64_t d = 0; d < dim; ++d) sums[d + c * dim] = sums[d + c * dim] + part[d + c * dim];
        }
    }
    free(part); free(c2); free(touched);
    if (!assignments) free(assign);
    return ORC_OK;
}

int orc_kmeans_reduce_update(float* centroids, const float* gathered_sums, const int64_t* gathered_counts,
                             int64_t world, int64_t dim, int64_t K, float tol, float* delta_out,
                             int* converged) {
    if (world < 1 || K <= 0 || dim <= 0) return ORC_EDIMENSION;
    float delta = 0.0f;
    float* newc = (float*)malloc(sizeof(float) * (size_t)(dim * K));
    for (int64_t c = 0; c < K; ++c) {
        int64_t cnt = 0;
        for (int64_t r = 0; r < world; ++r) cnt += gathered_counts[r * K + c];
        float cs = (float)(cnt > 1 ? cnt : 1);
        for (int64_t d = 0; d < dim; ++d) {
            float total = gathered_sums[d + c * dim];
            for (int64_t r = 1; r < world; ++r) total = total + gathered_sums[r * dim * K + d + c * dim];
            float v = total / cs;
            newc[d + c * dim] = v;
            float diff = fabsf(centroids[d + c * dim] - v);
            if (diff > delta) delta = diff;
        }
    }
    if (delta_out) *delta_out = delta;
    int conv = delta < tol;
    if (converged) *converged = conv;
    if (!conv) memcpy(centroids, newc, sizeof(float) * (size_t)(dim * K));
    free(newc);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* codec  src/indexing/codecs/residual.jl                                                       */
/* ------------------------------------------------------------------------------------------- */
/* residual.jl:67-81  argmax(embs' * centroids, dims=2): first maximum wins */
int orc_compress_into_codes(uint32_t* codes, int64_t n_codes, const float* centroids, int64_t dim,
                            int64_t K, const float* embs, int64_t n) {
    if (n_codes != n) return ORC_EDIMENSION; /* residual.jl:72-74 */
    if (K <= 0 && n > 0) return ORC_EARGUMENT;
#pragma omp parallel for schedule(static) if (n * K > 4096)
    for (int64_t e = 0; e < n; ++e) {
        const float* x = embs + e * dim;
        int64_t best = 0;
        float bestv = 0.0f;
        for (int64_t c = 0; c < K; ++c) {
            float v = orc_dot(x, centroids + c * dim, dim);
            if (c == 0 || v > bestv) {
                bestv = v;
                best = c;
            }
        }
        codes[e] = (uint32_t)(best + 1);
    }
    return ORC_OK;
}

/* residual.jl:197-208  bits[b + nbits*(d + dim*e)] = (data[d,e] >> b) & 1 */
int orc_binarize_bits(const int64_t* data, int64_t dim, int64_t b, int nbits, uint8_t* bits) {
    int64_t hi = ((int64_t)1 << nbits) - 1;
    for (int64_t j = 0; j < dim * b; ++j)
        if (data[j] < 0 || data[j] > hi) return ORC_EDOMAIN; /* residual.jl:198-200 */
    for (int64_t j = 0; j < dim * b; ++j)
        for (int bit = 0; bit < nbits; ++bit) bits[bit + (int64_t)nbits * j] = (uint8_t)((data[j] >> bit) & 1);
    return ORC_OK;
}

/* residual.jl:233-240 */
int orc_unbinarize(const uint8_t* bits, int nbits, int64_t dim, int64_t b, int64_t* out) {
    for (int64_t j = 0; j < dim * b; ++j) {
        int64_t v = 0;
        for (int bit = 0; bit < nbits; ++bit) v += (int64_t)bits[bit + (int64_t)nbits * j] << bit;
        out[j] = v;
    }
    return ORC_OK;
}

/* residual.jl:348-351  searchsortedfirst(cutoffs, x) - 1, as Julia's binary search does it */
static inline int64_t searchsortedfirst_m1(const float* v, int64_t n, float x) {
    int64_t lo = 0, hi = n + 1; /* 1-based bounds, exclusive */
    while (lo < hi - 1) {
        int64_t m = lo + ((hi - lo) >> 1);
        if (v[m - 1] < x) lo = m; else hi = m;
    }
    return hi - 1;
}
int orc_bucket_indices(const float* data, int64_t n, const float* cutoffs, int64_t ncut,
                       int64_t* out) {
    for (int64_t j = 0; j < n; ++j) out[j] = searchsortedfirst_m1(cutoffs, ncut, data[j]);
    return ORC_OK;
}

/* residual.jl:400-407  BitArray(vec(bits)).chunks reinterpreted as bytes: flat bit p -> byte p>>3,
 * bit p&7 (least-significant first). */
int orc_packbits(const uint8_t* bits, int nbits, int64_t dim, int64_t b, uint8_t* out) {
    if (dim % 8 != 0) return ORC_EDOMAIN; /* residual.jl:402-403 */
    int64_t total = (int64_t)nbits * dim * b;
    memset(out, 0, (size_t)(total >> 3));
    for (int64_t p = 0; p < total; ++p)
        if (bits[p]) out[p >> 3] |= (uint8_t)(1u << (p & 7));
    return ORC_OK;
}

/* residual.jl:428-441 */
int orc_unpackbits(const uint8_t* packed, int64_t rows, int64_t b, int nbits, uint8_t* bits) {
    if (rows % nbits != 0) return ORC_EDOMAIN; /* residual.jl:429-431 */
    int64_t total = rows * b * 8;
    for (int64_t p = 0; p < total; ++p) bits[p] = (uint8_t)((packed[p >> 3] >> (p & 7)) & 1);
    return ORC_OK;
}

/* residual.jl:518-536 */
int orc_binarize(int64_t dim, int nbits, const float* cutoffs, int64_t ncut, const float* residuals,
                 int64_t b, uint8_t* out) {
    if (dim % 8 != 0) return ORC_EDOMAIN;                          /* residual.jl:520 */
    if (ncut != ((int64_t)1 << nbits) - 1) return ORC_EDOMAIN;     /* residual.jl:521-522 */
    int64_t rows = dim / 8 * nbits;
#pragma omp parallel for schedule(static) if (b > 256)
    for (int64_t e = 0; e < b; ++e) {
        uint8_t* o = out + e * rows;
        memset(o, 0, (size_t)rows);
        for (int64_t d = 0; d < dim; ++d) {
            int64_t idx = searchsortedfirst_m1(cutoffs, ncut, residuals[d + e * dim]);
            for (int bit = 0; bit < nbits; ++bit) {
                int64_t p = d * nbits + bit;
                if ((idx >> bit) & 1) o[p >> 3] |= (uint8_t)(1u << (p & 7));
            }
        }
    }
    return ORC_OK;
}

/* residual.jl:586-604 */
int orc_compress(const float* centroids, int64_t K, const float* cutoffs, int64_t ncut, int64_t dim,
                 int nbits, const float* embs, int64_t n, uint32_t* codes, uint8_t* residuals) {
    int rc = orc_compress_into_codes(codes, n, centroids, dim, K, embs, n);
    if (rc) return rc;
    if (dim % 8 != 0) return ORC_EDOMAIN;
    if (ncut != ((int64_t)1 << nbits) - 1) return ORC_EDOMAIN;
    float* r = (float*)malloc(sizeof(float) * (size_t)(dim * n > 0 ? dim * n : 1));
#pragma omp parallel for schedule(static) if (n > 256)
    for (int64_t e = 0; e < n; ++e) {
        const float* c = centroids + (int64_t)(codes[e] - 1) * dim;
        for (int64_t d = 0; d < dim; ++d) r[d + e * dim] = embs[d + e * dim] - c[d];
    }
    rc = orc_binarize(dim, nbits, cutoffs, ncut, r, n, residuals);
    free(r);
    return rc;
}

/* residual.jl:698-721 */
int orc_decompress_residuals(int64_t dim, int nbits, const float* weights, int64_t nweights,
                             const uint8_t* packed, int64_t rows, int64_t b, float* out) {
    if (dim % 8 != 0) return ORC_EDOMAIN;                         /* residual.jl:701 */
    if (rows != dim / 8 * nbits) return ORC_EDOMAIN;              /* residual.jl:702-704 */
    if (nweights != ((int64_t)1 << nbits)) return ORC_EDOMAIN;    /* residual.jl:705-706 */
#pragma omp parallel for schedule(static) if (b > 256)
    for (int64_t e = 0; e < b; ++e) {
        const uint8_t* r = packed + e * rows;
        for (int64_t d = 0; d < dim; ++d) {
            int64_t idx = 0;
            for (int bit = 0; bit < nbits; ++bit) {
                int64_t p = d * nbits + bit;
                idx |= (int64_t)((r[p >> 3] >> (p & 7)) & 1) << bit;
            }
            out[d + e * dim] = weights[idx];
        }
    }
    return ORC_OK;
}

/* residual.jl:759-784 */
int orc_decompress(int64_t dim, int nbits, const float* centroids, int64_t K, const float* weights,
                   int64_t nweights, const uint32_t* codes, int64_t n_codes, const uint8_t* residuals,
                   int64_t res_rows, int64_t res_cols, float* out) {
    if (n_codes != res_cols) return ORC_EDOMAIN; /* residual.jl:763-765 */
    for (int64_t e = 0; e < n_codes; ++e)
        if (codes[e] < 1 || (int64_t)codes[e] > K) return ORC_EDOMAIN; /* residual.jl:766-768 */
    int rc = orc_decompress_residuals(dim, nbits, weights, nweights, residuals, res_rows, res_cols, out);
    if (rc) return rc;
#pragma omp parallel for schedule(static) if (n_codes > 256)
    for (int64_t e = 0; e < n_codes; ++e) {
        float* x = out + e * dim;
        const float* c = centroids + (int64_t)(codes[e] - 1) * dim;
        for (int64_t d = 0; d < dim; ++d) x[d] = c[d] + x[d];
        float den = sqrtf(orc_sumsq(x, dim)) + FLT_EPSILON;
        for (int64_t d = 0; d < dim; ++d) x[d] = x[d] / den;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* index planning / codec statistics / IVF  src/indexing/collection_indexer.jl                  */
/* ------------------------------------------------------------------------------------------- */
int64_t orc_num_sampled_pids(int64_t num_documents) {
    double v = 16.0 * sqrt(120.0 * (double)num_documents); /* collection_indexer.jl:18-19 */
    double m = 1.0 + floor(v);
    return (int64_t)(m < (double)num_documents ? m : (double)num_documents);
}

int64_t orc_heldout_size(int64_t num_sample_embs, float heldout_fraction) {
    /* collection_indexer.jl:85-86: Int(max(1, floor(min(50000, heldout_fraction * n)))) in Float32 */
    float prod = heldout_fraction * (float)num_sample_embs;
    float m = prod < 50000.0f ? prod : 50000.0f;
    float f = floorf(m);
    return (int64_t)(f > 1.0f ? f : 1.0f);
}

int orc_setup(int64_t num_documents, float avg_doclen_est, int64_t num_clustering_embs,
              int64_t chunksize, int64_t nranks, int64_t* out_chunksize, int64_t* out_num_chunks,
              int64_t* out_num_partitions, double* out_num_embeddings_est) {
    if (chunksize <= 0) { /* missing: collection_indexer.jl:117-119 */
        int64_t alt = 1 + num_documents / nranks;
        chunksize = alt < 25000 ? alt : 25000;
    }
    int64_t num_chunks = (num_documents + chunksize - 1) / chunksize;
    /* :124 Int * Float32 -> Float32 ; :126 16 * sqrt(Float32) stays Float32 */
    float est = (float)num_documents * avg_doclen_est;
    float arg = 16.0f * sqrtf(est);
    float p2 = exp2f(floorf(log2f(arg)));
    float parts = floorf(p2);
    float cap = (float)num_clustering_embs;
    *out_chunksize = chunksize;
    *out_num_chunks = num_chunks;
    *out_num_partitions = (int64_t)(cap < parts ? cap : parts);
    *out_num_embeddings_est = (double)est;
    return ORC_OK;
}

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}
/* Statistics.quantile, type 7 (alpha = beta = 1), Float64 p on Float32 data, then Float32(.) */
static float quantile7(const float* sorted, int64_t n, double p) {
    double m = 1.0 + p * (1.0 - 1.0 - 1.0);
    double aleph = (double)n * p + m;
    int64_t j = (int64_t)trunc(aleph);
    if (j < 1) j = 1;
    if (j > n - 1) j = n - 1;
    double g = aleph - (double)j;
    if (g < 0.0) g = 0.0;
    if (g > 1.0) g = 1.0;
    float a, b;
    if (n == 1) { a = sorted[0]; b = sorted[0]; }
    else { a = sorted[j - 1]; b = sorted[j]; }
    float diff = b - a; /* Float32 subtraction, then promoted */
    double r = (double)a + g * (double)diff;
    return (float)r;
}
/* collection_indexer.jl:141-152 */
int orc_bucket_cutoffs_and_weights(int nbits, float* values, int64_t n, float* cutoffs,
                                   float* weights) {
    if (n <= 0) return ORC_EARGUMENT;
    int64_t nopt = (int64_t)1 << nbits;
    qsort(values, (size_t)n, sizeof(float), cmp_float);
    for (int64_t q = 1; q < nopt; ++q) cutoffs[q - 1] = quantile7(values, n, (double)q / (double)nopt);
    for (int64_t q = 0; q < nopt; ++q)
        weights[q] = quantile7(values, n, (double)q / (double)nopt + 0.5 / (double)nopt);
    return ORC_OK;
}

/* collection_indexer.jl:177-195.  avg_residual = mean over dims of (mean over embeddings of |r|);
 * Julia's mean() accumulates Float32 pairwise -- the scalar is a diagnostic (never read on the hot
 * path), so the oracle uses a double accumulator and rounds once; compare with a tolerance. */
int orc_compute_avg_residuals(int nbits, const float* centroids, int64_t dim, int64_t K,
                              const float* heldout, int64_t n, uint32_t* codes, int64_t n_codes,
                              float* cutoffs, float* weights, float* avg_residual) {
    if (n_codes != n) return ORC_EDIMENSION; /* collection_indexer.jl:180-182 */
    int rc = orc_compress_into_codes(codes, n_codes, centroids, dim, K, heldout, n);
    if (rc) return rc;
    float* res = (float*)malloc(sizeof(float) * (size_t)(dim * n > 0 ? dim * n : 1));
    double tot = 0.0;
    for (int64_t e = 0; e < n; ++e) {
        const float* c = centroids + (int64_t)(codes[e] - 1) * dim;
        for (int64_t d = 0; d < dim; ++d) {
            float r = heldout[d + e * dim] - c[d];
            res[d + e * dim] = r;
            tot += fabs((double)r);
        }
    }
    *avg_residual = (float)(tot / (double)(dim * n));
    rc = orc_bucket_cutoffs_and_weights(nbits, res, dim * n, cutoffs, weights);
    free(res);
    return rc;
}

/* collection_indexer.jl:342-347 */
int orc_collect_embedding_id_offset(const int64_t* counts, int64_t n, int64_t* total,
                                    int64_t* offsets) {
    if (n <= 0) { *total = 0; offsets[0] = 0; return ORC_OK; }
    int64_t run = 1, sum = 0;
    for (int64_t i = 0; i < n; ++i) {
        offsets[i] = run;
        run += counts[i];
        sum += counts[i];
    }
    *total = sum;
    return ORC_OK;
}

/* collection_indexer.jl:349-353  sortperm(codes) is stable: counting sort by code */
int orc_build_ivf(const uint32_t* codes, int64_t n, int64_t K, int64_t* ivf, int64_t* ivf_lengths) {
    for (int64_t c = 0; c < K; ++c) ivf_lengths[c] = 0;
    for (int64_t e = 0; e < n; ++e) {
        if (codes[e] < 1 || (int64_t)codes[e] > K) return ORC_EBOUNDS; /* counts() would throw */
        ivf_lengths[codes[e] - 1] += 1;
    }
    int64_t* start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
    int64_t run = 0;
    for (int64_t c = 0; c < K; ++c) { start[c] = run; run += ivf_lengths[c]; }
    for (int64_t e = 0; e < n; ++e) ivf[start[codes[e] - 1]++] = e + 1;
    free(start);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* encoder epilogue  src/modelling/embedding_utils.jl, checkpoint.jl                            */
/* ------------------------------------------------------------------------------------------- */
int orc_mask_skiplist(uint8_t* mask, const int32_t* ids, int64_t count, const int64_t* skiplist,
                      int64_t nskip) {
    for (int64_t s = 0; s < nskip; ++s)
        for (int64_t j = 0; j < count; ++j) mask[j] = mask[j] & (uint8_t)(ids[j] != skiplist[s]);
    return ORC_OK;
}

int orc_doc_epilogue(float* D, int64_t dim, int64_t L, int64_t N, const int32_t* ids,
                     const int64_t* skiplist, int64_t nskip, float* out, int64_t* doclens,
                     int64_t* n_out) {
    int64_t count = L * N;
    uint8_t* mask = (uint8_t*)malloc((size_t)(count > 0 ? count : 1));
    memset(mask, 1, (size_t)count);
    orc_mask_skiplist(mask, ids, count, skiplist, nskip);          /* embedding_utils.jl:186-187 */
    for (int64_t j = 0; j < count; ++j)                             /* :191  D .= D .* mask */
        for (int64_t d = 0; d < dim; ++d) D[d + j * dim] = D[d + j * dim] * (float)mask[j];
    orc_normalize_columns(D, dim, count);                           /* checkpoint.jl:33 */
    int64_t kept = 0;
    for (int64_t nn = 0; nn < N; ++nn) {
        int64_t len = 0;
        for (int64_t l = 0; l < L; ++l) {
            int64_t j = l + nn * L;
            if (mask[j]) {
                memcpy(out + kept * dim, D + j * dim, sizeof(float) * (size_t)dim);
                ++kept;
                ++len;
            }
        }
        doclens[nn] = len;
    }
    *n_out = kept;
    free(mask);
    return ORC_OK;
}

int orc_query_epilogue(float* Q, int64_t dim, int64_t L, int64_t N, const int32_t* ids,
                       const int64_t* skiplist, int64_t nskip) {
    int64_t count = L * N;
    uint8_t* mask = (uint8_t*)malloc((size_t)(count > 0 ? count : 1));
    memset(mask, 1, (size_t)count);
    orc_mask_skiplist(mask, ids, count, skiplist, nskip);
    for (int64_t j = 0; j < count; ++j)
        for (int64_t d = 0; d < dim; ++d) Q[d + j * dim] = Q[d + j * dim] * (float)mask[j];
    orc_normalize_columns(Q, dim, count);
    free(mask);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* search  src/search/ranking.jl, src/searching.jl                                              */
/* ------------------------------------------------------------------------------------------- */

/* T independent canonical dot products against one vector x: acc[t] = dot(Q[:,t], x).  Qt is Q
 * transposed to [d][t] so the t-loop vectorises; every chain keeps the d-ascending fmaf order, so
 * the result is bit-identical to orc_dot. */
static void dots_all_tokens(const float* Qt, int64_t T, int64_t dim, const float* x, float* acc) {
    for (int64_t t = 0; t < T; ++t) acc[t] = 0.0f;
    for (int64_t d = 0; d < dim; ++d) {
        const float xd = x[d];
        const float* q = Qt + d * T;
        for (int64_t t = 0; t < T; ++t) acc[t] = fmaf(q[t], xd, acc[t]);
    }
}
static float* transpose_Q(const float* Q, int64_t dim, int64_t T) {
    float* Qt = (float*)malloc(sizeof(float) * (size_t)(dim * T > 0 ? dim * T : 1));
    for (int64_t t = 0; t < T; ++t)
        for (int64_t d = 0; d < dim; ++d) Qt[d * T + t] = Q[d + t * dim];
    return Qt;
}

/* searching.jl:82-91 */
int orc_build_emb2pid(const int64_t* doclens, int64_t n_docs, int64_t* emb2pid) {
    int64_t off = 0;
    for (int64_t p = 0; p < n_docs; ++p) {
        for (int64_t j = 0; j < doclens[p]; ++j) emb2pid[off + j] = p + 1;
        off += doclens[p];
    }
    return ORC_OK;
}

/* ranking.jl:7-21 */
int orc_cids_to_eids(int64_t* eids, int64_t n_eids, const int64_t* cids, int64_t n_cids,
                     const int64_t* ivf, int64_t n_ivf, const int64_t* ivf_lengths, int64_t K) {
    int64_t need = 0, tot = 0;
    for (int64_t j = 0; j < n_cids; ++j) {
        if (cids[j] < 1 || cids[j] > K) return ORC_EBOUNDS;
        need += ivf_lengths[cids[j] - 1];
    }
    for (int64_t c = 0; c < K; ++c) tot += ivf_lengths[c];
    if (n_eids != need) return ORC_EDIMENSION; /* ranking.jl:9-10 */
    if (n_ivf != tot) return ORC_EDIMENSION;   /* ranking.jl:11-12 */
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
    int64_t run = 0;
    for (int64_t c = 0; c < K; ++c) { off[c] = run; run += ivf_lengths[c]; }
    int64_t w = 0;
    for (int64_t j = 0; j < n_cids; ++j) {
        int64_t c = cids[j] - 1;
        memcpy(eids + w, ivf + off[c], sizeof(int64_t) * (size_t)ivf_lengths[c]);
        w += ivf_lengths[c];
    }
    free(off);
    return ORC_OK;
}

static int cmp_i64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}
static int64_t sort_unique_i64(int64_t* v, int64_t n) {
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof(int64_t), cmp_i64);
    int64_t w = 1;
    for (int64_t i = 1; i < n; ++i)
        if (v[i] != v[w - 1]) v[w++] = v[i];
    return w;
}

/* ranking.jl:23-44 */
int orc_retrieve(const int64_t* ivf, int64_t n_ivf, const int64_t* ivf_lengths, int64_t K,
                 const float* centroids, int64_t dim, const int64_t* emb2pid, int64_t n_emb,
                 int64_t nprobe, const float* Q, int64_t T, int64_t* out_pids, int64_t* n_out) {
    if (nprobe > K || nprobe < 0) return ORC_EBOUNDS;
    /* cells = Q' * centroids  (T, K)  ranking.jl:27 */
    float* cells = (float*)malloc(sizeof(float) * (size_t)(T * K > 0 ? T * K : 1));
    float* Qt = transpose_Q(Q, dim, T);
#pragma omp parallel for schedule(static) if (K > 256)
    for (int64_t c = 0; c < K; ++c) dots_all_tokens(Qt, T, dim, centroids + c * dim, cells + c * T);
    free(Qt);
    /* _topk(cells, nprobe, dims=2) ; sort(unique(vec(.)))  ranking.jl:31-32 */
    int64_t* top = (int64_t*)malloc(sizeof(int64_t) * (size_t)(T * nprobe > 0 ? T * nprobe : 1));
    int rc = orc_topk(cells, T, K, nprobe, 2, top);
    free(cells);
    if (rc) { free(top); return rc; }
    int64_t ncid = sort_unique_i64(top, T * nprobe);
    /* eids  ranking.jl:35-39 */
    int64_t need = 0;
    for (int64_t j = 0; j < ncid; ++j) need += ivf_lengths[top[j] - 1];
    int64_t* eids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(need > 0 ? need : 1));
    rc = orc_cids_to_eids(eids, need, top, ncid, ivf, n_ivf, ivf_lengths, K);
    free(top);
    if (rc) { free(eids); return rc; }
    int64_t ne = sort_unique_i64(eids, need);
    /* pids = sort(unique(emb2pid[eids]))  ranking.jl:42-43 */
    for (int64_t j = 0; j < ne; ++j) {
        if (eids[j] < 1 || eids[j] > n_emb) { free(eids); return ORC_EBOUNDS; }
        eids[j] = emb2pid[eids[j] - 1];
    }
    int64_t np = sort_unique_i64(eids, ne);
    memcpy(out_pids, eids, sizeof(int64_t) * (size_t)np);
    *n_out = np;
    free(eids);
    return ORC_OK;
}

/* ranking.jl:46-67 */
int orc_collect_compressed(const int64_t* doclens, int64_t n_docs, const uint32_t* codes,
                           const uint8_t* residuals, int64_t rows, const int64_t* pids,
                           int64_t n_pids, uint32_t* out_codes, uint8_t* out_res) {
    int64_t* pid_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_docs + 1));
    pid_off[0] = 0;
    for (int64_t p = 0; p < n_docs; ++p) pid_off[p + 1] = pid_off[p] + doclens[p];
    int64_t w = 0;
    for (int64_t j = 0; j < n_pids; ++j) {
        if (pids[j] < 1 || pids[j] > n_docs) { free(pid_off); return ORC_EBOUNDS; }
        int64_t p = pids[j] - 1, len = doclens[p], o = pid_off[p];
        memcpy(out_codes + w, codes + o, sizeof(uint32_t) * (size_t)len);
        memcpy(out_res + w * rows, residuals + o * rows, (size_t)(len * rows));
        w += len;
    }
    free(pid_off);
    return ORC_OK;
}

/* ranking.jl:69-86 */
int orc_maxsim(const float* Q, int64_t dim, int64_t T, const float* D, int64_t n_D,
               const int64_t* pids, int64_t n_pids, const int64_t* doclens, int64_t n_docs,
               float* scores) {
    int64_t tot = 0;
    for (int64_t j = 0; j < n_pids; ++j) {
        if (pids[j] < 1 || pids[j] > n_docs) return ORC_EBOUNDS;
        tot += doclens[pids[j] - 1];
    }
    if (tot != n_D) return ORC_EDIMENSION; /* ranking.jl:71-74 */
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_pids + 1));
    off[0] = 0;
    for (int64_t j = 0; j < n_pids; ++j) off[j + 1] = off[j] + doclens[pids[j] - 1];
    int rc = ORC_OK;
    float* Qt = transpose_Q(Q, dim, T);
#pragma omp parallel if (n_D > 2048)
    {
        float* s = (float*)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
        float* m = (float*)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t j = 0; j < n_pids; ++j) {
            int64_t len = off[j + 1] - off[j];
            if (len == 0) { /* maximum over an empty slice: Julia throws */
#pragma omp atomic write
                rc = ORC_EARGUMENT;
                continue;
            }
            /* query_doc_scores = Q' * D ; maximum(.., dims=2) ; sum  (ranking.jl:76,83) */
            for (int64_t e = 0; e < len; ++e) {
                dots_all_tokens(Qt, T, dim, D + (off[j] + e) * dim, s);
                for (int64_t t = 0; t < T; ++t)
                    if (e == 0 || s[t] > m[t]) m[t] = s[t];
            }
            float acc = 0.0f;
            for (int64_t t = 0; t < T; ++t) acc = acc + m[t];
            scores[j] = acc;
        }
        free(s);
        free(m);
    }
    free(Qt);
    free(off);
    return rc;
}

/* searching.jl:102-127 */
typedef struct { float s; int64_t i; } score_idx;
static int cmp_score_desc_stable(const void* a, const void* b) {
    const score_idx* x = (const score_idx*)a; const score_idx* y = (const score_idx*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
int orc_search(int64_t dim, int nbits, int64_t K, const float* centroids, const float* weights,
               int64_t n_docs, const int64_t* doclens, int64_t n_emb, const uint32_t* codes,
               const uint8_t* residuals, const int64_t* ivf, const int64_t* ivf_lengths,
               const int64_t* emb2pid, const float* Q, int64_t T, int64_t nprobe, int64_t k,
               int64_t* out_pids, float* out_scores, int64_t* n_cand) {
    int64_t rows = dim / 8 * nbits;
    int64_t* pids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_emb > 0 ? n_emb : 1));
    int64_t np = 0;
    int rc = orc_retrieve(ivf, n_emb, ivf_lengths, K, centroids, dim, emb2pid, n_emb, nprobe, Q, T,
                          pids, &np);
    if (rc) { free(pids); return rc; }
    if (n_cand) *n_cand = np;
    int64_t ne = 0;
    for (int64_t j = 0; j < np; ++j) ne += doclens[pids[j] - 1];
    uint32_t* pc = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(ne > 0 ? ne : 1));
    uint8_t* pr = (uint8_t*)malloc((size_t)(ne * rows > 0 ? ne * rows : 1));
    float* D = (float*)malloc(sizeof(float) * (size_t)(ne * dim > 0 ? ne * dim : 1));
    float* scores = (float*)malloc(sizeof(float) * (size_t)(np > 0 ? np : 1));
    score_idx* si = NULL;
    rc = orc_collect_compressed(doclens, n_docs, codes, residuals, rows, pids, np, pc, pr);
    if (!rc) rc = orc_decompress(dim, nbits, centroids, K, weights, (int64_t)1 << nbits, pc, ne, pr, rows, ne, D);
    if (!rc) rc = orc_maxsim(Q, dim, T, D, ne, pids, np, doclens, n_docs, scores);
    if (!rc) {
        si = (score_idx*)malloc(sizeof(score_idx) * (size_t)(np > 0 ? np : 1));
        for (int64_t j = 0; j < np; ++j) { si[j].s = scores[j]; si[j].i = j; }
        qsort(si, (size_t)np, sizeof(score_idx), cmp_score_desc_stable); /* sortperm(rev=true), stable */
        if (k > np) rc = ORC_EBOUNDS; /* searching.jl:127 pids[1:k] */
        else
            for (int64_t j = 0; j < k; ++j) { out_pids[j] = pids[si[j].i]; out_scores[j] = si[j].s; }
    }
    free(si); free(scores); free(D); free(pr); free(pc); free(pids);
    return rc;
}
