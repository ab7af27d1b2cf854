/*
 * colbert_oracle.h -- CPU restatement of ColBERT.jl's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X build.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path
 * (colbert.jl_amd/csrc -> libcolbert_hip.so) never links, loads or calls anything in oracle/.
 *
 * PINNING STATUS.  The reference is pure Julia and there is no `julia` binary in the build image
 * (nor on the GPU box), so the reference itself cannot be executed.  The oracle is pinned against
 * every known-answer vector the reference's own test-suite holds for this path (transcribed as data
 * in tests/golden/reference_kats.json; see tests/test_oracle_golden.py).  Functions whose values no
 * reference test pins (decompress values, kmeans on non-degenerate data, the final sortperm of
 * `search`) are pinned by the restatement alone -- DESIGN.md lists them as "parity unpinned".
 *
 * CONVENTIONS (all taken from the Julia reference):
 *   - every matrix is column-major, densely packed, exactly as a Julia Array would be;
 *   - centroid codes, pids, embedding ids are 1-based on every interface;
 *   - Int == int64_t, UInt32 codes, UInt8 packed residuals, Float32 values.
 *
 * CANONICAL ARITHMETIC ORDER.  Julia's `*` (OpenBLAS sgemm) and `sum` (@simd) leave the fp32
 * summation order unspecified, so "the reference result" is only defined up to ~1e-7 relative.
 * The oracle fixes one order, which the HIP kernels reproduce bit-for-bit:
 *   dot(a,b)      acc = +0; for d = 0..dim-1 ascending: acc = fmaf(a[d], b[d], acc)
 *   sumsq(x)      four interleaved partial sums p[g] += x[d]*x[d] (d = g mod 4, ascending, product
 *                 rounded before the add, as `sum(abs2, ...)` does), combined as (p0+p1)+(p2+p3)
 *   normalise     x[d] / (sqrtf(sumsq(x)) + FLT_EPSILON)         (src/utils.jl:320-325)
 *   maxsim        per token max over the document's embeddings, then a sequential fp32 sum over
 *                 tokens t = 0..T-1                                 (src/search/ranking.jl:69-86)
 * Every function returns 0 on success or one of the ORC_E* codes, which map 1:1 onto the Julia
 * exception the reference throws in the same situation.
 */
#ifndef COLBERT_ORACLE_H
#define COLBERT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_EDIMENSION 1 /* DimensionMismatch */
#define ORC_EDOMAIN 2    /* DomainError       */
#define ORC_EBOUNDS 3    /* BoundsError       */
#define ORC_EARGUMENT 4  /* ArgumentError     */

/* ---- src/utils.jl ------------------------------------------------------------------------- */
float orc_dot(const float* a, const float* b, int64_t dim);
float orc_sumsq(const float* x, int64_t dim);
/* _normalize_array!(X, dims=1)  utils.jl:320-325 ; X is (dim, n) */
int orc_normalize_columns(float* X, int64_t dim, int64_t n);
/* _normalize_array!(X, dims=2): every row of the (rows, cols) matrix */
int orc_normalize_rows(float* X, int64_t rows, int64_t cols);
/* _topk(data, k, dims)  utils.jl:327-332 ; data (rows, cols) col-major.
 * dims==2: out is (rows, k) col-major, 1-based column indices; dims==1: out is (k, cols). */
int orc_topk(const float* data, int64_t rows, int64_t cols, int64_t k, int dims, int64_t* out);
/* compute_distances_kernel!  utils.jl:38-59 ; dist is (K, b) */
int orc_kmeans_distances(float* dist, int64_t dist_rows, int64_t dist_cols, const float* data,
                         int64_t data_dim, int64_t b, const float* centroids, int64_t cent_dim,
                         int64_t K);
/* assign_clusters_kernel!  utils.jl:71-79 ; assignments 1-based Int32 */
int orc_kmeans_assign(int32_t* assign, int64_t n_assign, const float* dist, int64_t K, int64_t b);
/* onehot_encode!  utils.jl:81-89 ; one_hot is (k, b), only sets ones */
int orc_onehot_encode(float* one_hot, int64_t oh_rows, int64_t oh_cols, const int32_t* assign,
                      int64_t b, int64_t k);
/* update_centroids_kernel!  utils.jl:61-69 ; new_centroids(dim,K) += data(dim,b) * one_hot(K,b)' */
int orc_kmeans_update(float* new_centroids, int64_t nc_rows, int64_t nc_cols, const float* data,
                      int64_t dim, int64_t b, const float* one_hot, int64_t oh_rows,
                      int64_t oh_cols);
/* kmeans_gpu_onehot!  utils.jl:253-318, with the random initialisation (utils.jl:261) replaced by
 * the caller-supplied `centroids` (in/out).  Returns the iterations executed in *iters_done. */
int orc_kmeans(const float* data, int64_t dim, int64_t n, float* centroids, int64_t K,
               int64_t max_iters, float tol, int64_t point_bsize, int32_t* assignments,
               int64_t* iters_done);

/* One k-means iteration (utils.jl:271-314) split for passage/point shards -- the canonical order of the multi-GPU
 * build: shard pass = the batch loop over the shard's own points (un-normalised sums (dim,K), counts[K], 1-based
 * assignments or NULL); reduce_update = sums added in rank order, ./ max(count,1), delta, `delta < tol` keeps the
 * old centroids (*converged = 1). */
int orc_kmeans_shard_pass(const float* data, int64_t dim, int64_t n, const float* centroids, int64_t K,
                          int64_t point_bsize, float* sums, int64_t* counts, int32_t* assignments);
int orc_kmeans_reduce_update(float* centroids, const float* gathered_sums, const int64_t* gathered_counts,
                             int64_t world, int64_t dim, int64_t K, float tol, float* delta_out,
                             int* converged);

/* ---- src/indexing/codecs/residual.jl --------------------------------------------------------- */
/* compress_into_codes!  residual.jl:67-81 ; argmax inner product, first index on ties, 1-based */
int orc_compress_into_codes(uint32_t* codes, int64_t n_codes, const float* centroids, int64_t dim,
                            int64_t K, const float* embs, int64_t n);
/* _binarize  residual.jl:197-208 ; data (dim,b) ints -> bits (nbits,dim,b) as 0/1 bytes */
int orc_binarize_bits(const int64_t* data, int64_t dim, int64_t b, int nbits, uint8_t* bits);
/* _unbinarize  residual.jl:233-240 */
int orc_unbinarize(const uint8_t* bits, int nbits, int64_t dim, int64_t b, int64_t* out);
/* _bucket_indices  residual.jl:348-351 ; searchsortedfirst(cutoffs, x) - 1 */
int orc_bucket_indices(const float* data, int64_t n, const float* cutoffs, int64_t ncut,
                       int64_t* out);
/* _packbits  residual.jl:400-407 ; bits (nbits,dim,b) -> (dim/8*nbits, b) */
int orc_packbits(const uint8_t* bits, int nbits, int64_t dim, int64_t b, uint8_t* out);
/* _unpackbits  residual.jl:428-441 ; packed (rows,b) -> bits (nbits, rows/nbits*8, b) */
int orc_unpackbits(const uint8_t* packed, int64_t rows, int64_t b, int nbits, uint8_t* bits);
/* binarize  residual.jl:518-536 */
int orc_binarize(int64_t dim, int nbits, const float* cutoffs, int64_t ncut, const float* residuals,
                 int64_t b, uint8_t* out);
/* compress  residual.jl:586-604 */
int orc_compress(const float* centroids, int64_t K, const float* cutoffs, int64_t ncut, int64_t dim,
                 int nbits, const float* embs, int64_t n, uint32_t* codes, uint8_t* residuals);
/* decompress_residuals  residual.jl:698-721 */
int orc_decompress_residuals(int64_t dim, int nbits, const float* weights, int64_t nweights,
                             const uint8_t* packed, int64_t rows, int64_t b, float* out);
/* decompress  residual.jl:759-784 */
int orc_decompress(int64_t dim, int nbits, const float* centroids, int64_t K, const float* weights,
                   int64_t nweights, const uint32_t* codes, int64_t n_codes, const uint8_t* residuals,
                   int64_t res_rows, int64_t res_cols, float* out);

/* ---- src/indexing/collection_indexer.jl ---------------------------------------------------- */
/* _sample_pids count  collection_indexer.jl:17-20 (the RNG draw itself cannot be matched) */
int64_t orc_num_sampled_pids(int64_t num_documents);
/* _heldout_split size  collection_indexer.jl:85-86 */
int64_t orc_heldout_size(int64_t num_sample_embs, float heldout_fraction);
/* setup  collection_indexer.jl:115-139 */
int orc_setup(int64_t num_documents, float avg_doclen_est, int64_t num_clustering_embs,
              int64_t chunksize /* <=0: missing */, int64_t nranks, int64_t* out_chunksize,
              int64_t* out_num_chunks, int64_t* out_num_partitions, double* out_num_embeddings_est);
/* _bucket_cutoffs_and_weights  collection_indexer.jl:141-152 ; `values` is overwritten (sorted) */
int orc_bucket_cutoffs_and_weights(int nbits, float* values, int64_t n, float* cutoffs,
                                   float* weights);
/* _compute_avg_residuals!  collection_indexer.jl:177-195 */
int orc_compute_avg_residuals(int nbits, const float* centroids, int64_t dim, int64_t K,
                              const float* heldout, int64_t n, uint32_t* codes, int64_t n_codes,
                              float* cutoffs, float* weights, float* avg_residual);
/* _collect_embedding_id_offset  collection_indexer.jl:342-347 ; offsets has max(n,1) entries */
int orc_collect_embedding_id_offset(const int64_t* counts, int64_t n, int64_t* total,
                                    int64_t* offsets);
/* _build_ivf  collection_indexer.jl:349-353 */
int orc_build_ivf(const uint32_t* codes, int64_t n, int64_t K, int64_t* ivf, int64_t* ivf_lengths);

/* ---- src/modelling/embedding_utils.jl, checkpoint.jl (encoder epilogue) --------------------- */
/* mask_skiplist!  embedding_utils.jl:172-177 ; mask, ids are (L, N) */
int orc_mask_skiplist(uint8_t* mask, const int32_t* ids, int64_t count, const int64_t* skiplist,
                      int64_t nskip);
/* _doc_embeddings_and_doclens epilogue  checkpoint.jl:30-51 : clear skiplist tokens, normalise,
 * doclens = sum(mask, dims=1), keep unmasked columns.  D (dim, L, N) is modified in place;
 * out (dim, sum(doclens)) ; doclens[N].  Returns number of kept columns in *n_out. */
int orc_doc_epilogue(float* D, int64_t dim, int64_t L, int64_t N, const int32_t* ids,
                     const int64_t* skiplist, int64_t nskip, float* out, int64_t* doclens,
                     int64_t* n_out);
/* _query_embeddings epilogue  checkpoint.jl:61-69 */
int orc_query_epilogue(float* Q, int64_t dim, int64_t L, int64_t N, const int32_t* ids,
                       const int64_t* skiplist, int64_t nskip);

/* ---- src/search/ranking.jl, src/searching.jl ---------------------------------------------- */
/* _build_emb2pid  searching.jl:82-91 */
int orc_build_emb2pid(const int64_t* doclens, int64_t n_docs, int64_t* emb2pid);
/* _cids_to_eids!  ranking.jl:7-21 */
int orc_cids_to_eids(int64_t* eids, int64_t n_eids, const int64_t* cids, int64_t n_cids,
                     const int64_t* ivf, int64_t n_ivf, const int64_t* ivf_lengths, int64_t K);
/* retrieve  ranking.jl:23-44 ; out_pids needs room for n_emb entries at most */
int orc_retrieve(const int64_t* ivf, int64_t n_ivf, const int64_t* ivf_lengths, int64_t K,
                 const float* centroids, int64_t dim, const int64_t* emb2pid, int64_t n_emb,
                 int64_t nprobe, const float* Q, int64_t T, int64_t* out_pids, int64_t* n_out);
/* _collect_compressed_embs_for_pids  ranking.jl:46-67 */
int orc_collect_compressed(const int64_t* doclens, int64_t n_docs, const uint32_t* codes,
                           const uint8_t* residuals, int64_t rows, const int64_t* pids,
                           int64_t n_pids, uint32_t* out_codes, uint8_t* out_res);
/* maxsim  ranking.jl:69-86 */
int orc_maxsim(const float* Q, int64_t dim, int64_t T, const float* D, int64_t n_D,
               const int64_t* pids, int64_t n_pids, const int64_t* doclens, int64_t n_docs,
               float* scores);
/* search, after the encoder  searching.jl:102-127 : retrieve -> collect -> decompress -> maxsim ->
 * stable sortperm(rev=true) -> first k.  Structured like the reference (every intermediate is
 * materialised).  *n_cand receives the candidate count.  ORC_EBOUNDS when fewer than k. */
int orc_search(int64_t dim, int nbits, int64_t K, const float* centroids, const float* weights,
               int64_t n_docs, const int64_t* doclens, int64_t n_emb, const uint32_t* codes,
               const uint8_t* residuals, const int64_t* ivf, const int64_t* ivf_lengths,
               const int64_t* emb2pid, const float* Q, int64_t T, int64_t nprobe, int64_t k,
               int64_t* out_pids, float* out_scores, int64_t* n_cand);
/* number of OpenMP threads the oracle will use (for the cpu_baseline report) */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
