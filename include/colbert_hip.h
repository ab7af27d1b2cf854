/*
 * colbert_hip.h -- C ABI of libcolbert_hip.so, the MI355X (gfx950) implementation of ColBERT.jl's
 * hot path: candidate generation + fused decompress/MaxSim + top-k (src/search, src/searching.jl),
 * the residual codec and index-build kernels (src/indexing, src/utils.jl) and the encoder epilogue
 * (src/modelling/embedding_utils.jl).
 *
 * The reference (pure Julia) has no FFI of its own; each entry point below replaces the Julia call
 * site cited next to it, and `julia/ColBERT/src/ColBERT.jl` + INTEGRATION.md show the `ccall`
 * binding a maintainer adds.  Conventions are the reference's:
 *   - matrices are column-major and densely packed (a Julia Array's memory, passed as Ptr{T});
 *   - centroid codes, pids and embedding ids are 1-based; Int is int64_t;
 *   - element types: float (Float32), uint32_t codes, uint8_t packed residuals, int64_t doclens/pids/ivf.
 * Ownership: the caller owns every host buffer; the library copies in/out before returning and never
 * retains a host pointer.  All device memory belongs to the library (handles are opaque).
 * Threading: every call is synchronous (results are in the output buffers on return) except the
 * *_device entry points, which enqueue on the given HIP stream.  One handle, one thread at a time.
 * Errors: every function returns 0 or a CLB_E* code; codes 1..4 map 1:1 onto the Julia exception the
 * reference throws in the same situation.  clb_last_error() returns a message for the calling thread.
 * There is NO CPU fallback: without a GPU every compute entry point fails with CLB_EHIP.
 */
#ifndef COLBERT_HIP_H
#define COLBERT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CLB_OK 0
#define CLB_EDIMENSION 1   /* DimensionMismatch */
#define CLB_EDOMAIN 2      /* DomainError */
#define CLB_EBOUNDS 3      /* BoundsError */
#define CLB_EARGUMENT 4    /* ArgumentError */
#define CLB_EHIP 10        /* HIP runtime failure / no device */
#define CLB_EUNSUPPORTED 11/* valid for the reference, not implemented by the HIP path */
#define CLB_ENOMEM 12

typedef struct clb_searcher clb_searcher;

const char* clb_version(void);
const char* clb_last_error(void);
/* number of visible HIP devices (0 without a GPU); never initialises a device */
int clb_device_count(void);
/* The copy rate of `device` right now: `reps` device-to-device copies of `bytes` between two HIP events, by three forms of a
 * 16-bytes-per-lane kernel and by hipMemcpyAsync; *gb_per_s = (bytes read + bytes written) / time of the fastest.  Measurement support for the roofline record of bench.py
 * ("HBM 8.0 TB/s spec, 6.29 TB/s measured copy; re-measure on the box"); no counterpart in the reference. */
int clb_measure_copy_rate(int device, int64_t bytes, int reps, double* gb_per_s);
/* ... and of a kernel that only READS `bytes` (16 bytes per lane, four pieces in flight): *gb_per_s = bytes / time.  Pass 1 of
 * the search is all reads; bench.py quotes its measured HBM traffic per second against this number. */
int clb_measure_read_rate(int device, int64_t bytes, int reps, double* gb_per_s);

/* ------------------------------------------------------------------------------------------------
 * Searcher: the resident index  (struct Searcher, src/searching.jl:1-16; Searcher(index_path) :18-80,
 * _build_emb2pid :82-91).  The arrays are the fields the reference keeps on the host; here they are
 * uploaded once into HBM.  `pid_offset` is added to every returned pid (0 for an unsharded index;
 * for a passage shard, the number of passages in the shards before it).
 * ---------------------------------------------------------------------------------------------- */
int clb_searcher_create(int device, int64_t dim, int nbits, int64_t K,
                        const float* centroids /* (dim,K) */, const float* bucket_weights /* 2^nbits */,
                        int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                        const uint32_t* codes /* n_emb, 1-based */,
                        const uint8_t* residuals /* (dim/8*nbits, n_emb) */,
                        const int64_t* ivf /* n_emb, 1-based embedding ids */,
                        const int64_t* ivf_lengths /* K */, int64_t pid_offset,
                        clb_searcher** out);
/* The same Searcher from an index that is already in HBM (built there by clb_codec_compress_device /
 * clb_build_ivf_device: Searcher(index_path) loads exactly these arrays, src/searching.jl:41-59): d_centroids, d_codes,
 * d_residuals and d_ivf are device pointers on `device`, copied into the handle (the caller may free them on return);
 * bucket_weights, doclens and ivf_lengths are host arrays as above. */
int clb_searcher_create_device(int device, int64_t dim, int nbits, int64_t K, const float* d_centroids,
                               const float* bucket_weights, int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                               const uint32_t* d_codes, const uint8_t* d_residuals, const int64_t* d_ivf,
                               const int64_t* ivf_lengths, int64_t pid_offset, clb_searcher** out);
int clb_searcher_destroy(clb_searcher* s);
/* bytes of HBM held by the handle (index + workspace) */
int64_t clb_searcher_device_bytes(const clb_searcher* s);

/* search(searcher, query, k) after the encoder  (src/searching.jl:102-127):
 * retrieve -> gather -> decompress -> maxsim -> stable sortperm(rev=true) -> first k.
 * Q is (dim, T) for one query.  out_pids/out_scores have k entries.  Fewer than k candidates is the
 * reference's BoundsError (searching.jl:127) -> CLB_EBOUNDS.  *n_cand receives the candidate count. */
int clb_search(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t k,
               int64_t* out_pids, float* out_scores, int64_t* n_cand);
/* B queries at once: Q is (dim, T, B); out_pids/out_scores are (k, B); n_cand has B entries.
 * pad_short != 0: a query with fewer than k candidates is not an error; its tail is filled with
 * pid 0 and score -Inf (what a passage shard needs before the cross-shard merge). */
int clb_search_batch(clb_searcher* s, const float* Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                     int pad_short, int64_t* out_pids, float* out_scores, int64_t* n_cand);
/* Same, device-resident: every pointer is a device pointer on the searcher's device; the work is
 * enqueued in stream order on `hip_stream` (a hipStream_t; NULL = the HIP null stream) and not waited for:
 * work queued later on that stream sees the results.
 * Always pads short results (pid 0, -Inf).  Used by the multi-GPU driver and bench.py. */
int clb_search_batch_device(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                            int64_t k, int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand,
                            void* hip_stream);
/* The same with an explicit workspace slot (0..3): several batches can be in flight on streams of the caller, each
 * on its own per-batch scratch -- the latency-bound selection kernels of one batch then overlap the scoring kernels of
 * the other (bench.py).  Calls that use the same slot must be stream-ordered by the caller.  Every other entry point
 * uses slot 0. */
int clb_search_batch_device_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand,
                                 void* hip_stream);
/* Sharded (multi-GPU) search in two calls, so that every shard cuts at the GLOBAL k-th approximate score instead
 * of its own (a shard must otherwise re-score ~k passages exactly however small it is).  Two-pass mode only
 * (CLB_EUNSUPPORTED otherwise: use clb_search_batch_device).
 *   phase 1: S1..pass 1 on this shard; d_local_top (B, k) = its k largest approximate scores per query (-Inf padded);
 *   the caller all-gathers these blocks over the shards: d_all_top = [n_shards][B][k];
 *   phase 2: selection at the k-th largest gathered score, exact pass, top-k -- outputs as clb_search_batch_device.
 * Phase 2 must follow phase 1 of the same batch on the same handle and stream. */
int clb_search_shard_phase1(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                            float* d_local_top, void* hip_stream);
int clb_search_shard_phase2(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                            const float* d_all_top, int64_t n_shards, int64_t* d_out_pids, float* d_out_scores,
                            int64_t* d_n_cand, void* hip_stream);
/* The two phases on an explicit workspace slot (0..3), like clb_search_batch_device_slot: batches that are in flight on
 * different streams of the caller MUST use different slots -- phase 2 continues on the scratch phase 1 left in its slot. */
int clb_search_shard_phase1_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, float* d_local_top, void* hip_stream);
int clb_search_shard_phase2_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, const float* d_all_top, int64_t n_shards, int64_t* d_out_pids,
                                 float* d_out_scores, int64_t* d_n_cand, void* hip_stream);
/* 0: exact single pass (every candidate scored with the canonical fp32 arithmetic);
 * 1: two-pass (bf16-MFMA approximate pass with a proven error bound selects a superset of the top-k,
 *    which is then re-scored exactly) -- results are identical by construction.  Default 1 when the
 *    index shape supports it (dim 128, nbits 2), else 0. */
int clb_searcher_set_mode(clb_searcher* s, int mode);
int clb_searcher_get_mode(const clb_searcher* s);
/* Selection step of the two-pass mode (the k-th approximate score and the list it cuts; part of search(), no call site of
 * its own in the reference): one work-group per query, or -- "wide" -- 16 work-groups per query with one launch per radix
 * pass, for shards whose queries have far more candidates than one work-group holds in registers (10 M passages on one
 * GPU: ~96 k per query).  on = -1: chosen by the handle's candidate capacity (default), 0: never, 1: always.  The
 * results are identical either way. */
int clb_searcher_set_wide_select(clb_searcher* s, int on);
/* Gather form of pass 1 of the two-pass mode (how the fp16 centroid-score rows of ranking.jl:27 reach the scorer; no call
 * site of its own in the reference): 0 = per-lane VGPR loads, 1 = LDS-DMA with four adjacent lanes per row, -1 = chosen at
 * load from the index's own code statistics (default).  Results are identical either way.  The getter returns the form
 * in use and, through *adjacency (may be null), the statistic: the fraction of consecutive embeddings whose score rows
 * share a 128-byte line. */
int clb_searcher_set_pass1_gather(clb_searcher* s, int form);
int clb_searcher_get_pass1_gather(const clb_searcher* s, double* adjacency);
/* Row format of the centroid-score table pass 1 gathers from, for batches of 16 or more queries in the two-pass mode (the
 * (32, K) cells matrix of ranking.jl:27-30; no call site of its own in the reference): form = 0, 64-byte rows of fp16 scores;
 * form = 1, 32-byte rows of 8-bit cells, linear per (query, token) over the range that token's K scores span (measured by the
 * centroid kernel; the fp16 table is requantised), with half a step in the error bound -- half the gathered bytes, pass 1
 * 12-20 % faster where the codes are not id-adjacent, but about twice the rows re-scored exactly: measured end to end it
 * loses on every workload of bench.py, so form = -1, the default, is 0.  Returned pids and scores are identical either
 * way; on the shards of a group set the form alike on all shards.  The getter returns the form batches of 16+ queries take. */
int clb_searcher_set_score_rows(clb_searcher* s, int form);
int clb_searcher_get_score_rows(const clb_searcher* s);
/* Products of the batched centroid stage (the Q x centroids GEMM of ranking.jl:9-13 for batches of 16 or more queries in
 * the two-pass mode; no call site of its own in the reference): n = 1, the fp16 score table of pass 1 is made from ONE fp16
 * product per fp32 product, with the measured conversion errors of both operands in the error bound; n = 3, from the
 * three-product bf16 split (every smaller batch always is); n = -1, the default: 1 on a shard of a group (a handle that
 * clb_searcher_set_bound_consts / clb_searcher_sync_bound_consts has been called on), 3 on a single GPU -- measured on the
 * 1 M-passage workload the single product takes 0.020 ms off the centroid kernel (0.104 -> 0.084) and its wider bound
 * (+12 % re-scored rows) puts 0.01-0.02 ms back on pass 2; on N shards the centroid stage is replicated and the extra rows
 * are divided by N.  n = 1 on an index with a centroid component outside the fp16 range stays at 3.  Results are
 * identical either way (the nprobe best centroids are re-scored in canonical fp32, the table only feeds the approximate pass,
 * whose bound carries the difference).  The getter returns the count in use for such batches and, through *max_f16_error
 * (may be null), max over the centroids of ||c - fp16(c)|| (0: out of range). */
int clb_searcher_set_centroid_products(clb_searcher* s, int n);
int clb_searcher_get_centroid_products(const clb_searcher* s, float* max_f16_error);
/* Constants of the two-pass error bound of this handle: consts[0] = max ||centroid||, [1] = sqrt(dim) * max |bucket
 * weight|, [2] = max over the shard's embeddings of 1/(||c + r|| + eps), [3] = max ||bf16-rounded residual vector||,
 * [4] = sqrt(dim) * max |w - bf16(w)|, [5] = the quantisation error of the packed inv_norm.  Sharded search with a global threshold (clb_search_shard_phase1/2) needs ONE
 * bound on every shard: take the element-wise maximum over the shards (an all-reduce MAX of six floats at load time)
 * and set it on each handle.  `set` never lowers a value. */
int clb_searcher_get_bound_consts(const clb_searcher* s, float* consts /* 6 */);
int clb_searcher_set_bound_consts(clb_searcher* s, const float* consts /* 6 */);

/* retrieve()  (src/search/ranking.jl:23-44) on its own -- test hook.  out_pids needs n_docs entries. */
int clb_retrieve(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t* out_pids,
                 int64_t* n_out);

/* Test hook for the two-pass mode: for one query returns every candidate pid with its approximate (pass 1) and
 * exact score, the selection threshold tau (k-th largest approximate score), the proven error bound eps and the
 * number of candidates with approx >= tau - 2 eps (those the exact pass re-scores). */
int clb_debug_scores(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t k, int64_t cap,
                     int64_t* out_pids, float* out_approx, float* out_exact, int64_t* n_out, float* tau,
                     float* eps, int64_t* n_rescore);

/* Merge per-shard top-k lists (device pointers): `n_lists` lists of k (pid, score) records per query,
 * each sorted by (score desc, pid asc) and padded with (0, -Inf).  Input layout [n_lists][B][k] (k
 * fastest) -- what an all-gather of every rank's (k, B) result produces; output (k, B).  Replaces the
 * final sortperm of search() across passage shards (searching.jl:125-127). */
int clb_merge_topk_device(int device, const int64_t* d_pids, const float* d_scores, int64_t k,
                          int64_t n_lists, int64_t B, int64_t* d_out_pids, float* d_out_scores,
                          void* hip_stream);
/* The same merge over PACKED per-rank blocks, so that one all-gather moves a rank's whole result: block r of
 * `d_packed` is clb_packed_topk_bytes(k, B) bytes = [B*k int64 pids][B*k fp32 scores][pad to 8 bytes].  A rank
 * produces its block by pointing clb_search_batch_device's d_out_pids / d_out_scores into one such buffer. */
int64_t clb_packed_topk_bytes(int64_t k, int64_t B);
int clb_merge_topk_packed_device(int device, const void* d_packed, int64_t k, int64_t n_lists, int64_t B,
                                 int64_t* d_out_pids, float* d_out_scores, void* hip_stream);

/* Per-kernel timing with HIP events on the stream the kernels are launched on (bench.py's roofline).
 * enable, run searches, then read: names[i] (static strings), total milliseconds and launch counts.
 * Returns the number of entries written (<= cap). */
/* on = 0: off; 1: event timing only; 2: timing + the work counters read by clb_last_batch_stats (one extra
 * kernel per batch, so keep it out of timed regions). */
int clb_profile_enable(clb_searcher* s, int on);
int clb_profile_read(clb_searcher* s, const char** names, double* total_ms, int64_t* launches, int cap);
/* per-kernel work counters of the last batch: candidate passages / candidate embeddings summed over
 * the WHOLE batch of the last call (a batch above 64 queries runs as 64-query sub-batches inside the call: the counters
 * add up over them; the per-query flag / candidate-count scratch of a workspace slot holds the last sub-batch only), and
 * embeddings re-scored by the exact pass */
int clb_last_batch_stats(clb_searcher* s, int64_t* cand_docs, int64_t* cand_embs, int64_t* rescored_docs,
                         int64_t* rescored_embs);

/* ------------------------------------------------------------------------------------------------
 * Codec and ranking pieces as stand-alone calls (host buffers, run on `device`).
 * ---------------------------------------------------------------------------------------------- */
/* decompress  (src/indexing/codecs/residual.jl:759-784) ; out is (dim, n) */
int clb_decompress(int device, int64_t dim, int nbits, const float* centroids, int64_t K,
                   const float* bucket_weights, int64_t n_weights, const uint32_t* codes, int64_t n_codes,
                   const uint8_t* residuals, int64_t res_rows, int64_t res_cols, float* out);
/* maxsim  (src/search/ranking.jl:69-86) ; D is (dim, n_D) */
int clb_maxsim(int device, const float* Q, int64_t dim, int64_t T, const float* D, int64_t n_D,
               const int64_t* pids, int64_t n_pids, const int64_t* doclens, int64_t n_docs, float* scores);
/* compress_into_codes!  (residual.jl:67-81) */
int clb_compress_into_codes(int device, uint32_t* codes, int64_t n_codes, const float* centroids,
                            int64_t dim, int64_t K, const float* embs, int64_t n);
/* compress  (residual.jl:586-604) : codes + packed residuals */
int clb_compress(int device, const float* centroids, int64_t K, const float* bucket_cutoffs,
                 int64_t n_cutoffs, int64_t dim, int nbits, const float* embs, int64_t n,
                 uint32_t* codes, uint8_t* residuals);
/* _normalize_array!(X, dims=1)  (src/utils.jl:320-325) */
int clb_normalize_columns(int device, float* X, int64_t dim, int64_t n);

/* ------------------------------------------------------------------------------------------------
 * Index build  (src/utils.jl:253-318, src/indexing/collection_indexer.jl)
 * ---------------------------------------------------------------------------------------------- */
/* kmeans_gpu_onehot!  (utils.jl:253-318) with the random initial centroids (utils.jl:261) supplied by
 * the caller in `centroids` (in/out, (dim,K)).  assignments: Int32[n], 1-based. */
int clb_kmeans(int device, const float* data, int64_t dim, int64_t n, float* centroids, int64_t K,
               int64_t max_iters, float tol, int64_t point_bsize, int32_t* assignments,
               int64_t* iters_done);
/* The same iteration split for a point set sharded over several GPUs (BASELINE config 5).  Each rank
 * holds its points on its device (`clb_kmeans_shard`); per iteration it computes the un-normalised per-cluster sums
 * (dim,K) and counts of ITS points with the current centroids (clb_kmeans_shard_pass: the batch loop of utils.jl:271-300
 * over the shard, batches counted from the shard's first point), the ranks all-gather these blocks (RCCL), and
 * clb_kmeans_reduce_update adds them in RANK ORDER -- total = ((p_0 + p_1) + p_2) + ..., deterministic and identical
 * on every rank -- and applies utils.jl:302-314: new = total ./ max.(counts,1), delta = max|old - new|; `delta < tol`
 * leaves `centroids` untouched and sets *converged.  With one shard the result equals clb_kmeans bit for bit. */
typedef struct clb_kmeans_shard clb_kmeans_shard;
int clb_kmeans_shard_create(int device, const float* data /* (dim, n) */, int64_t dim, int64_t n, int64_t K,
                            int64_t point_bsize, clb_kmeans_shard** out);
int clb_kmeans_shard_destroy(clb_kmeans_shard* h);
int clb_kmeans_shard_pass(clb_kmeans_shard* h, const float* centroids /* (dim,K) */, float* sums /* (dim,K) */,
                          int64_t* counts /* K */, int32_t* assignments /* n, 1-based; may be NULL */);
/* The same iteration with the exchange kept on the device (the blocks are 64 MB per rank at K = 131 072: no host round
 * trip).  The centroids live in the shard handle: set_centroids uploads the initial ones (utils.jl:261), pass_device
 * writes this rank's block -- (dim,K) fp32 sums, padded to 8 bytes, then K int64 counts: clb_kmeans_shard_block_bytes --
 * into d_block on `hip_stream`, the ranks all-gather the blocks (clb_comm_all_gather / RCCL) and update_device reduces
 * the n_ranks gathered blocks in rank order and applies utils.jl:302-314 to the handle's centroids (one 4-byte
 * read-back per iteration: delta).  Bit-identical to clb_kmeans_shard_pass + clb_kmeans_reduce_update. */
int64_t clb_kmeans_shard_block_bytes(const clb_kmeans_shard* h);
/* `centroids` may be a host or a device pointer in both calls (the runtime tells them apart) */
int clb_kmeans_shard_set_centroids(clb_kmeans_shard* h, const float* centroids /* (dim,K) */);
int clb_kmeans_shard_get_centroids(clb_kmeans_shard* h, float* centroids /* (dim,K) */);
/* A shard whose points are ALREADY in HBM (the clustering sample of a collection too large to stage through host
 * memory: 14 M x 128 fp32 at 1 M passages): d_data (dim, n) is borrowed, not copied -- the caller keeps it alive and
 * unchanged until clb_kmeans_shard_destroy. */
int clb_kmeans_shard_create_device(int device, const float* d_data /* (dim, n) device */, int64_t dim, int64_t n,
                                   int64_t K, int64_t point_bsize, clb_kmeans_shard** out);
/* the assignments of the last pass (Int32, 1-based: `assignments` of kmeans_gpu_onehot!, utils.jl:253); host or
 * device pointer, n entries */
int clb_kmeans_shard_get_assignments(clb_kmeans_shard* h, int32_t* assignments);
int clb_kmeans_shard_pass_device(clb_kmeans_shard* h, void* d_block, void* hip_stream);
int clb_kmeans_shard_update_device(clb_kmeans_shard* h, const void* d_gathered /* n_ranks blocks */, int64_t n_ranks,
                                   float tol, float* delta_out, int* converged, void* hip_stream);
int clb_kmeans_reduce_update(int device, float* centroids /* (dim,K) in/out */,
                             const float* gathered_sums /* [world][dim*K] */,
                             const int64_t* gathered_counts /* [world][K] */, int64_t world, int64_t dim, int64_t K,
                             float tol, float* delta_out, int* converged);
/* _compute_avg_residuals!  (collection_indexer.jl:177-195) incl. _bucket_cutoffs_and_weights :141-152 */
int clb_compute_avg_residuals(int device, int nbits, const float* centroids, int64_t dim, int64_t K,
                              const float* heldout, int64_t n, uint32_t* codes, int64_t n_codes,
                              float* bucket_cutoffs, float* bucket_weights, float* avg_residual);
/* _build_ivf  (collection_indexer.jl:349-353) */
int clb_build_ivf(int device, const uint32_t* codes, int64_t n, int64_t K, int64_t* ivf,
                  int64_t* ivf_lengths);
/* Test hooks of the library's own stable radix sort and exclusive scan (csrc/sort.hip: the `sortperm(codes)` of
 * collection_indexer.jl:350, the quantile sort of :147-150, the load-time re-ordering of the Searcher, the general top-k):
 * host arrays in and out.  key_bits 32 = uint32 keys sorted stably on their low end_bit bits WITH uint32 values; 64 = uint64
 * keys, with or without values; -32 = float keys, ascending.  clb_debug_exclusive_scan: out[0..n] = prefix sums of in[0..n),
 * out[n] = the total (modulo 2^32). */
int clb_debug_sort(int device, int key_bits, const void* keys, const uint32_t* vals, int64_t n, int end_bit, void* keys_out,
                   uint32_t* vals_out);
int clb_debug_exclusive_scan(int device, const uint32_t* in, int64_t n, uint32_t* out);
/* The same over device arrays (80 M codes at 1 M passages never visit the host): d_codes UInt32[n] 1-based,
 * d_ivf Int64[n], d_ivf_lengths Int64[K], all on `device`; runs on `hip_stream` and waits for it (the range check of
 * counts(values, K) has to be read back: CLB_EBOUNDS). */
int clb_build_ivf_device(int device, const uint32_t* d_codes, int64_t n, int64_t K, int64_t* d_ivf,
                         int64_t* d_ivf_lengths, void* hip_stream);

/* The chunk loop of index()  (src/indexing.jl:102-118, collection_indexer.jl:271-297: encode a chunk of passages ->
 * compress -> save) with the codec resident: the centroids (64 MB at K = 131 072), their bf16 split and the
 * nearest-centroid scratch are uploaded / allocated once instead of once per chunk, and a chunk's embeddings, codes and
 * residuals are device arrays -- the encoder's output (clb_encode_docs on the device) or a generator's.
 * clb_codec_create: `centroids` (dim,K) and `bucket_cutoffs` (2^nbits - 1) may be host or device pointers.
 * clb_codec_compress_device = compress (residual.jl:586-604) for n embeddings d_embs (dim, n): d_codes UInt32[n] 1-based,
 * d_residuals UInt8 (dim/8*nbits, n); enqueued on hip_stream, not waited for.  Bit-identical to clb_compress. */
typedef struct clb_codec clb_codec;
int clb_codec_create(int device, int64_t dim, int nbits, int64_t K, const float* centroids,
                     const float* bucket_cutoffs, int64_t n_cutoffs, clb_codec** out);
int clb_codec_destroy(clb_codec* c);
int clb_codec_compress_device(clb_codec* c, const float* d_embs, int64_t n, uint32_t* d_codes, uint8_t* d_residuals,
                              void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Device arrays for a host that has none of its own (the Julia shim; Python uses torch tensors for the same purpose).
 * The device-resident route of index() (src/indexing.jl:63-147 with the sample, the codes and the residuals kept in HBM:
 * clb_encode_docs_packed_device -> clb_gather_rows_device -> clb_kmeans_shard_create_device -> clb_codec_compress_device ->
 * clb_build_ivf_device -> clb_searcher_create_device) needs to allocate, fill and read back plain device buffers and to
 * cut the sampled / shuffled rows out of a device matrix -- the reference's `sample[:, randperm(...)]` and
 * `embs[:, sampled columns]` (collection_indexer.jl:56-91) -- without the embeddings crossing PCIe.
 * ---------------------------------------------------------------------------------------------- */
int clb_device_malloc(int device, int64_t bytes, void** d_out);
int clb_device_free(int device, void* d_ptr);
/* blocking copies between host memory and a buffer of clb_device_malloc (any device pointer on `device` will do) */
int clb_device_upload(int device, void* d_dst, const void* src, int64_t bytes);
int clb_device_download(int device, void* dst, const void* d_src, int64_t bytes);
int clb_device_synchronize(int device);
/* free / total bytes of HBM on `device` (index() picks the device-resident route only if its footprint fits) */
int clb_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes);
/* d_dst[i, :] = d_src[d_rows[i], :] for i < n: rows of `row_bytes` bytes (a multiple of 4; column j of Julia's (dim, n_src)
 * Float32 matrix is row j here), d_rows Int64[n] on the device, 0-BASED, each in [0, n_src).  Runs on `hip_stream` and
 * waits for it: an index outside the range is CLB_EBOUNDS (nothing is read outside d_src; the destination row is zeroed).
 * d_dst must not overlap d_src. */
int clb_gather_rows_device(int device, const void* d_src, int64_t n_src, int64_t row_bytes, const int64_t* d_rows,
                           int64_t n, void* d_dst, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Exchange step of the sharded search / index build on RCCL  (the reference is single-GPU)
 * One communicator per process and GPU.  Rank 0 calls clb_comm_unique_id and hands the bytes to the other ranks by
 * whatever the host has (a file, MPI, a socket); every rank then calls clb_comm_create -- it blocks until all ranks
 * have joined.  librccl is opened at run time; without it these calls return CLB_EHIP and nothing else is affected.
 * The Python driver may use torch.distributed instead (the same RCCL underneath); hosts without torch -- the Julia
 * shim -- use these.
 * ---------------------------------------------------------------------------------------------- */
typedef struct clb_comm clb_comm;
int64_t clb_comm_unique_id_bytes(void);
int clb_comm_unique_id(void* id, int64_t bytes);
int clb_comm_create(int device, int rank, int n_ranks, const void* id, int64_t bytes, clb_comm** out);
int clb_comm_destroy(clb_comm* c);
int clb_comm_rank(const clb_comm* c);
int clb_comm_size(const clb_comm* c);
/* all-gather of `bytes_per_rank` bytes from every rank, in rank order, into d_recv (n_ranks * bytes_per_rank), enqueued
 * on hip_stream: the packed per-shard top-k blocks (clb_packed_topk_bytes -> clb_merge_topk_packed_device), the (B, k)
 * score blocks between clb_search_shard_phase1 and phase2, the cluster sums of clb_kmeans_shard_pass. */
int clb_comm_all_gather(clb_comm* c, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* hip_stream);
/* element-wise maximum over the ranks, in place (the six bound constants of clb_searcher_get/set_bound_consts) */
int clb_comm_all_reduce_max_f32(clb_comm* c, float* d_buf, int64_t n, void* hip_stream);
/* The whole round trip the two-phase sharded search needs once per shard group, inside the library (a host without its
 * own device arrays -- the Julia shim -- cannot hand clb_comm_all_reduce_max_f32 a device pointer):
 * clb_searcher_get_bound_consts -> all-reduce MAX over the communicator -> clb_searcher_set_bound_consts.  Collective:
 * every rank of `c` calls it with its shard's handle.  Blocks until done. */
int clb_searcher_sync_bound_consts(clb_searcher* s, clb_comm* c);

/* ------------------------------------------------------------------------------------------------
 * Encoder  (src/modelling/checkpoint.jl)
 * ---------------------------------------------------------------------------------------------- */
typedef struct clb_encoder clb_encoder;
/* The BERT + Dense weights of a ColBERT checkpoint (what load_hgf_pretrained_local returns,
 * src/local_loading.jl:139-209) as ONE flat fp32 blob, torch Linear layout [out][in]:
 *   word_emb [vocab][H], pos_emb [max_pos][H], type_emb [type_vocab][H], emb_ln_gamma [H], emb_ln_beta [H],
 *   per layer: Wq,Wk,Wv [3H][H], bq,bk,bv [3H], Wo [H][H], bo [H], ln1_gamma, ln1_beta [H],
 *              W1 [I][H], b1 [I], W2 [H][I], b2 [H], ln2_gamma, ln2_beta [H],
 *   linear_W [dim][H], linear_b [dim].
 * tools/export_checkpoint.py writes it from a HuggingFace directory. */
int clb_encoder_create(int device, int64_t vocab, int64_t hidden, int64_t layers, int64_t heads,
                       int64_t intermediate, int64_t max_pos, int64_t type_vocab, int64_t dim, float ln_eps,
                       const float* weights, int64_t n_weights, clb_encoder** out);
int clb_encoder_destroy(clb_encoder* e);
/* Arithmetic of the Linear layers (the reference multiplies in Float32, checkpoint.jl:21-25 through Transformers.jl).
 * Every fp32 operand is split into 16-bit planes and the product is a sum of exact MFMA plane products in an fp32
 * accumulator.  mode 3 (default) = "f16x3": two fp16 planes of the operand scaled by a power of two (round-to-nearest makes
 * two fp16 planes hold all 24 significant bits), three products, error per product < 2^-23 |a||b|; fp16's exponent range
 * limits the activations to |x| < 4 094 (an encode that leaves it reports CLB_EDOMAIN: non-finite output).
 * mode 2 = "bf16x6": three bf16 planes, six products, < 2^-22 per product over the whole fp32 range; mode 1 = "bf16x3": two
 * bf16 planes, three products, < 2^-15; mode 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the 16-bit rate). */
int clb_encoder_set_gemm_mode(clb_encoder* e, int mode);
/* Self-attention for head size 64 and up to 512 positions: mode 0 (default) = fused in registers -- behind the f16x3 Linear
 * layers on the 16-bit matrix pipe, on fp16 planes the Q/K/V projection writes for it (three exact products per fp32
 * product; batches of more than 64 tokens), else fp32 MFMA (all score tiles resident up to 64 keys, online softmax beyond);
 * 1 = fp32 MFMA, register-resident for every length; 2 = the three-kernel path (scores in memory; always taken for other head
 * sizes); 3 = mode 0 on the fp32 MFMA whatever the GEMM mode; 5 = mode 0 with the K / V tiles of a (sequence, head) staged once
 * in LDS for all its query blocks instead of every wave loading its own (round 5: bit-identical to 0, measured slower) --
 * 1 to 5 exist for comparison. */
int clb_encoder_set_attention_mode(clb_encoder* e, int mode);
/* LayerNorm folded around the Linear layers (f16x3 only): the Linear that produces a LayerNorm's input stores the raw rows and
 * their partial (mean, M2); the Linear that consumes it multiplies the raw rows with gamma (.) W and applies
 * rstd (a . (gamma (.) W)^T - mean u) + c in its epilogue -- no stand-alone LayerNorm pass (48.7 us x 24 per 64 x 300 passage
 * batch).  Same function of the inputs as `doc` (src/modelling/checkpoint.jl:21-25), different rounding.
 * mode: -1 = batches too long to split over K (the default: a 64 x 300 passage batch 14.34 -> 13.93 ms,
 * profiles/r05_experiments.md), 0 = never, 1 = always (tests). */
int clb_encoder_set_ln_fold(clb_encoder* e, int mode);
/* doc(bert, linear, integer_ids, bitmask)  (checkpoint.jl:21-25): integer_ids Int32 (L, N), 1-based token ids;
 * bitmask (L, N) 0/1 bytes = attention (key) mask; out Float32 (dim, L, N). */
int clb_encode(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N, float* out);
/* _doc_embeddings_and_doclens  (checkpoint.jl:27-52): forward, clear skiplist tokens, normalise, doclens,
 * compaction.  out_embs (dim, <= L*N); doclens Int64[N]; *n_out = kept columns.
 * When every unattended token (bitmask 0) is one the skiplist drops -- the [PAD] padding of tensorize_docs is -- and the
 * encoder runs the fp16-plane attention, the rows the output never sees are not computed (the batch is packed on the way to
 * the device, attended tokens keep their positions): same outputs up to the rounding of a different tile plan, about twice
 * the throughput on batches padded to their longest passage.  Otherwise the whole (L, N) batch is computed. */
int clb_encode_docs(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N,
                    const int64_t* skiplist, int64_t n_skip, float* out_embs, int64_t* doclens, int64_t* n_out);
/* _query_embeddings  (checkpoint.jl:54-71): forward, clear skiplist tokens, normalise.  out (dim, L, N). */
int clb_encode_queries(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N,
                       const int64_t* skiplist, int64_t n_skip, float* out);

/* _query_embeddings with every pointer on the encoder's device, enqueued on `hip_stream` and not waited for: the
 * (dim, L, N) output is what clb_search_batch_device takes as d_Q, so encode_queries + search (src/searching.jl:93-127)
 * run back to back without leaving HBM.  d_skiplist: n_skip Int64 ids on the device.  An id outside the vocabulary
 * cannot be reported from an asynchronous call: it is clamped (use clb_encode_queries to validate inputs). */
int clb_encode_queries_device(clb_encoder* e, const int32_t* d_integer_ids, const uint8_t* d_bitmask, int64_t L, int64_t N,
                              const int64_t* d_skiplist, int64_t n_skip, float* d_out, void* hip_stream);
/* _doc_embeddings_and_doclens  (checkpoint.jl:27-52) with every pointer on the encoder's device, enqueued on `hip_stream`
 * and not waited for: forward, skiplist mask, normalisation, doclens and compaction without a host round trip -- the chunk
 * loop of index() (src/indexing.jl:102-118) then hands the embeddings of a batch straight to clb_codec_compress_device.
 * d_out_embs: capacity (dim, L*N); the first *d_n_out columns are valid (read d_n_out / d_doclens back when needed).
 * Ids outside the vocabulary are clamped and reported by clb_encoder_check_last_ids, like clb_encode_queries_device. */
int clb_encode_docs_device(clb_encoder* e, const int32_t* d_integer_ids, const uint8_t* d_bitmask, int64_t L, int64_t N,
                           const int64_t* d_skiplist, int64_t n_skip, float* d_out_embs, int64_t* d_doclens,
                           int64_t* d_n_out, void* hip_stream);
/* The same for a PACKED batch -- N passages back to back without padding rows: d_ids[rows] token ids (1-based), d_pos[rows]
 * the position of every token in its passage (0-based), d_seq[rows] its passage (0..N-1), d_cu[N + 1] the row offsets
 * (d_cu[N] = rows), Lmax the longest passage.  tensorize_docs (doc_tokenization.jl:143-156) pads a batch to its longest
 * passage with [PAD], which the attention mask hides and the skiplist drops (checkpoint.jl:37-43): the padding rows never
 * reach the output, and with passages of ~80 tokens in batches padded to ~160 they are half of the encoder's work.  Here
 * they are not computed.  Every row counts as attended.  Needs the fp16-plane attention (head size 64, GEMM mode 3,
 * attention mode 0): CLB_EARGUMENT otherwise -- the caller then pads.  Outputs as clb_encode_docs_device. */
int clb_encode_docs_packed_device(clb_encoder* e, const int32_t* d_ids, const int32_t* d_pos, const int32_t* d_seq,
                                  const int32_t* d_cu, int64_t N, int64_t Lmax, int64_t rows, const int64_t* d_skiplist,
                                  int64_t n_skip, float* d_out_embs, int64_t* d_doclens, int64_t* d_n_out, void* hip_stream);
/* The asynchronous device path cannot report an id outside the vocabulary when it is enqueued (it clamps): this call
 * waits for the device and returns the BoundsError of ANY device-path encode since the previous check (the flag is
 * sticky: set by the kernels, cleared by this call), or CLB_EDOMAIN when an encode produced non-finite embeddings (the
 * f16x3 split's range).  The host-buffer entry points report both themselves. */
int clb_encoder_check_last_ids(clb_encoder* e);
/* The sticky flag itself: *d_flag receives the device address of the int32 the encode kernels OR their findings into
 * (bit 0: id outside the vocabulary, bit 1: non-finite output).  A serving loop copies its 4 bytes back together with the
 * results of a query (search(searcher, query::String, k), src/searching.jl:93-127) and calls clb_encoder_check_last_ids only
 * when it is non-zero -- an asynchronous encode whose activations leave the f16 split's range would otherwise hand NaN
 * scores to the caller without an error.  The address is stable for the life of the handle; valid after the first encode
 * or this call (which allocates and clears the flag if no encode has run yet). */
int clb_encoder_error_flag_device(clb_encoder* e, void** d_flag);
/* Per-stage HIP-event timing of the encoder forward (bench.py's encoder roofline; the stages are the Linear layers of
 * `doc`, src/modelling/checkpoint.jl:21-25, by role).  enable, run encodes, then read: names[i] (static strings), total
 * milliseconds and stage executions since the last read.  Returns the number of entries written (<= cap), -1 on error. */
int clb_encoder_profile_enable(clb_encoder* e, int on);
int clb_encoder_profile_read(clb_encoder* e, const char** names, double* total_ms, int64_t* launches, int cap);


/* Encoder epilogue as stand-alone calls  (src/modelling/checkpoint.jl:27-71, embedding_utils.jl:172-205) */
/* _doc_embeddings_and_doclens after doc(): clear skiplist tokens, normalise, doclens, compaction.
 * D (dim, L, N) is read only; out (dim, <= L*N); doclens Int64[N]; *n_out = kept columns. */
int clb_doc_epilogue(int device, const float* D, int64_t dim, int64_t L, int64_t N,
                     const int32_t* integer_ids, const int64_t* skiplist, int64_t n_skip, float* out,
                     int64_t* doclens, int64_t* n_out);
/* _query_embeddings after doc(): clear skiplist tokens, normalise (in place). */
int clb_query_epilogue(int device, float* Q, int64_t dim, int64_t L, int64_t N,
                       const int32_t* integer_ids, const int64_t* skiplist, int64_t n_skip);

#ifdef __cplusplus
}
#endif
#endif
