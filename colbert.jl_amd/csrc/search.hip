// search.hip -- the Searcher handle and the search entry points of the C ABI (include/colbert_hip.h).
// Replaces: struct Searcher / Searcher(index_path) (src/searching.jl:1-91) and search() after the
// encoder (src/searching.jl:102-127), retrieve/gather/maxsim (src/search/ranking.jl).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "approx_kernels.hpp"
#include "generic_kernels.hpp"
#include "search_kernels.hpp"
#include "sort.hpp"

using namespace clb;

namespace {

enum KernelId {
    KID_CENTROID_SCORES = 0,
    KID_TOPN,
    KID_MARK,
    KID_COMPACT,
    KID_SCORE_EXACT,
    KID_SCORE_APPROX,
    KID_SELECT,
    KID_ROWS,
    KID_TOPK,
    KID_COUNT
};
const char* kKernelNames[KID_COUNT] = {"centroid_scores", "top_nprobe", "mark_candidates", "compact_candidates",
                                       "score_exact",     "score_approx", "select_margin", "rescore_rows",
                                       "topk"};

struct Prof {
    bool on = false;        // HIP-event timing of every kernel
    bool counters = false;  // additionally count the work of each batch (one extra kernel per batch)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[KID_COUNT];
    std::vector<hipEvent_t> pool;
    // end event of the previous timed kernel: the next timed kernel on the same stream starts from it instead of
    // recording its own start (half the event records per batch).  Cleared wherever untimed work is enqueued.
    hipEvent_t chain = nullptr;
    hipStream_t chain_stream = nullptr;
    double total_ms[KID_COUNT] = {0};
    int64_t launches[KID_COUNT] = {0};
    bool failed = false;    // an event could not be created: that kernel goes untimed, clb_profile_read reports it
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { failed = true; return nullptr; }
        return e;
    }
};

}  // namespace

// Per-batch workspace (everything a batch of queries needs besides the resident index).
struct Workspace {
    int64_t Bcap = 0, Tcap = 0, npcap = 0, kcap = 0;
    int64_t Ttuned = 0;         // largest query length <= 128 this slot has seen (what the tuned-path buffers are sized for)
    size_t cand_cap = 0;
    int W = 0, nblk_bitmap = 0, topn_blocks = 0;
    bool x1_table = false;      // the score table in cells_q came from the single-fp16-product kernel: the bound carries its terms
    bool have_range = false;    // tscale holds this batch's measured score ranges (the batched centroid kernel ran)
    bool cell8 = false;         // pass 1 gathers from cells8: 32-byte rows of 8-bit cells requantised from cells_q by tscale
    bool stats_keep = false;    // set for the 2nd, 3rd ... sub-batch of one call: the work counters accumulate over the call
    // two-phase sharded search: what clb_search_shard_phase1 left behind (phase 2 must continue exactly that batch)
    struct { bool valid = false; const float* dQ = nullptr; int64_t T = 0, B = 0, nprobe = 0, k = 0; void* stream = nullptr; } pending;
    DevBuf Qdev, cells, cells_q, partial, sel, bitmap, blocksum, ncand, cand, cand_hdr, scores, list, nlist, thresh,
        outp, outs, flags, stats, redo, rowmask, eps_pair, tokmax, tau_glob, wsel, bounds, tscale, rangep, cells8;
    DevBuf g_cells, g_keys, g_keys2, g_vals, g_vals2, g_scratch, g_sort_tmp;   // general-shape path (generic_kernels.hpp)
};

constexpr int kWorkspaceSlots = 4;

struct clb_searcher {
    int device = 0;
    int64_t dim = 0, K = 0, n_docs = 0, n_emb = 0, pid_offset = 0;
    int nbits = 0;
    int mode = 0;
    int wide_select = -1;      // selection by kWideBlocks work-groups per query: -1 by candidate capacity, 0 never, 1 always
    bool approx_ok = false;
    bool bounds_synced = false;   // clb_searcher_set_bound_consts has been called: the error bound is the shard group's, not this shard's
    bool ivf_sorted = false;   // every IVF list holds non-decreasing passage ids (mark_count_kernel<true> needs it)
    bool generic = false;      // dim != 128 or nbits == 8: every query takes the general-shape path
    int64_t max_doclen = 0;
    hipStream_t stream = nullptr;
    // resident index (HBM)
    DevBuf centroids;   // fp32 [K][128]
    DevBuf weights;     // fp32 [2^nbits]
    DevBuf codes0;      // u32 [n_emb], 0-based
    DevBuf residuals;   // u8 [n_emb][16*nbits]
    DevBuf doc_off;     // u32 [n_docs+1]
    DevBuf ivf_off;     // u32 [K+1]
    DevBuf ivf_pid;     // u32 [n_emb] local passage ids grouped by centroid
    DevBuf codeinv;     // u32 [n_emb]: code | quantised inv_norm, the one word pass 1 streams per embedding (two-pass mode)
    int cbits = 0;      // bits of the code field
    float inv_lo = 0.f, inv_step = 0.f;
    DevBuf cent_hi, cent_lo;  // bf16 [K][128] split of the centroids (bf16x3 centroid scoring)
    DevBuf cent_f16;          // fp16 [K][128]: the one operand of the single-product score table (batches of 16+ queries)
    float dc_f16 = 0.f;       // max ||c - fp16(c)|| over the centroids; approx_consts.dc_max carries it only for tables made from cent_f16
    int s1_x1 = -1;           // 16+ queries: score table from ONE fp16 product (to_f16_kernel's comment)?  1 yes, 0 three bf16 products,
                              // -1 by the handle's role: yes on a shard of a group (bounds_synced: the centroid stage is replicated on every
                              // shard while the pass-2 rows its wider bound adds are divided among them), no on a single GPU, where
                              // -0.020 ms on the centroid kernel meets +0.01-0.02 ms on pass 2 (profiles/r05_experiments.md)
    int s1_mode = 1;    // 1: bf16x3 + exact refine, 0: fp32 MFMA
    int gather_lds = 0; // pass 1: score rows through LDS-DMA, four adjacent lanes per row (0: the per-lane VGPR gather); set at load
    double code_adjacency = 0.0;   // fraction of consecutive embeddings that share a 128-B line of the score table
    int cell8 = -1;     // batches of 16+ queries: score rows as 32 bytes of 8-bit cells?  1 yes, 0 / -1 (default) fp16 rows.  On the
                        // shards of a group every shard's bound must cover every shard's table: set it alike on all shards
    ApproxConsts approx_consts{};
    std::vector<uint32_t> ivf_len_sorted;  // descending, for the candidate-capacity bound
    Workspace ws[kWorkspaceSlots];   // per-batch scratch, grown on demand (ensure_workspace); slots 1..: further batches in flight
    Prof prof;
    int64_t last_cand_docs = 0, last_cand_embs = 0, last_resc_docs = 0, last_resc_embs = 0;
    int64_t index_bytes = 0;
};

namespace {

struct Timed {
    clb_searcher* s;
    int id;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(clb_searcher* s_, int id_, hipStream_t st_) : s(s_), id(id_), st(st_) {
        if (s->prof.on) {
            if (s->prof.chain && s->prof.chain_stream == st) {
                a = s->prof.chain;
            } else {
                a = s->prof.get();
                if (a && hipEventRecord(a, st) != hipSuccess) { s->prof.pool.push_back(a); a = nullptr; s->prof.failed = true; }
            }
            b = a ? s->prof.get() : nullptr;
        }
    }
    ~Timed() {
        if (!s->prof.on) return;
        if (a && b && hipEventRecord(b, st) == hipSuccess) {
            s->prof.pending[id].push_back({a, b});
            s->prof.chain = b;
            s->prof.chain_stream = st;
        } else {                       // untimed launch: the next timed kernel records its own start
            s->prof.failed = true;
            s->prof.chain = nullptr;
        }
    }
};

// token tiles of 32 for the cells table: Tpad in {32, 64, 128} so that it divides the 256-thread scan
inline int token_tiles(int64_t T) { return T <= 32 ? 1 : T <= 64 ? 2 : 4; }

// 8-bit score rows for this handle's batches of 16+ queries?
// (round 6: only when asked for -- measured end to end the format loses on all four workloads, profiles/r06_experiments.md)
inline bool cell8_rows(const clb_searcher* s) { return s->cell8 == 1 && s->approx_ok; }

// the top-k kernel sorts up to kMaxTopK 8-byte keys in LDS: beyond 64 KB the attribute has to be raised
void allow_large_topk_lds() {
    allow_dynamic_lds(reinterpret_cast<const void*>(topk_kernel), (int)(sizeof(unsigned long long) * kMaxTopK));
}

constexpr int64_t kSubBatch = 64;      // queries per pass of the single-call search entry points (see clb_search_batch_device_slot)

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

int ensure_workspace(clb_searcher* s, Workspace& w, int64_t B, int64_t T, int64_t nprobe, int64_t k) {
    const int64_t t_tuned = T <= 128 ? T : 0;       // queries of up to 128 tokens take the tuned (or batched general) kernels
    if (B <= w.Bcap && T <= w.Tcap && t_tuned <= w.Ttuned && nprobe <= w.npcap && k <= w.kcap) return CLB_OK;
    CLB_HIP(hipDeviceSynchronize());       // buffers may be in use on any of the caller's streams
    B = std::max(B, w.Bcap); T = std::max(T, w.Tcap);
    nprobe = std::max(nprobe, w.npcap); k = std::max(k, w.kcap);
    // T is the LARGEST query length this slot has seen; the buffers of the tuned kernels (the fp32 / fp16 score tables:
    // B K Tpad entries, 2 GB at B = 32, K = 131 072) are sized for the largest query of UP TO 128 tokens it has seen --
    // a longer query (per-query general path, which touches none of them) neither allocates them for 128 tokens nor
    // leaves a later short query without them
    const int64_t Ttuned = std::max(w.Ttuned, t_tuned);
    const int64_t Tpad = token_tiles(std::max<int64_t>(Ttuned, 1)) * 32;
    // candidates of one query <= sum of the T*nprobe longest IVF lists (and <= n_docs)
    size_t lists = (size_t)std::min<int64_t>(T * nprobe, s->K);
    size_t cap = 0;
    for (size_t i = 0; i < lists; ++i) cap += s->ivf_len_sorted[i];
    cap = std::min<size_t>(cap, (size_t)s->n_docs);
    cap = std::max<size_t>(cap, 1);
    w.cand_cap = (cap + 3) & ~(size_t)3;
    w.W = (int)((s->n_docs + 31) / 32);
    w.nblk_bitmap = (w.W + kScanBlock * kWordsPerThread - 1) / (kScanBlock * kWordsPerThread);
    w.topn_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(256, s->K / 512));
    const int64_t NPs = nprobe <= 2 ? 2 : nprobe <= 8 ? 8 : nprobe <= 32 ? 32 : nprobe;
    const bool general = s->generic;
    CLB_TRY(w.Qdev.ensure(sizeof(float) * B * T * s->dim));
    // the fp32 T x K score matrix is only materialised by the unfused S1/S2 path (nprobe > 2 or T > 32)
    // (the general-shape path scores its centroids into the same matrix whenever its batched kernels apply)
    const bool general_batched = general && s->dim % 4 == 0 && s->dim <= 256;
    if (Ttuned > 0 && ((!general && !(nprobe <= 2 && Ttuned <= 32)) || general_batched))
        CLB_TRY(w.cells.ensure(sizeof(float) * B * s->K * Tpad));
    if (Ttuned > 0 && (!general || general_batched))
        CLB_TRY(w.partial.ensure(sizeof(ValIdx) * B * w.topn_blocks * Tpad * std::min<int64_t>(NPs, 32)));
    CLB_TRY(w.sel.ensure(sizeof(int) * B * std::max<int64_t>(Tpad, T) * NPs));
    const size_t bm_bytes = sizeof(uint32_t) * (size_t)B * w.W;
    const bool bm_new = bm_bytes > w.bitmap.bytes || !w.bitmap.p;
    CLB_TRY(w.bitmap.ensure(bm_bytes));
    if (bm_new) CLB_HIP(hipMemsetAsync(w.bitmap.p, 0, w.bitmap.bytes, s->stream));
    CLB_TRY(w.blocksum.ensure(sizeof(int) * B * w.nblk_bitmap));
    CLB_TRY(w.ncand.ensure(sizeof(int) * B));
    CLB_TRY(w.cand.ensure(sizeof(uint32_t) * B * w.cand_cap));
    {   // slice boundaries of every selected list (mark_count_kernel<true>, shards of more than 16 slices)
        const size_t nsl = ((size_t)w.nblk_bitmap + kMarkSliceBlocks - 1) / kMarkSliceBlocks;
        const size_t nbig = ((size_t)w.nblk_bitmap + kMarkSliceBlocksBig - 1) / kMarkSliceBlocksBig;
        if (nsl > 16) CLB_TRY(w.bounds.ensure(sizeof(uint32_t) * B * T * nprobe * (nbig + 1)));
    }
    CLB_TRY(w.cand_hdr.ensure(sizeof(uint2) * B * w.cand_cap));
    CLB_TRY(w.scores.ensure(sizeof(float) * B * w.cand_cap));
    CLB_TRY(w.list.ensure(sizeof(int) * B * w.cand_cap));
    CLB_TRY(w.nlist.ensure(sizeof(int) * B));
    CLB_TRY(w.thresh.ensure(sizeof(float) * B * 2));
    CLB_TRY(w.outp.ensure(sizeof(int64_t) * B * k));
    CLB_TRY(w.outs.ensure(sizeof(float) * B * k));
    CLB_TRY(w.flags.ensure(sizeof(int) * B));
    if (!w.stats.p) {
        CLB_TRY(w.stats.ensure(sizeof(unsigned long long) * 8));
        CLB_HIP(hipMemset(w.stats.p, 0, sizeof(unsigned long long) * 8));
    }
    if (s->approx_ok && Ttuned > 0) {
        CLB_TRY(w.cells_q.ensure(approx_cells_bytes(B, s->K, Tpad)));
        CLB_TRY(w.rowmask.ensure(sizeof(unsigned long long) * 4 * B * w.cand_cap));
        CLB_TRY(w.eps_pair.ensure(sizeof(float) * B));
        CLB_TRY(w.tokmax.ensure(sizeof(uint16_t) * 32 * B * w.cand_cap));
    }
    w.Bcap = B; w.Tcap = T; w.Ttuned = Ttuned; w.npcap = nprobe; w.kcap = k;
    CLB_HIP(hipStreamSynchronize(s->stream));
    return CLB_OK;
}

template <int NP>
void launch_topn(clb_searcher* s, Workspace& w, hipStream_t st, int B, int Tpad) {
    hipLaunchKernelGGL(topn_partial_kernel<NP>, dim3(w.topn_blocks, B), dim3(256), 0, st,
                       w.cells.as<float>(), w.partial.as<ValIdx>(), (int)s->K, Tpad);
    hipLaunchKernelGGL(topn_final_kernel<NP>, dim3(B), dim3(Tpad), 0, st, w.partial.as<ValIdx>(),
                       w.sel.as<int>(), w.topn_blocks, Tpad);
}

template <int NBITS>
void launch_score_exact(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, const int* list,
                        const int* nlist, int grid_x) {
    hipLaunchKernelGGL(score_exact_kernel<NBITS>, dim3(grid_x, B), dim3(256), 0, st, s->centroids.as<float>(),
                       s->weights.as<float>(), s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(),
                       w.cand_hdr.as<uint2>(), dQ, w.ncand.as<int>(), w.scores.as<float>(), T, w.cand_cap, list,
                       nlist);
}

int select_by_sort(clb_searcher* s, Workspace& w, hipStream_t st, const float* cells, size_t stride_t, size_t stride_c,
                   int T, int nprobe, int NP, int* sel_b);

// Candidate generation S1-S3 for B queries on stream st; leaves cand/ncand on the device.
int mark_and_compact(clb_searcher* s, Workspace& w, hipStream_t st, int B, int T, int Tpad, int NPs, int nprobe);

int run_retrieve(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, int nprobe) {
    const int TT = token_tiles(T), Tpad = TT * 32;
    const int NPs = nprobe <= 2 ? 2 : nprobe <= 8 ? 8 : nprobe <= 32 ? 32 : nprobe;
    const int n_tiles = (int)((s->K + 31) / 32);
    const bool want_half = s->mode == 1 && s->approx_ok && T <= 32;
    w.x1_table = false;
    w.have_range = false;
    w.cell8 = false;
    if (nprobe <= 2 && T <= 32) {
        // fused S1+S2: no fp32 score matrix; fp16 pairs only when pass 1 will gather them
        // batches of 8+ queries share each staged centroid tile between 8 queries (centroid_top_bf16x3_mq_kernel)
        // (from 6 queries: the shared-tile kernel with two idle query slots, 0.047 ms, beats six or seven per-query passes
        // over the table, 0.06-0.07 ms)
        const bool mq = s->s1_mode == 1 && s->cent_hi.p && B >= 6;
        const int groups = (B + kMqQueries - 1) / kMqQueries;
        int gx = mq ? std::max(1, std::min(n_tiles, std::min(256, std::max(512 / groups, 16))))
                          : std::max(1, std::min(n_tiles / 2 + 1, std::min(256, std::max(1024 / std::max(1, B), 16))));
        const int gx_dbg = CLB_KNOB("CLB_DEBUG_S1_GX", 0);
        if (gx_dbg > 0 && !mq) gx = std::min(gx_dbg, n_tiles / 2 + 1);
        // 16+ queries and a score table to write: two teams of four waves per work-group, 16 queries per staged tile
        // (centroid_top_bf16x3_teams_kernel); one 8-wave work-group per CU
        const bool teams = mq && want_half && B >= kTeamQueries && CLB_KNOB("CLB_DEBUG_S1_TEAMS", 1);
        const int team_groups = (B + kTeamQueries - 1) / kTeamQueries;
        if (teams) gx = std::max(1, std::min(n_tiles, std::min(256, std::max(256 / team_groups, 16))));
        const bool x1 = teams && (s->s1_x1 == 1 || (s->s1_x1 < 0 && s->bounds_synced)) && s->cent_f16.p && s->dc_f16 > 0.f;
        w.x1_table = x1;
        w.cell8 = teams && cell8_rows(s);
        w.have_range = w.cell8;
        if (w.cell8) {
            CLB_TRY(w.tscale.ensure(sizeof(float4) * 32 * B));
            CLB_TRY(w.rangep.ensure(sizeof(float2) * 32 * B * gx));
            CLB_TRY(w.cells8.ensure((size_t)B * s->K * 32));
        }
        const int nslots = mq ? gx * 2 : gx * 4;
        CLB_TRY(w.partial.ensure(sizeof(ValIdx) * (size_t)B * nslots * 32 * kTopPartial));
        const size_t lds_f32 = 2 * 32 * kCentTileStride * sizeof(float);
        if (s->s1_mode == 1 && s->cent_hi.p) {
            if (w.redo.bytes < sizeof(int) * B) {          // cumulative fallback counter (statistics only)
                CLB_TRY(w.redo.ensure(sizeof(int) * B));
                CLB_HIP(hipMemsetAsync(w.redo.p, 0, w.redo.bytes, st));
            }
            {
                Timed t(s, KID_CENTROID_SCORES, st);
                const size_t lds_b16 = 2 * 2 * 32 * kRowBytes16;
                if (teams) {
                    // 66 KB of dynamic LDS: above the 64-KB default limit of a launch
                    auto kern = x1 ? (w.cell8 ? centroid_top_bf16x3_teams_kernel<true, true> : centroid_top_bf16x3_teams_kernel<true, false>)
                                   : (w.cell8 ? centroid_top_bf16x3_teams_kernel<false, true> : centroid_top_bf16x3_teams_kernel<false, false>);
                    allow_dynamic_lds(reinterpret_cast<const void*>(kern), 2 * 2 * 32 * kRowBytes16 + 8 * 4096);
                    hipLaunchKernelGGL(kern, dim3(gx, team_groups), dim3(512), lds_b16 + 8 * 4096, st,
                                       x1 ? s->cent_f16.as<uint16_t>() : s->cent_hi.as<uint16_t>(),
                                       x1 ? (const uint16_t*)nullptr : (const uint16_t*)s->cent_lo.as<uint16_t>(), dQ,
                                       w.partial.as<ValIdx>(), w.cells_q.as<uint32_t>(), (int)s->K, T, B, n_tiles,
                                       w.rangep.as<float2>());
                    if (w.cell8) {     // 8-bit rows: every token's measured score range, then the table rewritten by it
                        hipLaunchKernelGGL(token_range_kernel, dim3(B), dim3(1024), 0, st, (const float2*)w.rangep.as<float2>(), gx, T,
                                           w.tscale.as<float4>());
                        hipLaunchKernelGGL(requantise_cells_kernel, dim3(std::max(1, 2048 / B), B), dim3(256), 0, st,
                                           (const uint32_t*)w.cells_q.as<uint32_t>(), (const float4*)w.tscale.as<float4>(),
                                           w.cells8.as<uint32_t>(), (int)s->K);
                    }
                } else if (mq && want_half)
                    hipLaunchKernelGGL(centroid_top_bf16x3_mq_kernel<true>, dim3(gx, groups), dim3(256), lds_b16 + 4 * 2048, st,
                                       s->cent_hi.as<uint16_t>(), s->cent_lo.as<uint16_t>(), dQ,
                                       w.partial.as<ValIdx>(), w.cells_q.as<uint32_t>(), (int)s->K, T, B, n_tiles);
                else if (mq)
                    hipLaunchKernelGGL(centroid_top_bf16x3_mq_kernel<false>, dim3(gx, groups), dim3(256), lds_b16, st,
                                       s->cent_hi.as<uint16_t>(), s->cent_lo.as<uint16_t>(), dQ,
                                       w.partial.as<ValIdx>(), (uint32_t*)nullptr, (int)s->K, T, B, n_tiles);
                else if (want_half)
                    hipLaunchKernelGGL(centroid_top_bf16x3_kernel<true>, dim3(gx, B), dim3(128), lds_b16 + 2 * 2048, st,
                                       s->cent_hi.as<uint16_t>(), s->cent_lo.as<uint16_t>(), dQ,
                                       w.partial.as<ValIdx>(), w.cells_q.as<uint32_t>(), (int)s->K, T, n_tiles);
                else
                    hipLaunchKernelGGL(centroid_top_bf16x3_kernel<false>, dim3(gx, B), dim3(128), lds_b16, st,
                                       s->cent_hi.as<uint16_t>(), s->cent_lo.as<uint16_t>(), dQ,
                                       w.partial.as<ValIdx>(), (uint32_t*)nullptr, (int)s->K, T, n_tiles);
            }
            {
                Timed t(s, KID_TOPN, st);
                hipLaunchKernelGGL(top_refine_kernel, dim3(32, B), dim3(64), 0, st, w.partial.as<ValIdx>(),
                                   s->centroids.as<float>(), dQ, T, (int)s->K, nslots, s->approx_consts.cn_max,
                                   w.sel.as<int>(), w.redo.as<int>(), x1 ? s->dc_f16 : 0.f);
            }
        } else {
            {
                Timed t(s, KID_CENTROID_SCORES, st);
                if (want_half)
                    hipLaunchKernelGGL(centroid_top2_kernel<true>, dim3(gx, B), dim3(128), lds_f32, st,
                                       s->centroids.as<float>(), dQ, w.partial.as<ValIdx>(),
                                       w.cells_q.as<uint32_t>(), (int)s->K, T, n_tiles, (const int*)nullptr);
                else
                    hipLaunchKernelGGL(centroid_top2_kernel<false>, dim3(gx, B), dim3(128), lds_f32, st,
                                       s->centroids.as<float>(), dQ, w.partial.as<ValIdx>(), (uint32_t*)nullptr,
                                       (int)s->K, T, n_tiles, (const int*)nullptr);
            }
            {
                Timed t(s, KID_TOPN, st);
                hipLaunchKernelGGL(top2_merge_kernel, dim3(32, B), dim3(64), 0, st, w.partial.as<ValIdx>(),
                                   w.sel.as<int>(), nslots, (const int*)nullptr);
            }
        }
    } else {
        {
            Timed t(s, KID_CENTROID_SCORES, st);
            const int gx = std::max(1, std::min(n_tiles / 2 + 1, 2048 / std::max(1, B * TT)));
            hipLaunchKernelGGL(centroid_scores_kernel, dim3(gx, B * TT), dim3(128),
                               2 * 32 * kCentTileStride * sizeof(float), st, s->centroids.as<float>(), dQ,
                               w.cells.as<float>(), (int)s->K, T, TT, n_tiles);
        }
        {
            Timed t(s, KID_TOPN, st);
            if (NPs == 2) launch_topn<2>(s, w, st, B, Tpad);
            else if (NPs == 8) launch_topn<8>(s, w, st, B, Tpad);
            else if (NPs == 32) launch_topn<32>(s, w, st, B, Tpad);
            else   // nprobe > 32: stable sort of every token's K scores (one query at a time)
                for (int b = 0; b < B; ++b)
                    CLB_TRY(select_by_sort(s, w, st, w.cells.as<float>() + (size_t)b * s->K * Tpad, 1, (size_t)Tpad, T,
                                           nprobe, NPs, w.sel.as<int>() + (size_t)b * Tpad * NPs));
        }
        s->prof.chain = nullptr;   // untimed conversion below
        if (want_half)
            hipLaunchKernelGGL(cells_to_half_kernel, dim3(std::max(1, 1024 / B), B), dim3(256), 0, st,
                               w.cells.as<float>(), w.cells_q.as<uint32_t>(), (int)s->K);
    }
    return mark_and_compact(s, w, st, B, T, Tpad, NPs, nprobe);
}

// S3: the union of the selected IVF lists as an ascending pid list + passage headers, for B queries (w.sel -> w.cand,
// w.cand_hdr, w.ncand); shared by the tuned and the general-shape path
int mark_and_compact(clb_searcher* s, Workspace& w, hipStream_t st, int B, int T, int Tpad, int NPs, int nprobe) {
    static_assert(kScanBlock * kWordsPerThread == 1024, "mark_count_kernel writes 1024-word count blocks");
    const int nslices = (w.nblk_bitmap + kMarkSliceBlocks - 1) / kMarkSliceBlocks;
    const int nslices_big = (w.nblk_bitmap + kMarkSliceBlocksBig - 1) / kMarkSliceBlocksBig;
    // every slice re-reads the query's lists: beyond 16 slices (2 M passages per shard) that costs more than the atomics
    // it saves -- unless the lists are sorted by passage id: then slice_bounds_kernel cuts every list at the slice
    // boundaries first (binary searches) and larger slices keep the number of work-groups down.  For a few queries the
    // 64 one-list work-groups of the atomic path finish sooner (one query: 8 us against 27)
    const bool sliced = (nslices <= 16 || s->ivf_sorted) && B >= 8 && !CLB_KNOB("CLB_DEBUG_ATOMIC_MARK", 0);
    {
        Timed t(s, KID_MARK, st);
        if (sliced && nslices <= 16)     // mark + per-block counts, the bitmap slice of a work-group in LDS (no global atomics)
            hipLaunchKernelGGL((mark_count_kernel<false, kMarkSliceBlocks>), dim3(nslices, B), dim3(1024), 0, st, w.sel.as<int>(),
                               s->ivf_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), w.bitmap.as<uint32_t>(),
                               w.blocksum.as<int>(), T, Tpad, NPs, nprobe, w.W, w.nblk_bitmap);
        else if (sliced) {
            const int nb = T * nprobe * (nslices_big + 1);
            hipLaunchKernelGGL(slice_bounds_kernel, dim3((nb + 255) / 256, B), dim3(256), 0, st, w.sel.as<int>(),
                               s->ivf_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), T, Tpad, NPs, nprobe, nslices_big,
                               (uint32_t)(kMarkSliceBlocksBig * 1024 * 32), w.bounds.as<uint32_t>());
            hipLaunchKernelGGL((mark_count_kernel<true, kMarkSliceBlocksBig>), dim3(nslices_big, B), dim3(1024), 0, st,
                               w.sel.as<int>(), s->ivf_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), w.bitmap.as<uint32_t>(),
                               w.blocksum.as<int>(), T, Tpad, NPs, nprobe, w.W, w.nblk_bitmap,
                               (const uint32_t*)w.bounds.as<uint32_t>());
        } else
            hipLaunchKernelGGL(mark_candidates_kernel, dim3(T * nprobe, B), dim3(256), 0, st, w.sel.as<int>(),
                               s->ivf_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), w.bitmap.as<uint32_t>(), T,
                               Tpad, NPs, nprobe, w.W);
    }
    {
        Timed t(s, KID_COMPACT, st);
        if (!sliced)
            hipLaunchKernelGGL(bitmap_count_kernel, dim3(w.nblk_bitmap, B), dim3(kScanBlock), 0, st,
                               w.bitmap.as<uint32_t>(), w.blocksum.as<int>(), w.W);
        hipLaunchKernelGGL(bitmap_emit_kernel, dim3(w.nblk_bitmap, B), dim3(kScanBlock), 0, st,
                           w.bitmap.as<uint32_t>(), w.blocksum.as<int>(), w.cand.as<uint32_t>(),
                           s->doc_off.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.W, w.cand_cap, w.ncand.as<int>());
    }
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

int check_search_args(clb_searcher* s, int64_t T, int64_t B, int64_t nprobe, int64_t k) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (T < 1) return fail(CLB_EARGUMENT, "query length must be >= 1");
    if (B < 1) return fail(CLB_EARGUMENT, "batch size must be >= 1");
    if (nprobe < 1 || nprobe > s->K) return fail(CLB_EBOUNDS, "nprobe=%lld outside 1..K=%lld (partialsortperm)", (long long)nprobe, (long long)s->K);
    if (k < 1) return fail(CLB_EBOUNDS, "k must be >= 1");
    if (T * nprobe > (int64_t)0x7fffffff / 4 || T * s->K > (int64_t)0x7fffffff) return fail(CLB_EUNSUPPORTED, "T * K too large");
    return CLB_OK;
}

// ---- general-shape pieces (generic_kernels.hpp) --------------------------------------------------------------
// top-nprobe of every token of ONE query by a stable sort of the T x K scores; sel_b: [T][NP] (0-based centroid ids)
int select_by_sort(clb_searcher* s, Workspace& w, hipStream_t st, const float* cells, size_t stride_t, size_t stride_c,
                   int T, int nprobe, int NP, int* sel_b) {
    const size_t n = (size_t)T * s->K;
    CLB_TRY(w.g_keys.ensure(sizeof(uint64_t) * n));
    CLB_TRY(w.g_keys2.ensure(sizeof(uint64_t) * n));
    CLB_TRY(w.g_vals.ensure(sizeof(uint32_t) * n));
    CLB_TRY(w.g_vals2.ensure(sizeof(uint32_t) * n));
    hipLaunchKernelGGL(generic_sel_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cells, stride_t,
                       stride_c, (int)s->K, T, w.g_keys.as<unsigned long long>(), w.g_vals.as<uint32_t>());
    CLB_TRY(sort_pairs_u64(w.g_keys.as<uint64_t>(), w.g_keys2.as<uint64_t>(), w.g_vals.as<uint32_t>(),
                           w.g_vals2.as<uint32_t>(), n, st, &w.g_sort_tmp));
    hipLaunchKernelGGL(generic_sel_extract_kernel, dim3((unsigned)((T * nprobe + 255) / 256)), dim3(256), 0, st,
                       w.g_vals2.as<uint32_t>(), (int)s->K, T, nprobe, NP, sel_b);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

// top-k of ONE query by a full stable sort (k above the single-work-group sort of topk_kernel).  The count stays on the
// device: the sort runs over the slot's candidate capacity with the positions past the count keyed to the end, the
// emit kernel writes the short-result flag and the candidate count; rocPRIM's temporary storage lives in the workspace
// slot (g_sort_tmp, sized on first use), so the sort is only enqueued -- no allocation and no host synchronisation per
// query.  The price of keeping the count on the device: the sort covers cand_cap keys, not the query's own count.
int topk_by_sort(clb_searcher* s, Workspace& w, hipStream_t st, int b, const int* list, const int* nlist, int k,
                 int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand) {
    const int cap = (int)w.cand_cap;
    const float* sc = w.scores.as<float>() + (size_t)b * w.cand_cap;
    const int* lst = list ? list + (size_t)b * w.cand_cap : nullptr;
    const int* n_ptr = list ? nlist + b : w.ncand.as<int>() + b;
    CLB_TRY(w.g_keys.ensure(sizeof(uint64_t) * std::max(cap, 1)));
    CLB_TRY(w.g_keys2.ensure(sizeof(uint64_t) * std::max(cap, 1)));
    hipLaunchKernelGGL(generic_topk_keys_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, sc, lst, n_ptr, cap,
                       w.g_keys.as<unsigned long long>());
    CLB_TRY(sort_keys_u64(w.g_keys.as<uint64_t>(), w.g_keys2.as<uint64_t>(), (size_t)cap, st, &w.g_sort_tmp));
    hipLaunchKernelGGL(generic_topk_emit_kernel, dim3((std::max(k, 1) + 255) / 256), dim3(256), 0, st,
                       w.g_keys2.as<unsigned long long>(), sc, w.cand.as<uint32_t>() + (size_t)b * w.cand_cap, lst, n_ptr,
                       w.ncand.as<int>() + b, k, s->pid_offset, d_out_pids + (size_t)b * k, d_out_scores + (size_t)b * k,
                       w.flags.as<int>() + b, d_n_cand ? d_n_cand + b : nullptr);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

// S1-S3 of ONE query on the general-shape path: leaves cand / cand_hdr / ncand of slot b
int run_retrieve_general(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int b, int T, int nprobe) {
    const float* q = dQ + (size_t)b * T * s->dim;
    CLB_TRY(w.g_cells.ensure(sizeof(float) * (size_t)T * s->K));
    hipLaunchKernelGGL(generic_cells_kernel, dim3((unsigned)((s->K + 127) / 128), T), dim3(128), sizeof(float) * s->dim, st,
                       s->centroids.as<float>(), q, (int)s->dim, (int)s->K, w.g_cells.as<float>());
    int* sel_b = w.sel.as<int>();                 // one query at a time: slot 0 of the selection buffer
    CLB_TRY(select_by_sort(s, w, st, w.g_cells.as<float>(), (size_t)s->K, 1, T, nprobe, nprobe, sel_b));
    uint32_t* bm = w.bitmap.as<uint32_t>() + (size_t)b * w.W;
    int* bs = w.blocksum.as<int>() + (size_t)b * w.nblk_bitmap;
    hipLaunchKernelGGL(mark_candidates_kernel, dim3(T * nprobe, 1), dim3(256), 0, st, sel_b, s->ivf_off.as<uint32_t>(),
                       s->ivf_pid.as<uint32_t>(), bm, T, T, nprobe, nprobe, w.W);
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(w.nblk_bitmap, 1), dim3(kScanBlock), 0, st, bm, bs, w.W);
    hipLaunchKernelGGL(bitmap_emit_kernel, dim3(w.nblk_bitmap, 1), dim3(kScanBlock), 0, st, bm, bs,
                       w.cand.as<uint32_t>() + (size_t)b * w.cand_cap, s->doc_off.as<uint32_t>(),
                       w.cand_hdr.as<uint2>() + (size_t)b * w.cand_cap, w.W, w.cand_cap, w.ncand.as<int>() + b);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

// S1..S3 of B queries on the general-shape path when the shape allows the batched kernels: fp32-MFMA centroid scores in
// the tuned path's [b][centroid][Tpad] layout, then the tuned path's own selection, marking and compaction
bool general_batched_ok(const clb_searcher* s, int T) { return s->dim % 4 == 0 && s->dim <= 256 && T <= 128; }

int run_retrieve_general_batched(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, int nprobe) {
    const int TT = token_tiles(T), Tpad = TT * 32;
    const int NPs = nprobe <= 2 ? 2 : nprobe <= 8 ? 8 : nprobe <= 32 ? 32 : nprobe;
    const dim3 grid((unsigned)((s->K + 63) / 64), (unsigned)B);
    if (s->dim <= 128)
        hipLaunchKernelGGL(generic_cells_mfma_kernel<32>, grid, dim3(256), 0, st, s->centroids.as<float>(), dQ, (int)s->dim, (int)s->K,
                           T, Tpad, w.cells.as<float>());
    else
        hipLaunchKernelGGL(generic_cells_mfma_kernel<64>, grid, dim3(256), 0, st, s->centroids.as<float>(), dQ, (int)s->dim, (int)s->K,
                           T, Tpad, w.cells.as<float>());
    if (NPs == 2) launch_topn<2>(s, w, st, B, Tpad);
    else if (NPs == 8) launch_topn<8>(s, w, st, B, Tpad);
    else if (NPs == 32) launch_topn<32>(s, w, st, B, Tpad);
    else
        for (int b = 0; b < B; ++b)
            CLB_TRY(select_by_sort(s, w, st, w.cells.as<float>() + (size_t)b * s->K * Tpad, 1, (size_t)Tpad, T, nprobe, NPs,
                                   w.sel.as<int>() + (size_t)b * Tpad * NPs));
    return mark_and_compact(s, w, st, B, T, Tpad, NPs, nprobe);
}

// the whole search of B queries on the general-shape path (exact scoring only)
int run_search_general(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, int nprobe, int k,
                       int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand) {
    s->prof.chain = nullptr;
    if (general_batched_ok(s, T) && !CLB_KNOB("CLB_DEBUG_GENERIC_SCALAR", 0)) {
        // batched: every stage is one launch for the B queries (the loop below: ~12 launches per query, the scoring
        // kernel a chain of dependent loads)
        CLB_TRY(run_retrieve_general_batched(s, w, st, dQ, B, T, nprobe));
        const dim3 grid((unsigned)std::max(64, 2048 / std::max(1, B)), (unsigned)B);
        const size_t lds = sizeof(float) * (((size_t)1 << s->nbits) + (size_t)T * (s->dim + 1));   // <= 133 KB (T 128, dim 256)
        if (lds > 64 * 1024) {
            allow_dynamic_lds(reinterpret_cast<const void*>(generic_score_mfma_fast_kernel<32>), (int)lds);
            allow_dynamic_lds(reinterpret_cast<const void*>(generic_score_mfma_fast_kernel<64>), (int)lds);
        }
        if (s->dim <= 128)
            hipLaunchKernelGGL(generic_score_mfma_fast_kernel<32>, grid, dim3(256), lds, st, s->centroids.as<float>(),
                               s->weights.as<float>(), s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(), w.cand_hdr.as<uint2>(),
                               w.ncand.as<int>(), dQ, (int)s->dim, s->nbits, T, w.cand_cap, w.scores.as<float>());
        else
            hipLaunchKernelGGL(generic_score_mfma_fast_kernel<64>, grid, dim3(256), lds, st, s->centroids.as<float>(),
                               s->weights.as<float>(), s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(), w.cand_hdr.as<uint2>(),
                               w.ncand.as<int>(), dQ, (int)s->dim, s->nbits, T, w.cand_cap, w.scores.as<float>());
        CLB_HIP(hipGetLastError());
        if (k <= kMaxTopK) {
            const int kpow2 = next_pow2(k);
            allow_large_topk_lds();
            hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), sizeof(unsigned long long) * kpow2, st, w.scores.as<float>(),
                               w.cand.as<uint32_t>(), w.ncand.as<int>(), (const int*)nullptr, (const int*)nullptr, k, kpow2,
                               w.cand_cap, s->pid_offset, d_out_pids, d_out_scores, w.flags.as<int>(), d_n_cand);
            CLB_HIP(hipGetLastError());
        } else {
            for (int b = 0; b < B; ++b) CLB_TRY(topk_by_sort(s, w, st, b, nullptr, nullptr, k, d_out_pids, d_out_scores, d_n_cand));
        }
        return CLB_OK;
    }
    const int grid = 1024;
    const size_t max_len = (size_t)std::max<int64_t>(s->max_doclen, 1);
    CLB_TRY(w.g_scratch.ensure(sizeof(float) * grid * max_len * s->dim));
    for (int b = 0; b < B; ++b) {
        CLB_TRY(run_retrieve_general(s, w, st, dQ, b, T, nprobe));
        if (T <= 16 * kGenericMaxTokenGroups && s->dim % 4 == 0 && !CLB_KNOB("CLB_DEBUG_GENERIC_SCALAR", 0))
            // fp32 MFMA, one wave per passage (the canonical arithmetic of the scalar kernel, bit for bit)
            hipLaunchKernelGGL(generic_score_mfma_kernel, dim3(grid), dim3(256), sizeof(float) * ((size_t)1 << s->nbits), st,
                               s->centroids.as<float>(), s->weights.as<float>(), s->codes0.as<uint32_t>(),
                               s->residuals.as<uint8_t>(), w.cand_hdr.as<uint2>() + (size_t)b * w.cand_cap,
                               w.ncand.as<int>() + b, dQ + (size_t)b * T * s->dim, (int)s->dim, s->nbits, T,
                               w.scores.as<float>() + (size_t)b * w.cand_cap);
        else
        hipLaunchKernelGGL(generic_score_kernel, dim3(grid), dim3(256), sizeof(float) * T, st, s->centroids.as<float>(),
                           s->weights.as<float>(), s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(),
                           w.cand_hdr.as<uint2>() + (size_t)b * w.cand_cap, w.ncand.as<int>() + b,
                           dQ + (size_t)b * T * s->dim, (int)s->dim, s->nbits, T, w.g_scratch.as<float>(), max_len,
                           w.scores.as<float>() + (size_t)b * w.cand_cap);
        CLB_HIP(hipGetLastError());
        if (k <= kMaxTopK) {
            const int kpow2 = next_pow2(k);
            allow_large_topk_lds();
            hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), sizeof(unsigned long long) * kpow2, st,
                               w.scores.as<float>() + (size_t)b * w.cand_cap, w.cand.as<uint32_t>() + (size_t)b * w.cand_cap,
                               w.ncand.as<int>() + b, (const int*)nullptr, (const int*)nullptr, k, kpow2, w.cand_cap,
                               s->pid_offset, d_out_pids + (size_t)b * k, d_out_scores + (size_t)b * k,
                               w.flags.as<int>() + b, d_n_cand ? d_n_cand + b : nullptr);
            CLB_HIP(hipGetLastError());
        } else {
            CLB_TRY(topk_by_sort(s, w, st, b, nullptr, nullptr, k, d_out_pids, d_out_scores, d_n_cand));
        }
    }
    return CLB_OK;
}

// The whole search for B device-resident queries, enqueued on st.
// phase 0: the whole search.  Sharded search in two calls (clb_search_shard_phase1/2, two-pass mode only):
// phase 1 = candidate generation, pass 1, local selection, and the shard's k largest approximate scores per query
// to `d_local_top`; phase 2 = global tau from the gathered scores `d_all_top` ([n_shards][B][k]), selection at that
// tau, pass 2, top-k.  Phase 2 continues on the workspace phase 1 left behind.
// tau and the list {approx >= tau - 2 eps} of every query of the batch (two-pass mode).  One work-group per query keeps
// up to 32 768 candidates in registers; shards whose queries can have several times that (candidate capacity >= 131 072:
// roughly 3 M passages and up) take the wide selection -- kWideBlocks work-groups per query, one launch per radix pass.
constexpr size_t kWideSelectCap = 131072;

// Pass 1 over every candidate of the batch: the gather form by the index's code statistics, the row format by the batch's table
void launch_pass1(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, dim3 grid, int B, int T) {
#if CLB_APPROX_WAVES <= 12
    auto kern = w.cell8 ? (s->gather_lds ? score_approx32_kernel<false, 0, 1, 0, true> : score_approx32_kernel<false, 0, 0, 0, true>)
                        : (s->gather_lds ? score_approx32_kernel<false, 0, 1, 0, false> : score_approx32_kernel<false, 0, 0, 0, false>);
#else
    auto kern = w.cell8 ? score_approx32_kernel<false, 0, 0, 0, true> : score_approx32_kernel<false, 0, 0, 0, false>;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(kApproxThreads), 0, st, s->weights.as<float>(),
                       s->codeinv.as<uint32_t>(), s->residuals.as<uint8_t>(), s->cbits, s->inv_lo, s->inv_step, dQ,
                       w.cell8 ? w.cells8.as<uint32_t>() : w.cells_q.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.ncand.as<int>(),
                       w.scores.as<float>(), (int)s->K, T, B, w.cand_cap, w.tokmax.as<uint16_t>(), (const int*)nullptr,
                       (const int*)nullptr, (const float*)nullptr, (unsigned long long*)nullptr,
                       (const float4*)w.tscale.as<float4>());
}
// The centroid side of the single-product score table in the error bound: when this batch's table was made that way -- and,
// on a shard of a group, whenever a shard MAY make its tables that way (the threshold tau comes from every shard's
// approximate scores, so one bound has to cover them all; set clb_searcher_set_centroid_products alike on all shards).
inline float bound_dc(const clb_searcher* s, const Workspace& w) {
    return (w.x1_table || (s->bounds_synced && s->s1_x1 != 0 && s->cent_f16.p)) ? s->dc_f16 : 0.f;
}
int launch_select(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, int k, const float* tau_in,
                  bool coarse_tau = false) {
    const bool wide = s->wide_select == 1 || (s->wide_select < 0 && w.cand_cap >= kWideSelectCap);
    if (!wide) {
        ApproxConsts ac = s->approx_consts;
        ac.dc_max = bound_dc(s, w);
        // tuning builds: CLB_DEBUG_EPS_T_ADD_1E6 widens the per-(token, embedding) bound by about that many millionths
        // through the inv_norm quantisation term (1.01 * inv_qerr * qn * (cn + rn)) -- what a coarser score-table format
        // would cost pass 2 (lists and row masks grow), measured on the real pipeline with correct results
        if (const int add = CLB_KNOB("CLB_DEBUG_EPS_T_ADD_1E6", 0)) ac.inv_qerr += add * 1e-6f / (1.01f * (ac.cn_max + ac.rn_max));
        hipLaunchKernelGGL(select_margin_kernel, dim3(B), dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), dQ, T, k,
                           w.cand_cap, ac, w.list.as<int>(), w.nlist.as<int>(), w.thresh.as<float>(),
                           w.eps_pair.as<float>(), tau_in, coarse_tau ? 1 : 0,
                           w.have_range ? (const float4*)w.tscale.as<float4>() : (const float4*)nullptr, w.cell8 ? 1 : 0,
                           CLB_KNOB("CLB_DEBUG_SELECT_STOP", 0));
        return CLB_OK;
    }
    CLB_TRY(w.wsel.ensure(sizeof(WideSel) * B));
    CLB_HIP(hipMemsetAsync(w.wsel.p, 0, sizeof(WideSel) * B, st));
    const dim3 grid(kWideBlocks, B);
    ApproxConsts acw = s->approx_consts;
    acw.dc_max = bound_dc(s, w);
    hipLaunchKernelGGL(wide_minmax_kernel, grid, dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), dQ, T, w.cand_cap,
                       acw, w.wsel.as<WideSel>(), w.eps_pair.as<float>(),
                       w.have_range ? (const float4*)w.tscale.as<float4>() : (const float4*)nullptr, w.cell8 ? 1 : 0);
    if (!tau_in)
        for (int pass = 0; pass < 4; ++pass)
            hipLaunchKernelGGL(wide_hist_kernel, grid, dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), k, w.cand_cap,
                               w.wsel.as<WideSel>(), pass);
    hipLaunchKernelGGL(wide_count_kernel, grid, dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), k, w.cand_cap,
                       w.wsel.as<WideSel>(), tau_in);
    hipLaunchKernelGGL(wide_emit_kernel, grid, dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), k, w.cand_cap,
                       (const WideSel*)w.wsel.as<WideSel>(), tau_in, w.list.as<int>(), w.nlist.as<int>(), w.thresh.as<float>());
    return CLB_OK;
}

int run_search(clb_searcher* s, Workspace& w, hipStream_t st, const float* dQ, int B, int T, int nprobe, int k,
               int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand = nullptr, int phase = 0,
               float* d_local_top = nullptr, const float* d_all_top = nullptr, int n_shards = 0) {
    if (s->generic || T > 128) {
        if (phase != 0) return fail(CLB_EUNSUPPORTED, "the two-phase sharded search needs the two-pass mode");
        return run_search_general(s, w, st, dQ, B, T, nprobe, k, d_out_pids, d_out_scores, d_n_cand);
    }
    const int kpow2 = next_pow2(std::min(k, (int)kMaxTopK));
    const int* list = nullptr;
    const int* nlist = nullptr;
    s->prof.chain = nullptr;           // the first timed kernel of a call records its own start
    const bool two_pass = s->mode == 1 && s->approx_ok && T <= 32;
    if (phase != 0 && !two_pass) return fail(CLB_EUNSUPPORTED, "the two-phase sharded search needs the two-pass mode");
    if (phase == 2) {
        CLB_TRY(w.tau_glob.ensure(sizeof(float) * B));
        hipLaunchKernelGGL(global_tau_kernel, dim3(B), dim3(1024), 0, st, d_all_top, n_shards, B, k,
                           w.tau_glob.as<float>());
        Timed t(s, KID_SELECT, st);
        CLB_TRY(launch_select(s, w, st, dQ, B, T, k, (const float*)w.tau_glob.as<float>()));
        list = w.list.as<int>();
        nlist = w.nlist.as<int>();
    }
    if (phase != 2) {
    CLB_TRY(run_retrieve(s, w, st, dQ, B, T, nprobe));
    if (s->prof.counters) {
        // one set of counters per CALL: the sub-batches of a large batch add onto those of the sub-batches before them
        if (!w.stats_keep) CLB_HIP(hipMemsetAsync(w.stats.p, 0, sizeof(unsigned long long) * 8, st));
        s->prof.chain = nullptr;
    }
    if (two_pass) {
        {
            Timed t(s, KID_SCORE_APPROX, st);
            const int wgpg = CLB_KNOB("CLB_DEBUG_APPROX_WGPG", 32);   // x 8 XCD groups: one 12-wave work-group per CU
            // Grid: XCD-affine 1-D launch (all work-groups of an XCD share one query's score table in L2) for
            // large candidate sets; for small ones (a shard of a multi-GPU run: < ~6 k candidate passages per
            // query, estimated from the mean IVF list) a (G, B) launch whose few waves per query each get a long
            // run of passages -- the pipeline fill otherwise dominates.
            const int approx_2d = CLB_KNOB("CLB_DEBUG_APPROX_2D", -1);
            const double est_cand = 0.5 * T * nprobe * (double)s->n_emb / (double)std::max<int64_t>(1, s->K);
            // work-groups per query of the (G, B) launch (0 = the 1-D launch): enough to fill the chip for small batches,
            // eight for large ones (measured at 4 and 8 shards, B = 8 ... 256: one or two per query cost 12 % of the pass
            // at B = 256, sixteen and more cost as much at B = 32)
            int gxq = est_cand < 6000.0 ? std::max(8, 256 / B) : 0;
            if (approx_2d >= 0) gxq = approx_2d > 0 ? std::max(1, approx_2d / B) : 0;
            const dim3 approx_grid = gxq > 0 && B > 1 ? dim3(gxq, B) : dim3(8 * wgpg);
#define CLB_LAUNCH_APPROX(ABL)                                                                                        \
    hipLaunchKernelGGL((score_approx32_kernel<false, ABL>), approx_grid, dim3(kApproxThreads), 0, st, s->weights.as<float>(), \
                       s->codeinv.as<uint32_t>(), s->residuals.as<uint8_t>(), s->cbits, s->inv_lo, s->inv_step, dQ,  \
                       w.cells_q.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.ncand.as<int>(), w.scores.as<float>(), \
                       (int)s->K, T, B, w.cand_cap, w.tokmax.as<uint16_t>(), (const int*)nullptr,                    \
                       (const int*)nullptr, (const float*)nullptr, (unsigned long long*)nullptr)
#ifdef CLB_ABLATIONS
            if (CLB_KNOB("CLB_DEBUG_APPROX_PIPE", 0)) {     // round-3 experiment: epilogue of step i-1 under step i's MFMAs (slower: profiles/r03_experiments.md)
#if CLB_APPROX_WAVES <= 12
                auto kern = s->gather_lds ? score_approx32_kernel<false, 0, 1, 1> : score_approx32_kernel<false, 0, 0, 1>;
#else
                auto kern = score_approx32_kernel<false, 0, 0, 1>;
#endif
                hipLaunchKernelGGL(kern, approx_grid, dim3(kApproxThreads), 0, st, s->weights.as<float>(),
                                   s->codeinv.as<uint32_t>(), s->residuals.as<uint8_t>(), s->cbits, s->inv_lo, s->inv_step, dQ,
                                   w.cells_q.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.ncand.as<int>(), w.scores.as<float>(),
                                   (int)s->K, T, B, w.cand_cap, w.tokmax.as<uint16_t>(), (const int*)nullptr,
                                   (const int*)nullptr, (const float*)nullptr, (unsigned long long*)nullptr, (const float4*)nullptr);
            } else
            switch (CLB_KNOB("CLB_DEBUG_APPROX_VARIANT", 0)) {
                case 1: CLB_LAUNCH_APPROX(1); break;
                case 2: CLB_LAUNCH_APPROX(2); break;
                case 3: CLB_LAUNCH_APPROX(3); break;
                case 4: CLB_LAUNCH_APPROX(4); break;
                case 5: CLB_LAUNCH_APPROX(5); break;
                case 6: CLB_LAUNCH_APPROX(6); break;
                case 7: CLB_LAUNCH_APPROX(7); break;
                case 8: CLB_LAUNCH_APPROX(8); break;
                case 11: CLB_LAUNCH_APPROX(11); break;
                case 10:     // the fused row mask writes the slot-indexed row-mask buffer (4 words per candidate slot)
                    hipLaunchKernelGGL((score_approx32_kernel<false, 10>), approx_grid, dim3(kApproxThreads), 0, st, s->weights.as<float>(),
                                       s->codeinv.as<uint32_t>(), s->residuals.as<uint8_t>(), s->cbits, s->inv_lo, s->inv_step, dQ,
                                       w.cells_q.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.ncand.as<int>(), w.scores.as<float>(),
                                       (int)s->K, T, B, w.cand_cap, w.tokmax.as<uint16_t>(), (const int*)nullptr,
                                       (const int*)nullptr, (const float*)nullptr, w.rowmask.as<unsigned long long>());
                    break;
                default: launch_pass1(s, w, st, dQ, approx_grid, B, T);
            }
#else
            launch_pass1(s, w, st, dQ, approx_grid, B, T);
#endif
#undef CLB_LAUNCH_APPROX
        }
        {
            Timed t(s, KID_SELECT, st);
            CLB_TRY(launch_select(s, w, st, dQ, B, T, k, nullptr, /*coarse_tau=*/phase == 0));
        }
        list = w.list.as<int>();
        nlist = w.nlist.as<int>();
    }
    }   // phase != 2
    if (phase == 1) {
        hipLaunchKernelGGL(local_top_kernel, dim3(B), dim3(1024), 0, st, w.scores.as<float>(), list, nlist,
                           w.thresh.as<float>(), k, w.cand_cap, d_local_top);
        CLB_HIP(hipGetLastError());
        return CLB_OK;
    }
    const bool subset = list && !CLB_KNOB("CLB_DEBUG_NO_SUBSET", 0);   // two-pass mode: re-score only the rows that matter
    if (subset) {
        Timed t(s, KID_ROWS, st);
        // the pass-1 pipeline again, over the listed passages only: marks the rows that can hold a token maximum
        const int rows_gx = CLB_KNOB("CLB_DEBUG_ROWS_GX", 256);
        const dim3 rows_grid = B > 1 ? dim3(std::max(1, rows_gx / B), B) : dim3(8 * 32);
        // (the row sweep keeps the VGPR gather: its ~1 200 passages per query are faster with it on every workload measured)
        auto rows_kernel = w.cell8 ? score_approx32_kernel<true, 0, 0, 0, true> : score_approx32_kernel<true, 0, 0, 0, false>;
        hipLaunchKernelGGL(rows_kernel, rows_grid, dim3(kApproxThreads), 0, st, s->weights.as<float>(),
                           s->codeinv.as<uint32_t>(), s->residuals.as<uint8_t>(), s->cbits, s->inv_lo, s->inv_step, dQ,
                           w.cell8 ? w.cells8.as<uint32_t>() : w.cells_q.as<uint32_t>(), w.cand_hdr.as<uint2>(), w.ncand.as<int>(),
                           w.scores.as<float>(), (int)s->K, T, B, w.cand_cap, w.tokmax.as<uint16_t>(), list, nlist,
                           w.eps_pair.as<float>(), w.rowmask.as<unsigned long long>(), (const float4*)w.tscale.as<float4>());
    }
    {
        Timed t(s, KID_SCORE_EXACT, st);
        const int gxl = CLB_KNOB("CLB_DEBUG_EXACT_GX", 1024);
        const int gx = list ? std::max(1, gxl / B) : std::max(1, 2048 / B);
        const int flat_gx = CLB_KNOB("CLB_DEBUG_FLAT_GX", 768);   // one resident round at 3 work-groups per CU
        if (subset) {
            hipLaunchKernelGGL(score_exact_flat_kernel, dim3(std::max(1, flat_gx / B), B), dim3(256), 0, st, s->centroids.as<float>(),
                               s->weights.as<float>(), s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(),
                               w.cand_hdr.as<uint2>(), dQ, w.scores.as<float>(), T, w.cand_cap, list, nlist,
                               w.rowmask.as<unsigned long long>());
        } else
        switch (s->nbits) {
            case 1: launch_score_exact<1>(s, w, st, dQ, B, T, list, nlist, gx); break;
            case 2: launch_score_exact<2>(s, w, st, dQ, B, T, list, nlist, gx); break;
            case 4: launch_score_exact<4>(s, w, st, dQ, B, T, list, nlist, gx); break;
            default: return fail(CLB_EUNSUPPORTED, "nbits=%d not supported by the HIP search path", s->nbits);
        }
    }
    if (k <= kMaxTopK) {
        Timed t(s, KID_TOPK, st);
        allow_large_topk_lds();
        // two-pass mode: the ~1.2 k listed passages of a query are ranked by kRankBlocks work-groups (no sorting network);
        // a query whose list is longer than kRankMax falls through to the one-work-group select + sort
        const int ranked = list != nullptr && !CLB_KNOB("CLB_DEBUG_NO_RANK", 0);
        if (ranked)
            hipLaunchKernelGGL(topk_rank_kernel, dim3(kRankBlocks, B), dim3(1024), 0, st, w.scores.as<float>(),
                               w.cand.as<uint32_t>(), w.ncand.as<int>(), list, nlist, k, w.cand_cap, s->pid_offset,
                               d_out_pids, d_out_scores, w.flags.as<int>(), d_n_cand);
        hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), sizeof(unsigned long long) * kpow2, st,
                           w.scores.as<float>(), w.cand.as<uint32_t>(), w.ncand.as<int>(), list, nlist, k,
                           kpow2, w.cand_cap, s->pid_offset, d_out_pids, d_out_scores, w.flags.as<int>(), d_n_cand, ranked);
    } else {      // k above the single-work-group sort: a full stable sort per query (synchronises)
        s->prof.chain = nullptr;
        for (int b = 0; b < B; ++b) CLB_TRY(topk_by_sort(s, w, st, b, list, nlist, k, d_out_pids, d_out_scores, d_n_cand));
    }
    if (s->prof.counters) {
        hipLaunchKernelGGL(batch_stats_kernel, dim3(32, B), dim3(256), 0, st, w.cand.as<uint32_t>(),
                           w.ncand.as<int>(), list, nlist, s->doc_off.as<uint32_t>(), w.cand_cap,
                           w.stats.as<unsigned long long>(),
                           subset ? w.rowmask.as<unsigned long long>() : (const unsigned long long*)nullptr);
    }
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

}  // namespace

extern "C" {

#ifdef CLB_ABLATIONS
const char* clb_version(void) { return "colbert_hip 0.1 (gfx950, tuning build: ablation variants and comparison kernels)"; }
#else
const char* clb_version(void) { return "colbert_hip 0.1 (gfx950)"; }
#endif
const char* clb_last_error(void) { return clb::last_error().c_str(); }
int clb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// `big_on_device`: centroids, codes, residuals and ivf are device pointers on `device` (an index that was built there:
// clb_codec_compress_device / clb_build_ivf_device); the per-passage and per-centroid lengths are host arrays either way
static int searcher_create_impl(int device, int64_t dim, int nbits, int64_t K, const float* centroids,
                                const float* bucket_weights, int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                                const uint32_t* codes, const uint8_t* residuals, const int64_t* ivf,
                                const int64_t* ivf_lengths, int64_t pid_offset, bool big_on_device, clb_searcher** out);

int clb_searcher_create(int device, int64_t dim, int nbits, int64_t K, const float* centroids,
                        const float* bucket_weights, int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                        const uint32_t* codes, const uint8_t* residuals, const int64_t* ivf,
                        const int64_t* ivf_lengths, int64_t pid_offset, clb_searcher** out) {
    return searcher_create_impl(device, dim, nbits, K, centroids, bucket_weights, n_docs, doclens, n_emb, codes, residuals,
                                ivf, ivf_lengths, pid_offset, false, out);
}

int clb_searcher_create_device(int device, int64_t dim, int nbits, int64_t K, const float* d_centroids,
                               const float* bucket_weights, int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                               const uint32_t* d_codes, const uint8_t* d_residuals, const int64_t* d_ivf,
                               const int64_t* ivf_lengths, int64_t pid_offset, clb_searcher** out) {
    return searcher_create_impl(device, dim, nbits, K, d_centroids, bucket_weights, n_docs, doclens, n_emb, d_codes,
                                d_residuals, d_ivf, ivf_lengths, pid_offset, true, out);
}

static int searcher_create_impl(int device, int64_t dim, int nbits, int64_t K, const float* centroids,
                                const float* bucket_weights, int64_t n_docs, const int64_t* doclens, int64_t n_emb,
                                const uint32_t* codes, const uint8_t* residuals, const int64_t* ivf,
                                const int64_t* ivf_lengths, int64_t pid_offset, bool big_on_device, clb_searcher** out) {
    const hipMemcpyKind big_kind = big_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!out) return fail(CLB_EARGUMENT, "out is null");
    *out = nullptr;
    if (dim < 8 || dim % 8 != 0) return fail(CLB_EDOMAIN, "dim should be a multiple of 8!");          // residual.jl:763-768
    if (nbits != 1 && nbits != 2 && nbits != 4 && nbits != 8)
        return fail(CLB_EUNSUPPORTED, "the HIP codec supports nbits in {1,2,4,8} (got %d)", nbits);
    if (K < 1 || n_docs < 0 || n_emb < 0) return fail(CLB_EARGUMENT, "negative or empty sizes");
    if (n_emb >= (int64_t)0xffffffffll || n_docs >= (int64_t)0x7fffffffll)
        return fail(CLB_EUNSUPPORTED, "a shard holds at most 2^32-1 embeddings / 2^31-1 passages");
    // host-side structure checks (the reference's DimensionMismatch in _cids_to_eids!, ranking.jl:9-12)
    std::vector<uint32_t> doc_off((size_t)n_docs + 1);
    int64_t run = 0;
    for (int64_t p = 0; p < n_docs; ++p) {
        if (doclens[p] < 0) return fail(CLB_EARGUMENT, "negative doclen at passage %lld", (long long)(p + 1));
        doc_off[p] = (uint32_t)run;
        run += doclens[p];
    }
    doc_off[n_docs] = (uint32_t)run;
    if (run != n_emb) return fail(CLB_EDIMENSION, "sum(doclens)=%lld must equal the number of embeddings %lld", (long long)run, (long long)n_emb);
    std::vector<uint32_t> ivf_off((size_t)K + 1);
    run = 0;
    for (int64_t c = 0; c < K; ++c) {
        if (ivf_lengths[c] < 0) return fail(CLB_EARGUMENT, "negative ivf length");
        ivf_off[c] = (uint32_t)run;
        run += ivf_lengths[c];
    }
    ivf_off[K] = (uint32_t)run;
    if (run != n_emb) return fail(CLB_EDIMENSION, "length(ivf) must be equal to sum(ivf_lengths)!");
    CLB_TRY(use_device(device));

    clb_searcher* s = new clb_searcher();
    s->device = device; s->dim = dim; s->nbits = nbits; s->K = K; s->n_docs = n_docs; s->n_emb = n_emb;
    s->pid_offset = pid_offset;
    s->generic = !(dim == kDim && nbits <= 4);     // the tuned kernels are built for dim 128, nbits 1/2/4
    for (int64_t p = 0; p < n_docs; ++p) s->max_doclen = std::max(s->max_doclen, doclens[p]);
    auto bail = [&](int rc) { clb_searcher_destroy(s); return rc; };
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(CLB_EHIP, "hipStreamCreate failed"));
    s->ivf_len_sorted.resize((size_t)K);
    for (int64_t c = 0; c < K; ++c) s->ivf_len_sorted[c] = (uint32_t)ivf_lengths[c];
    std::sort(s->ivf_len_sorted.begin(), s->ivf_len_sorted.end(), std::greater<uint32_t>());

    const size_t rows = (size_t)(dim / 8 * nbits);
    int rc;
    if ((rc = s->centroids.alloc(sizeof(float) * dim * K))) return bail(rc);
    if (hipMemcpyAsync(s->centroids.p, centroids, sizeof(float) * dim * K, big_kind, s->stream) != hipSuccess)
        return bail(fail(CLB_EHIP, "index upload failed"));
    if ((rc = upload(s->weights, bucket_weights, sizeof(float) * ((size_t)1 << nbits), s->stream))) return bail(rc);
    // pad the per-embedding arrays by one step (dummy steps read embeddings 0 .. kStepRows-1 even of a tiny index)
    constexpr int64_t kPad = kStepRows;
    if ((rc = s->codes0.alloc(sizeof(uint32_t) * (n_emb + kPad)))) return bail(rc);
    if ((rc = s->residuals.alloc(rows * (n_emb + kPad)))) return bail(rc);
    if (hipMemsetAsync(s->codes0.p, 0, s->codes0.bytes, s->stream) != hipSuccess ||
        hipMemsetAsync(s->residuals.p, 0, s->residuals.bytes, s->stream) != hipSuccess ||
        hipMemcpyAsync(s->codes0.p, codes, sizeof(uint32_t) * n_emb, big_kind, s->stream) != hipSuccess ||
        hipMemcpyAsync(s->residuals.p, residuals, rows * n_emb, big_kind, s->stream) != hipSuccess)
        return bail(fail(CLB_EHIP, "index upload failed"));
    if ((rc = upload(s->doc_off, doc_off.data(), sizeof(uint32_t) * doc_off.size(), s->stream))) return bail(rc);
    if ((rc = upload(s->ivf_off, ivf_off.data(), sizeof(uint32_t) * ivf_off.size(), s->stream))) return bail(rc);
    if ((rc = s->ivf_pid.alloc(sizeof(uint32_t) * n_emb))) return bail(rc);
    DevBuf ivf_raw, err;
    if ((rc = ivf_raw.alloc(sizeof(int64_t) * n_emb))) return bail(rc);
    if (n_emb > 0 && hipMemcpyAsync(ivf_raw.p, ivf, sizeof(int64_t) * n_emb, big_kind, s->stream) != hipSuccess)
        return bail(fail(CLB_EHIP, "index upload failed"));
    if ((rc = err.alloc(sizeof(int)))) return bail(rc);
    if (hipMemsetAsync(err.p, 0, sizeof(int), s->stream) != hipSuccess) return bail(fail(CLB_EHIP, "memset failed"));
    if (n_emb > 0) {
        const int blocks = (int)((n_emb + 255) / 256);
        hipLaunchKernelGGL(ivf_to_pid_kernel, dim3(blocks), dim3(256), 0, s->stream, ivf_raw.as<int64_t>(),
                           s->doc_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), n_emb, (int)n_docs, err.as<int>());
        hipLaunchKernelGGL(codes_to_zero_based_kernel, dim3(blocks), dim3(256), 0, s->stream,
                           s->codes0.as<uint32_t>(), n_emb, (uint32_t)K, err.as<int>());
        hipLaunchKernelGGL(ivf_lists_sorted_kernel, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, s->stream,
                           s->ivf_off.as<uint32_t>(), s->ivf_pid.as<uint32_t>(), (int)K, err.as<int>());
    }
    int herr = 0;
    if (hipMemcpyAsync(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess)
        return bail(fail(CLB_EHIP, "index upload failed: %s", hipGetErrorString(hipGetLastError())));
    if (herr & 1) return bail(fail(CLB_EBOUNDS, "ivf holds embedding ids outside 1..n_emb"));
    if (herr & 2) return bail(fail(CLB_EDOMAIN, "All the codes must be in the valid range of centroid IDs!"));
    s->ivf_sorted = !(herr & 4);

    if (!s->generic && n_emb > 0 && !CLB_KNOB("CLB_DEBUG_NO_SORT", 0)) {
        // Order every passage's embeddings by centroid code (results cannot change: MaxSim maximises over a passage's
        // embeddings; row masks, headers and the IVF are positional or per passage).  Equal and neighbouring codes then
        // sit in adjacent lanes of a pass-1 step: their 64-byte score rows coalesce into fewer, larger requests.
        const size_t row_bytes = rows;                    // multiple of 16 for dim 128
        DevBuf keys, keys2, vals, perm, codes_new, res_new;
        if ((rc = keys.alloc(sizeof(uint64_t) * n_emb)) || (rc = keys2.alloc(sizeof(uint64_t) * n_emb)) ||
            (rc = vals.alloc(sizeof(uint32_t) * n_emb)) || (rc = perm.alloc(sizeof(uint32_t) * n_emb)) ||
            (rc = codes_new.alloc(sizeof(uint32_t) * (n_emb + kPad))) || (rc = res_new.alloc(row_bytes * (n_emb + kPad))))
            return bail(rc);
        const int blocks = (int)((n_emb + 255) / 256);
        hipLaunchKernelGGL(passage_code_keys_kernel, dim3(blocks), dim3(256), 0, s->stream, s->codes0.as<uint32_t>(),
                           s->doc_off.as<uint32_t>(), n_emb, (int)n_docs, keys.as<unsigned long long>(), vals.as<uint32_t>());
        if ((rc = sort_pairs_u64(keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<uint32_t>(), perm.as<uint32_t>(),
                                 (size_t)n_emb, s->stream)))
            return bail(rc);
        if (hipMemsetAsync(codes_new.p, 0, codes_new.bytes, s->stream) != hipSuccess ||
            hipMemsetAsync(res_new.p, 0, res_new.bytes, s->stream) != hipSuccess)
            return bail(fail(CLB_EHIP, "memset failed"));
        hipLaunchKernelGGL(permute_codes_kernel, dim3(blocks), dim3(256), 0, s->stream, perm.as<uint32_t>(),
                           s->codes0.as<uint32_t>(), codes_new.as<uint32_t>(), n_emb);
        const int pieces = (int)(row_bytes / 16);
        hipLaunchKernelGGL(permute_rows16_kernel, dim3((unsigned)(((int64_t)n_emb * pieces + 255) / 256)), dim3(256), 0,
                           s->stream, perm.as<uint32_t>(), s->residuals.as<uint4>(), res_new.as<uint4>(), n_emb, pieces);
        if (hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess)
            return bail(fail(CLB_EHIP, "reordering the index failed"));
        std::swap(s->codes0.p, codes_new.p); std::swap(s->codes0.bytes, codes_new.bytes);
        std::swap(s->residuals.p, res_new.p); std::swap(s->residuals.bytes, res_new.bytes);
    }
    if (s->generic) {
        s->mode = 0;
        s->index_bytes = (int64_t)(s->centroids.bytes + s->weights.bytes + s->codes0.bytes + s->residuals.bytes +
                                   s->doc_off.bytes + s->ivf_off.bytes + s->ivf_pid.bytes);
        *out = s;
        return CLB_OK;
    }
    {   // bf16 hi/lo split of the centroids for the bf16x3 centroid scoring
        if ((rc = s->cent_hi.alloc(sizeof(uint16_t) * dim * K))) return bail(rc);
        if ((rc = s->cent_lo.alloc(sizeof(uint16_t) * dim * K))) return bail(rc);
        const int64_t nel = dim * K;
        hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, s->stream,
                           s->centroids.as<float>(), s->cent_hi.as<uint16_t>(), s->cent_lo.as<uint16_t>(), nel);
        if ((rc = s->cent_f16.alloc(sizeof(uint16_t) * dim * K))) return bail(rc);
        hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, s->stream,
                           s->centroids.as<float>(), s->cent_f16.as<uint16_t>(), nel);
        if (const char* g = CLB_ENV("COLBERT_S1_PRODUCTS")) s->s1_x1 = strcmp(g, "1") == 0;      // "1" / "3": comparison runs
    }
    s->cbits = 1;
    while (((int64_t)1 << s->cbits) < K) ++s->cbits;
    // the packed word leaves 32 - cbits bits for inv_norm: at least 12 (K <= 2^20), otherwise exact mode only
    s->approx_ok = approx_supported((int)dim, nbits) && s->cbits <= 20;
    {   // Pass 1's gather form, by the index's own code statistics: when neighbouring embeddings of a passage often share
        // a 128-byte line of the score table (id-adjacent codes: the L1 merges those requests of the per-lane VGPR
        // gather) the VGPR form is faster (1 M topical passages: 0.67 against 0.76 ms per batch); when they do not
        // (uniform codes, a k-means-built index) the LDS-DMA form is (uniform: 1.47 against 1.61 ms)
        DevBuf cnt;
        if ((rc = cnt.alloc(sizeof(unsigned long long)))) return bail(rc);
        unsigned long long adj = 0;
        const int64_t n_sample = std::min<int64_t>(n_emb, (int64_t)1 << 24);
        if (hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s->stream) != hipSuccess) return bail(fail(CLB_EHIP, "memset failed"));
        if (n_sample > 1)
            hipLaunchKernelGGL(code_adjacency_kernel, dim3(1024), dim3(256), 0, s->stream, s->codes0.as<uint32_t>(), n_sample,
                               cnt.as<unsigned long long>());
        if (hipMemcpyAsync(&adj, cnt.p, sizeof adj, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
            hipStreamSynchronize(s->stream) != hipSuccess)
            return bail(fail(CLB_EHIP, "code statistics failed"));
        s->code_adjacency = n_sample > 1 ? (double)adj / (double)(n_sample - 1) : 0.0;
        // ... and only pays when the query's score table (64 B per centroid) does not fit the 4-MB L2 of an XCD: with a
        // resident table the two forms are equal within 2 % (built index, K = 32 768: 0.663 / 0.668 ms)
        s->gather_lds = s->code_adjacency < 0.2 && (int64_t)K * 64 > ((int64_t)4 << 20);
        if (const char* g = CLB_ENV("COLBERT_PASS1_GATHER")) s->gather_lds = strcmp(g, "vgpr") != 0;   // "vgpr" / "lds": comparison runs
    }
    if (s->approx_ok) {
        if ((rc = s->codeinv.alloc(sizeof(uint32_t) * (n_emb + kStepRows)))) return bail(rc);
        if (hipMemsetAsync(s->codeinv.p, 0, s->codeinv.bytes, s->stream) != hipSuccess) return bail(fail(CLB_EHIP, "memset failed"));
    }
    if ((rc = build_approx_tables(s->stream, s->centroids.as<float>(), s->weights.as<float>(),
                                  s->codes0.as<uint32_t>(), s->residuals.as<uint8_t>(), n_emb, (int)K,
                                  s->approx_ok ? s->codeinv.as<uint32_t>() : nullptr, s->cbits, 1 << nbits,
                                  &s->approx_consts, &s->inv_lo, &s->inv_step)))
        return bail(rc);
    // the fp16 side's error lives beside the consts: a table made by the three-product kernels (fewer than 16 queries) keeps
    // the tighter bound.  Infinite (a centroid component beyond the fp16 range): the single-product kernel is never chosen.
    s->dc_f16 = std::isfinite(s->approx_consts.dc_max) ? s->approx_consts.dc_max : 0.f;
    s->approx_consts.dc_max = 0.f;
    s->mode = s->approx_ok ? 1 : 0;
    s->index_bytes = (int64_t)(s->centroids.bytes + s->weights.bytes + s->codes0.bytes + s->residuals.bytes +
                               s->doc_off.bytes + s->ivf_off.bytes + s->ivf_pid.bytes + s->codeinv.bytes);
    *out = s;
    return CLB_OK;
}

int clb_searcher_destroy(clb_searcher* s) {
    if (!s) return CLB_OK;
    (void)hipSetDevice(s->device);
    if (s->stream) {
        (void)hipStreamSynchronize(s->stream);
        for (auto& v : s->prof.pending)
            for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto e : s->prof.pool) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(s->stream);
    }
    delete s;
    return CLB_OK;
}

int64_t clb_searcher_device_bytes(const clb_searcher* s) {
    if (!s) return 0;
    int64_t tot = s->index_bytes;
    for (const auto& w : s->ws) {
        const DevBuf* bufs[] = {&w.Qdev, &w.cells, &w.cells_q, &w.partial, &w.sel, &w.bitmap, &w.blocksum, &w.ncand, &w.cand,
                                &w.cand_hdr, &w.scores, &w.list, &w.nlist, &w.thresh, &w.outp, &w.outs, &w.flags, &w.stats, &w.redo, &w.rowmask, &w.eps_pair, &w.tokmax, &w.tau_glob, &w.tscale, &w.rangep, &w.cells8};
        for (auto* b : bufs) tot += (int64_t)b->bytes;
    }
    return tot;
}

int clb_searcher_set_mode(clb_searcher* s, int mode) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (mode != 0 && mode != 1) return fail(CLB_EARGUMENT, "mode must be 0 (exact) or 1 (two-pass)");
    if (mode == 1 && !s->approx_ok) return fail(CLB_EUNSUPPORTED, "two-pass mode needs dim=128, nbits=2");
    s->mode = mode;
    return CLB_OK;
}
int clb_searcher_get_mode(const clb_searcher* s) { return s ? s->mode : -1; }

int clb_searcher_set_wide_select(clb_searcher* s, int on) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (on < -1 || on > 1) return fail(CLB_EARGUMENT, "wide select must be -1 (by candidate capacity), 0 (never) or 1 (always)");
    s->wide_select = on;
    return CLB_OK;
}

int clb_searcher_sync_bound_consts(clb_searcher* s, clb_comm* c) {
    if (!s || !c) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(s->device));
    float consts[6];
    CLB_TRY(clb_searcher_get_bound_consts(s, consts));
    DevBuf d;
    CLB_TRY(upload(d, consts, sizeof consts, s->stream));
    CLB_TRY(clb_comm_all_reduce_max_f32(c, d.as<float>(), 6, s->stream));
    CLB_HIP(hipMemcpyAsync(consts, d.p, sizeof consts, hipMemcpyDeviceToHost, s->stream));
    CLB_HIP(hipStreamSynchronize(s->stream));
    return clb_searcher_set_bound_consts(s, consts);
}

int clb_searcher_set_pass1_gather(clb_searcher* s, int form) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (form < -1 || form > 1) return fail(CLB_EARGUMENT, "pass-1 gather form must be -1 (by the code statistics), 0 (VGPR) or 1 (LDS-DMA)");
    s->gather_lds = form < 0 ? (s->code_adjacency < 0.2 && s->K * 64 > ((int64_t)4 << 20)) : form;
    return CLB_OK;
}
int clb_searcher_get_pass1_gather(const clb_searcher* s, double* adjacency) {
    if (!s) return -1;
    if (adjacency) *adjacency = s->code_adjacency;
    return s->gather_lds;
}

int clb_searcher_set_score_rows(clb_searcher* s, int form) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (form < -1 || form > 1) return fail(CLB_EARGUMENT, "score rows must be -1 (default: fp16), 0 (64-byte fp16 rows) or 1 (32-byte rows of 8-bit cells)");
    if (form == 1 && !s->approx_ok) return fail(CLB_EUNSUPPORTED, "8-bit score rows need the two-pass mode (dim=128, nbits=2)");
    s->cell8 = form;
    return CLB_OK;
}
int clb_searcher_get_score_rows(const clb_searcher* s) { return s ? (cell8_rows(s) ? 1 : 0) : -1; }

int clb_searcher_set_centroid_products(clb_searcher* s, int n) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    if (n != -1 && n != 1 && n != 3) return fail(CLB_EARGUMENT, "centroid products must be -1 (default), 1 (one fp16 product) or 3 (bf16 split)");
    s->s1_x1 = n == 1 ? 1 : n == 3 ? 0 : -1;
    return CLB_OK;
}
int clb_searcher_get_centroid_products(const clb_searcher* s, float* max_f16_error) {
    if (!s) return -1;
    if (max_f16_error) *max_f16_error = s->dc_f16;
    return (s->s1_x1 == 1 || (s->s1_x1 < 0 && s->bounds_synced)) && s->cent_f16.p && s->dc_f16 > 0.f ? 1 : 3;
}

int clb_searcher_get_bound_consts(const clb_searcher* s, float* consts) {
    if (!s || !consts) return fail(CLB_EARGUMENT, "null argument");
    consts[0] = s->approx_consts.cn_max; consts[1] = s->approx_consts.rn_max; consts[2] = s->approx_consts.inv_max;
    consts[3] = s->approx_consts.rb_max; consts[4] = s->approx_consts.dw_rn; consts[5] = s->approx_consts.inv_qerr;
    return CLB_OK;
}
int clb_searcher_set_bound_consts(clb_searcher* s, const float* consts) {
    if (!s || !consts) return fail(CLB_EARGUMENT, "null argument");
    for (int i = 0; i < 6; ++i)
        if (!(consts[i] >= 0.f)) return fail(CLB_EARGUMENT, "bound constants must be non-negative numbers");
    s->approx_consts.cn_max = std::max(s->approx_consts.cn_max, consts[0]);
    s->approx_consts.rn_max = std::max(s->approx_consts.rn_max, consts[1]);
    s->approx_consts.inv_max = std::max(s->approx_consts.inv_max, consts[2]);
    s->approx_consts.rb_max = std::max(s->approx_consts.rb_max, consts[3]);
    s->approx_consts.dw_rn = std::max(s->approx_consts.dw_rn, consts[4]);
    s->approx_consts.inv_qerr = std::max(s->approx_consts.inv_qerr, consts[5]);
    s->bounds_synced = true;
    return CLB_OK;
}

int clb_search_batch_device(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                            int64_t k, int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand,
                            void* hip_stream) {
    return clb_search_batch_device_slot(s, 0, d_Q, T, B, nprobe, k, d_out_pids, d_out_scores, d_n_cand, hip_stream);
}

int clb_search_batch_device_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, int64_t* d_out_pids, float* d_out_scores, int64_t* d_n_cand,
                                 void* hip_stream) {
    CLB_TRY(check_search_args(s, T, B, nprobe, k));
    if (slot < 0 || slot >= kWorkspaceSlots) return fail(CLB_EARGUMENT, "workspace slot must be 0..%d", kWorkspaceSlots - 1);
    CLB_TRY(use_device(s->device));
    hipStream_t st = (hipStream_t)hip_stream;   // NULL = the HIP null stream, as for any HIP API
    Workspace& w = s->ws[slot];
    w.pending.valid = false;
    // A large batch runs as sub-batches of kSubBatch queries, back to back on the caller's stream and on this slot's scratch:
    // every query carries an 8-MB fp16 score table (K = 131 072), and from ~64 queries on the tables of a batch outgrow
    // the 256-MB Infinity Cache before pass 1 reads them (measured: 29.4 k queries/s at 64, 27.8 k at 256 in one piece);
    // the centroid kernel shares a staged tile between 16 queries whatever the batch, so nothing is lost above that.
    // (The two-phase sharded calls keep the whole batch: phase 2 continues on the scratch of phase 1.)
    for (int64_t b0 = 0; b0 < B; b0 += kSubBatch) {
        const int64_t bn = std::min<int64_t>(kSubBatch, B - b0);
        CLB_TRY(ensure_workspace(s, w, bn, T, nprobe, k));
        w.stats_keep = b0 > 0;
        const int rc = run_search(s, w, st, d_Q + (size_t)b0 * T * s->dim, (int)bn, (int)T, (int)nprobe, (int)k, d_out_pids + (size_t)b0 * k,
                                  d_out_scores + (size_t)b0 * k, d_n_cand ? d_n_cand + b0 : nullptr);
        w.stats_keep = false;
        if (rc) return rc;
    }
    return CLB_OK;
}

int clb_search_shard_phase1(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                            float* d_local_top, void* hip_stream) {
    return clb_search_shard_phase1_slot(s, 0, d_Q, T, B, nprobe, k, d_local_top, hip_stream);
}

int clb_search_shard_phase1_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, float* d_local_top, void* hip_stream) {
    CLB_TRY(check_search_args(s, T, B, nprobe, k));
    if (!d_local_top) return fail(CLB_EARGUMENT, "d_local_top is null");
    if (slot < 0 || slot >= kWorkspaceSlots) return fail(CLB_EARGUMENT, "workspace slot must be 0..%d", kWorkspaceSlots - 1);
    CLB_TRY(use_device(s->device));
    Workspace& w = s->ws[slot];
    CLB_TRY(ensure_workspace(s, w, B, T, nprobe, k));
    w.pending.valid = false;
    CLB_TRY(run_search(s, w, (hipStream_t)hip_stream, d_Q, (int)B, (int)T, (int)nprobe, (int)k, nullptr, nullptr, nullptr,
                       1, d_local_top));
    w.pending.valid = true; w.pending.dQ = d_Q; w.pending.T = T; w.pending.B = B; w.pending.nprobe = nprobe;
    w.pending.k = k; w.pending.stream = hip_stream;
    return CLB_OK;
}

int clb_search_shard_phase2(clb_searcher* s, const float* d_Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                            const float* d_all_top, int64_t n_shards, int64_t* d_out_pids, float* d_out_scores,
                            int64_t* d_n_cand, void* hip_stream) {
    return clb_search_shard_phase2_slot(s, 0, d_Q, T, B, nprobe, k, d_all_top, n_shards, d_out_pids, d_out_scores, d_n_cand,
                                        hip_stream);
}

int clb_search_shard_phase2_slot(clb_searcher* s, int slot, const float* d_Q, int64_t T, int64_t B, int64_t nprobe,
                                 int64_t k, const float* d_all_top, int64_t n_shards, int64_t* d_out_pids,
                                 float* d_out_scores, int64_t* d_n_cand, void* hip_stream) {
    CLB_TRY(check_search_args(s, T, B, nprobe, k));
    if (!d_all_top || n_shards < 1) return fail(CLB_EARGUMENT, "d_all_top is null or n_shards < 1");
    if (slot < 0 || slot >= kWorkspaceSlots) return fail(CLB_EARGUMENT, "workspace slot must be 0..%d", kWorkspaceSlots - 1);
    CLB_TRY(use_device(s->device));
    Workspace& w = s->ws[slot];
    const auto& pd = w.pending;
    if (!pd.valid || pd.dQ != d_Q || pd.T != T || pd.B != B || pd.nprobe != nprobe || pd.k != k || pd.stream != hip_stream)
        return fail(CLB_EARGUMENT, "clb_search_shard_phase2 without a matching clb_search_shard_phase1 "
                                   "on this workspace slot (same queries, T, B, nprobe, k and stream, and no other search on the slot in between)");
    // every shard must cut at tau_global - 2 eps with ONE eps (the largest): a shard still on its own bound constants
    // could drop a member of the global top-k silently, so phase 2 with other shards' scores is refused until the host
    // has shared them (clb_searcher_get_bound_consts on every shard -> element-wise maximum -> clb_searcher_set_bound_consts)
    if (n_shards > 1 && !s->bounds_synced)
        return fail(CLB_EARGUMENT, "clb_search_shard_phase2 with %lld shards before clb_searcher_set_bound_consts: the shards "
                                   "must share one error bound (all-reduce MAX of clb_searcher_get_bound_consts)", (long long)n_shards);
    w.pending.valid = false;
    return run_search(s, w, (hipStream_t)hip_stream, d_Q, (int)B, (int)T, (int)nprobe, (int)k, d_out_pids, d_out_scores,
                      d_n_cand, 2, nullptr, d_all_top, (int)n_shards);
}

int clb_search_batch(clb_searcher* s, const float* Q, int64_t T, int64_t B, int64_t nprobe, int64_t k,
                     int pad_short, int64_t* out_pids, float* out_scores, int64_t* n_cand) {
    CLB_TRY(check_search_args(s, T, B, nprobe, k));
    CLB_TRY(use_device(s->device));
    Workspace& w = s->ws[0];
    w.pending.valid = false;
    hipStream_t st = s->stream;
    std::vector<int> nc((size_t)B), fl((size_t)B);
    for (int64_t b0 = 0; b0 < B; b0 += kSubBatch) {           // sub-batches: see clb_search_batch_device_slot
        const int64_t bn = std::min<int64_t>(kSubBatch, B - b0);
        CLB_TRY(ensure_workspace(s, w, bn, T, nprobe, k));
        CLB_HIP(hipMemcpyAsync(w.Qdev.p, Q + (size_t)b0 * T * s->dim, sizeof(float) * bn * T * s->dim, hipMemcpyHostToDevice, st));
        w.stats_keep = b0 > 0;
        const int rc = run_search(s, w, st, w.Qdev.as<float>(), (int)bn, (int)T, (int)nprobe, (int)k, w.outp.as<int64_t>(), w.outs.as<float>());
        w.stats_keep = false;
        if (rc) return rc;
        CLB_HIP(hipMemcpyAsync(out_pids + (size_t)b0 * k, w.outp.p, sizeof(int64_t) * bn * k, hipMemcpyDeviceToHost, st));
        CLB_HIP(hipMemcpyAsync(out_scores + (size_t)b0 * k, w.outs.p, sizeof(float) * bn * k, hipMemcpyDeviceToHost, st));
        CLB_HIP(hipMemcpyAsync(nc.data() + b0, w.ncand.p, sizeof(int) * bn, hipMemcpyDeviceToHost, st));
        CLB_HIP(hipMemcpyAsync(fl.data() + b0, w.flags.p, sizeof(int) * bn, hipMemcpyDeviceToHost, st));
    }
    CLB_HIP(hipStreamSynchronize(st));
    int64_t docs = 0;
    for (int64_t b = 0; b < B; ++b) {
        if (n_cand) n_cand[b] = nc[b];
        docs += nc[b];
    }
    s->last_cand_docs = docs;
    if (!pad_short)
        for (int64_t b = 0; b < B; ++b)
            if (fl[b])  // searching.jl:127 `pids[1:k]` on a shorter vector
                return fail(CLB_EBOUNDS, "query %lld has %d candidate passages, fewer than k=%lld", (long long)b, nc[b], (long long)k);
    return CLB_OK;
}

int clb_search(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t k, int64_t* out_pids,
               float* out_scores, int64_t* n_cand) {
    return clb_search_batch(s, Q, T, 1, nprobe, k, 0, out_pids, out_scores, n_cand);
}

int clb_retrieve(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t* out_pids,
                 int64_t* n_out) {
    CLB_TRY(check_search_args(s, T, 1, nprobe, 1));
    CLB_TRY(use_device(s->device));
    Workspace& w = s->ws[0];
    w.pending.valid = false;
    CLB_TRY(ensure_workspace(s, w, 1, T, nprobe, 1));
    hipStream_t st = s->stream;
    CLB_HIP(hipMemcpyAsync(w.Qdev.p, Q, sizeof(float) * T * s->dim, hipMemcpyHostToDevice, st));
    if (s->generic || T > 128) CLB_TRY(run_retrieve_general(s, w, st, w.Qdev.as<float>(), 0, (int)T, (int)nprobe));
    else CLB_TRY(run_retrieve(s, w, st, w.Qdev.as<float>(), 1, (int)T, (int)nprobe));
    int nc = 0;
    CLB_HIP(hipMemcpyAsync(&nc, w.ncand.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> c((size_t)nc);
    if (nc) CLB_HIP(hipMemcpy(c.data(), w.cand.p, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; ++i) out_pids[i] = s->pid_offset + (int64_t)c[i] + 1;
    *n_out = nc;
    return CLB_OK;
}

static int merge_topk_launch(int device, const int64_t* d_pids, const float* d_scores, int64_t k, int64_t n_lists,
                             int64_t B, size_t pid_stride, size_t score_stride, int64_t* d_out_pids,
                             float* d_out_scores, void* hip_stream) {
    if (k < 1 || n_lists < 1 || B < 1) return fail(CLB_EARGUMENT, "k, n_lists and B must be >= 1");
    CLB_TRY(use_device(device));
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)((B * k + 255) / 256)), dim3(256), 0, st, d_out_pids,
                       d_out_scores, B * k);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)((n_lists * k + 255) / 256), (unsigned)B), dim3(256), 0, st,
                       d_pids, d_scores, (int)k, (int)n_lists, (int)B, pid_stride, score_stride, d_out_pids,
                       d_out_scores);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

int clb_merge_topk_device(int device, const int64_t* d_pids, const float* d_scores, int64_t k, int64_t n_lists,
                          int64_t B, int64_t* d_out_pids, float* d_out_scores, void* hip_stream) {
    return merge_topk_launch(device, d_pids, d_scores, k, n_lists, B, (size_t)(B * k), (size_t)(B * k), d_out_pids,
                             d_out_scores, hip_stream);
}

int64_t clb_packed_topk_bytes(int64_t k, int64_t B) { return (B * k * 12 + 7) / 8 * 8; }

int clb_merge_topk_packed_device(int device, const void* d_packed, int64_t k, int64_t n_lists, int64_t B,
                                 int64_t* d_out_pids, float* d_out_scores, void* hip_stream) {
    if (k < 1 || B < 1) return fail(CLB_EARGUMENT, "k, n_lists and B must be >= 1");
    const size_t block = (size_t)clb_packed_topk_bytes(k, B);
    const char* base = static_cast<const char*>(d_packed);
    return merge_topk_launch(device, reinterpret_cast<const int64_t*>(base),
                             reinterpret_cast<const float*>(base + (size_t)B * k * 8), k, n_lists, B, block / 8, block / 4,
                             d_out_pids, d_out_scores, hip_stream);
}

int clb_debug_scores(clb_searcher* s, const float* Q, int64_t T, int64_t nprobe, int64_t k, int64_t cap,
                     int64_t* out_pids, float* out_approx, float* out_exact, int64_t* n_out, float* tau,
                     float* eps, int64_t* n_rescore) {
    CLB_TRY(check_search_args(s, T, 1, nprobe, k));
    if (!s->approx_ok || T > 32) return fail(CLB_EUNSUPPORTED, "two-pass mode not available for this index/query");
    CLB_TRY(use_device(s->device));
    Workspace& w = s->ws[0];
    w.pending.valid = false;
    // 8-bit score rows exist for batches of 16+ queries only: the query then runs as sixteen copies of itself (the report is
    // copy 0's), so that this hook sees the table format, the scaled query operand and the bound a real batch gets
    const int Bd = cell8_rows(s) ? kTeamQueries : 1;
    CLB_TRY(ensure_workspace(s, w, Bd, T, nprobe, k));
    hipStream_t st = s->stream;
    for (int c = 0; c < Bd; ++c)
        CLB_HIP(hipMemcpyAsync(w.Qdev.as<float>() + (size_t)c * T * kDim, Q, sizeof(float) * T * kDim, hipMemcpyHostToDevice, st));
    const float* dQ = w.Qdev.as<float>();
    CLB_TRY(run_retrieve(s, w, st, dQ, Bd, (int)T, (int)nprobe));
    launch_pass1(s, w, st, dQ, dim3(8 * 32), Bd, (int)T);
    hipLaunchKernelGGL(select_margin_kernel, dim3(1), dim3(1024), 0, st, w.scores.as<float>(), w.ncand.as<int>(), dQ,
                       (int)T, (int)k, w.cand_cap, s->approx_consts, w.list.as<int>(), w.nlist.as<int>(),
                       w.thresh.as<float>(), w.eps_pair.as<float>(), (const float*)nullptr, 0,
                       w.have_range ? (const float4*)w.tscale.as<float4>() : (const float4*)nullptr, w.cell8 ? 1 : 0);
    int nc = 0, nl = 0;
    float th[2];
    CLB_HIP(hipMemcpyAsync(&nc, w.ncand.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemcpyAsync(&nl, w.nlist.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemcpyAsync(th, w.thresh.p, sizeof th, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    if (nc > cap) return fail(CLB_EARGUMENT, "output capacity %lld < %d candidates", (long long)cap, nc);
    std::vector<uint32_t> c((size_t)nc);
    if (nc) {
        CLB_HIP(hipMemcpy(c.data(), w.cand.p, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost));
        CLB_HIP(hipMemcpy(out_approx, w.scores.p, sizeof(float) * nc, hipMemcpyDeviceToHost));
    }
    switch (s->nbits) {
        case 2: launch_score_exact<2>(s, w, st, dQ, 1, (int)T, nullptr, nullptr, 2048); break;
        default: return fail(CLB_EUNSUPPORTED, "nbits");
    }
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipStreamSynchronize(st));
    if (nc) CLB_HIP(hipMemcpy(out_exact, w.scores.p, sizeof(float) * nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; ++i) out_pids[i] = s->pid_offset + (int64_t)c[i] + 1;
    *n_out = nc; *tau = th[0]; *eps = th[1]; *n_rescore = nl;
    return CLB_OK;
}

int clb_profile_enable(clb_searcher* s, int on) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    s->prof.on = on != 0;
    s->prof.counters = on >= 2;
    return CLB_OK;
}

int clb_profile_read(clb_searcher* s, const char** names, double* total_ms, int64_t* launches, int cap) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    int n = 0;
    std::vector<hipEvent_t> seen;       // chained kernels share events: every event goes back to the pool once
    s->prof.chain = nullptr;
    for (int id = 0; id < KID_COUNT && n < cap; ++id) {
        for (auto& pr : s->prof.pending[id]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                s->prof.total_ms[id] += ms;
                s->prof.launches[id] += 1;
            }
            seen.push_back(pr.first);
            seen.push_back(pr.second);
        }
        s->prof.pending[id].clear();
        names[n] = kKernelNames[id];
        total_ms[n] = s->prof.total_ms[id];
        launches[n] = s->prof.launches[id];
        s->prof.total_ms[id] = 0;
        s->prof.launches[id] = 0;
        ++n;
    }
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    s->prof.pool.insert(s->prof.pool.end(), seen.begin(), seen.end());
    if (s->prof.failed) {               // some launches went untimed: the totals above are incomplete
        s->prof.failed = false;
        (void)fail(CLB_EHIP, "HIP event creation/record failed while profiling: timings are incomplete");
        return -1;
    }
    return n;
}

int clb_last_batch_stats(clb_searcher* s, int64_t* cand_docs, int64_t* cand_embs, int64_t* rescored_docs,
                         int64_t* rescored_embs) {
    if (!s) return fail(CLB_EARGUMENT, "null searcher");
    CLB_TRY(use_device(s->device));
    CLB_HIP(hipDeviceSynchronize());
    // computed on demand from the device-side counts of the last batch
    unsigned long long h[8] = {0};
    for (auto& w : s->ws) {
        unsigned long long t[8] = {0};
        if (w.stats.p) CLB_HIP(hipMemcpy(t, w.stats.p, sizeof t, hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) h[i] += t[i];
    }
    if (cand_docs) *cand_docs = (int64_t)h[0];
    if (cand_embs) *cand_embs = (int64_t)h[1];
    if (rescored_docs) *rescored_docs = (int64_t)h[2];
    if (rescored_embs) *rescored_embs = (int64_t)h[3];
    return CLB_OK;
}

}  // extern "C"
